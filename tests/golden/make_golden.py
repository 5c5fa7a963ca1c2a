#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container only).

What produces the numbers: the locally installed third-party `transformers` package
(`WhisperForConditionalGeneration`, `WhisperFeatureExtractor`) — the upstream definition of the
graphs the reference executes through ONNX Runtime (reference scripts/export_onnx_whisper.py:19-28).
It is NOT the reference and nothing from /root/reference is imported or executed here.  Models
are built from a local `WhisperConfig` (no `from_pretrained`, no network) and loaded with the
hash-seeded weights of whisper-rust-ort_amd/modelspec.py.

The greedy loop below re-states reference src/main.rs:753-829 on top of the HF forward (full
decoder for the prompt, then one token per call with `past_key_values`), because HF `generate`
adds logits processors the reference does not have.

Outputs (small, committed):  tests/golden/<preset>_s<seed>_c<clip>.npz
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

from transformers import WhisperConfig, WhisperForConditionalGeneration, WhisperFeatureExtractor  # noqa: E402

ENC_ROWS = (0, 1, 2, 3, 748, 749, 750, 751, 1496, 1497, 1498, 1499)


def build_hf(dims: ms.WhisperDims, sd_np):
    small = dims.vocab < 50300
    cfg = WhisperConfig(
        vocab_size=dims.vocab, num_mel_bins=dims.n_mels, d_model=dims.d_model,
        encoder_layers=dims.enc_layers, encoder_attention_heads=dims.n_heads,
        decoder_layers=dims.dec_layers, decoder_attention_heads=dims.n_heads,
        encoder_ffn_dim=dims.ffn, decoder_ffn_dim=dims.ffn,
        max_source_positions=dims.n_audio_ctx, max_target_positions=dims.n_text_ctx,
        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        pad_token_id=0 if small else 50256, bos_token_id=1 if small else 50257,
        eos_token_id=2 if small else 50257, decoder_start_token_id=3 if small else 50258,
        suppress_tokens=None, begin_suppress_tokens=None,
    )
    cfg._attn_implementation = "eager"
    model = WhisperForConditionalGeneration(cfg).eval()
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in m for m in missing), missing
    return model


def masked_argmax(row: np.ndarray, suppress) -> int:
    """reference src/main.rs:709-735"""
    best_i, best_v = 0, -np.inf
    sup = set(int(s) for s in suppress)
    for i, v in enumerate(row.tolist()):
        if i in sup:
            continue
        if v > best_v:
            best_v, best_i = v, i
    return best_i


@torch.no_grad()
def greedy(model, enc, prompt, max_new, eot, suppress, begin_suppress, forced=None):
    """reference src/main.rs:753-829 on the HF forward; returns tokens and per-step logits rows"""
    tokens = list(prompt)
    rows = []
    out = model(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([tokens]), use_cache=True)
    past = out.past_key_values
    row = out.logits[0, -1].numpy().astype(np.float32)
    rows.append(row)
    nxt = masked_argmax(row, list(suppress) + list(begin_suppress))
    tokens.append(nxt)
    gen = 1
    if forced is not None and gen <= len(forced):
        nxt = forced[gen - 1]
    elif nxt == eot:
        return tokens, rows
    for _ in range(1, max_new):
        out = model(encoder_outputs=(enc,), decoder_input_ids=torch.tensor([[nxt]]),
                    past_key_values=past, use_cache=True)
        past = out.past_key_values
        row = out.logits[0, -1].numpy().astype(np.float32)
        rows.append(row)
        nxt = masked_argmax(row, suppress)
        tokens.append(nxt)
        gen += 1
        if forced is not None and gen <= len(forced):
            nxt = forced[gen - 1]
        elif nxt == eot:
            break
    return tokens, rows


def topk(row: np.ndarray, k: int):
    idx = np.argsort(-row, kind="stable")[:k]
    return idx.astype(np.int32), row[idx].astype(np.float32)


def make(preset: str, seed: int, clip: int, max_new: int, out_dir: str) -> str:
    torch.set_num_threads(8)
    dims = ms.PRESETS[preset]
    sd = ms.synth_state_dict(dims, seed)
    model = build_hf(dims, sd)
    pcm = ms.synth_clip(clip)
    fe = WhisperFeatureExtractor(feature_size=dims.n_mels)
    mel = fe(pcm, sampling_rate=16000, return_tensors="np").input_features[0].astype(np.float32)
    assert mel.shape == (dims.n_mels, 3000)
    with torch.no_grad():
        enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state
    enc_np = enc[0].numpy().astype(np.float32)

    if dims.vocab == 51866:
        prompt, eot = [50258, 50259, 50360, 50364], 50257  # large-v3 ids of the same four specials (one more language token)
    elif dims.vocab > 50400:
        prompt, eot = [50258, 50259, 50359, 50363], 50257  # reference src/main.rs:549-566
    else:
        prompt, eot = [3, 5, 7, 9], 2
    # (a) free-running greedy, empty suppress sets (generation_config.json absent → :651-653)
    toks_a, rows_a = greedy(model, enc, prompt, max_new, eot, [], [])
    # (b) greedy with EOT + a few ids suppressed and a begin-suppress set
    sup = [eot, int(toks_a[len(prompt)])]          # forbid the unconstrained first choice as well
    bsup = [int(toks_a[len(prompt) + 1])] if len(toks_a) > len(prompt) + 1 else [1]
    toks_b, rows_b = greedy(model, enc, prompt, max_new, eot, sup, bsup)
    # (c) teacher-forced with hash-random tokens: exercises the cache with a non-repeating history
    rng = np.random.Generator(np.random.PCG64(seed * 1000 + clip))
    n_forced = min(max_new, 24) - 1
    forced = rng.integers(0, dims.vocab, size=n_forced).tolist()
    toks_c, rows_c = greedy(model, enc, prompt, n_forced + 1, eot, [], [], forced=forced)

    k = 8
    def pack(rows):
        ids, vals = zip(*[topk(r, k) for r in rows])
        return np.stack(ids), np.stack(vals)

    ta_i, ta_v = pack(rows_a)
    tb_i, tb_v = pack(rows_b)
    tc_i, tc_v = pack(rows_c)
    full_rows = min(len(rows_c), 4)
    wsum = np.array([np.float64(sd[n].astype(np.float64).sum()) for n, _ in ms.tensor_table(dims)])
    out = os.path.join(out_dir, f"{preset}_s{seed}_c{clip}.npz")
    np.savez_compressed(
        out,
        preset=preset, seed=seed, clip=clip, max_new=max_new,
        pcm_sha256=hashlib.sha256(pcm.tobytes()).hexdigest(),
        pcm_head=pcm[:64],
        weight_sums=wsum,
        mel_slice=mel[:, ::25].copy(),                     # [n_mels, 120]
        mel_sum=np.float64(mel.astype(np.float64).sum()),
        mel_abs_sum=np.float64(np.abs(mel.astype(np.float64)).sum()),
        enc_rows=np.asarray(ENC_ROWS, np.int32),
        enc_slice=enc_np[list(ENC_ROWS)].copy(),
        enc_col_mean=enc_np.astype(np.float64).mean(axis=0).astype(np.float32),
        enc_abs_sum=np.float64(np.abs(enc_np.astype(np.float64)).sum()),
        prompt=np.asarray(prompt, np.int64), eot=eot,
        tokens_a=np.asarray(toks_a, np.int64), top_ids_a=ta_i, top_vals_a=ta_v,
        suppress_b=np.asarray(sup, np.int64), begin_suppress_b=np.asarray(bsup, np.int64),
        tokens_b=np.asarray(toks_b, np.int64), top_ids_b=tb_i, top_vals_b=tb_v,
        forced_c=np.asarray(forced, np.int64),
        tokens_c=np.asarray(toks_c, np.int64), top_ids_c=tc_i, top_vals_c=tc_v,
        logits_c_head=np.stack(rows_c[:full_rows])[:, :2048].copy(),
        logits_c_rowsum=np.array([np.float64(r.astype(np.float64).sum()) for r in rows_c]),
    )
    print(f"wrote {out}: tokens_a={toks_a[:12]}… n={len(toks_a)} size={os.path.getsize(out)/1024:.1f} KiB")
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    jobs = [("nano", 7, 0, 24), ("nano", 7, 1, 24), ("micro", 11, 2, 32), ("base", 1234, 0, 128),
            ("base", 1234, 3, 128), ("large-v3", 5, 7, 12)]
    for preset, seed, clip, max_new in jobs:
        if a.only and a.only != preset:
            continue
        make(preset, seed, clip, max_new, a.out_dir)


if __name__ == "__main__":
    main()
