#!/usr/bin/env python3
"""How accurate is a split-bf16 ("bf16x3") contraction against f32?  CPU experiment, build container only.

Every contraction of the HF Whisper forward (Linear, Conv1d, the two attention matmuls) is replaced by
    x = hi(x) + lo(x),  hi = bf16(x), lo = bf16(x - hi)        (both operands)
    x . w ~= hi.hi + hi.lo + lo.hi                              (lo.lo dropped; f32 accumulation)
i.e. three bf16 matrix-core products per contraction.  `--terms 6` adds a third limb (hi, mid, lo: six products).
Prints max |delta logit| of teacher-forced steps against the unpatched f32 forward on the same weights.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402
import make_golden as mg  # noqa: E402

TERMS = 3
LIMB = torch.bfloat16   # --limb f16: the same split with fp16 limbs (11 + 11 bits; range handled by the caller's scaling)
ROUND_ONLY = False   # operands rounded to hi + lo, then ONE f32 product (isolates the dropped lo.lo term)


def split(x):
    hi = x.to(LIMB).to(torch.float32)
    r = x - hi
    lo = r.to(LIMB).to(torch.float32)
    if TERMS == 3:
        return hi, lo, None
    lo2 = (r - lo).to(LIMB).to(torch.float32)
    return hi, lo, lo2


SCALE_A, SCALE_B = 1.0, 1.0   # exact power-of-two operand scales (--scale-a/-b log2), undone on the result


def mm3(a, b, op):
    if SCALE_A != 1.0 or SCALE_B != 1.0:
        sa, sb = SCALE_A, SCALE_B
        return _mm3(a * sa, b * sb, op) * (1.0 / (sa * sb))
    return _mm3(a, b, op)


def _mm3(a, b, op):
    ah, al, al2 = split(a)
    bh, bl, bl2 = split(b)
    if ROUND_ONLY:
        return op(ah + al, bh + bl)
    out = op(ah, bh) + (op(ah, bl) + op(al, bh))
    if TERMS == 6:
        out = out + (op(al, bl) + op(ah, bl2) + op(al2, bh))
    return out


_linear, _conv1d, _matmul = F.linear, F.conv1d, torch.matmul


def linear(x, w, b=None):
    y = mm3(x, w, lambda p, q: _linear(p, q))
    return y if b is None else y + b


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    y = mm3(x, w, lambda p, q: _conv1d(p, q, None, stride, padding, dilation, groups))
    return y if b is None else y + b[None, :, None]


VARIANT = ""   # --variant: operands carried as ONE fp16 limb in chosen places (what would it cost in accuracy?)


def matmul(a, b):
    # HF eager attention: scores = matmul(q, k^T) with k^T [.., 64, keys]; output = matmul(P, V) with P [.., queries, keys], V [.., keys, 64]
    is_scores = b.shape[-2] == 64 and a.shape[-1] == 64
    keys = b.shape[-1] if is_scores else b.shape[-2]
    enc_self = keys == 1500 and a.shape[-2] == 1500
    dec_self = keys <= 448
    ah, al, _ = split(a)
    bh, bl, _ = split(b)
    if VARIANT == "enc_p_single" and enc_self and not is_scores:          # encoder P . V with P as one limb
        return _matmul(ah, bh) + _matmul(ah, bl)
    if VARIANT == "dec_self_kv_single" and dec_self:                      # decoder self-attention with the cached K and V as one limb
        return _matmul(ah, bh) + _matmul(al, bh)
    if VARIANT == "enc_p_and_dec_kv" and ((enc_self and not is_scores) or dec_self):
        return (_matmul(ah, bh) + _matmul(ah, bl)) if (enc_self and not is_scores) else (_matmul(ah, bh) + _matmul(al, bh))
    return mm3(a, b, _matmul)


def run(model, mel, prompt, forced, eot):
    with torch.no_grad():
        enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state
    toks, rows = mg.greedy(model, enc, prompt, len(forced) + 1, eot, [], [], forced=forced)
    return enc[0].numpy(), np.stack(rows)


def main():
    global TERMS, ROUND_ONLY, LIMB, SCALE_A, SCALE_B, VARIANT
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="base")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--clip", type=int, default=0)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--terms", type=int, default=3)
    ap.add_argument("--round-only", action="store_true")
    ap.add_argument("--limb", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--variant", default="", choices=["", "enc_p_single", "dec_self_kv_single", "enc_p_and_dec_kv"])
    ap.add_argument("--scale-a", type=int, default=0)
    ap.add_argument("--scale-b", type=int, default=0)
    a = ap.parse_args()
    TERMS, ROUND_ONLY = a.terms, a.round_only
    LIMB = torch.float16 if a.limb == "f16" else torch.bfloat16
    SCALE_A, SCALE_B = 2.0 ** a.scale_a, 2.0 ** a.scale_b
    VARIANT = a.variant
    torch.set_num_threads(8)
    dims = ms.PRESETS[a.preset]
    sd = ms.synth_state_dict(dims, a.seed)
    model = mg.build_hf(dims, sd)
    pcm = ms.synth_clip(a.clip)
    from transformers import WhisperFeatureExtractor
    mel = WhisperFeatureExtractor(feature_size=dims.n_mels)(pcm, sampling_rate=16000, return_tensors="np").input_features[0].astype(np.float32)
    if dims.vocab == 51866:
        prompt, eot = [50258, 50259, 50360, 50364], 50257
    elif dims.vocab > 50400:
        prompt, eot = [50258, 50259, 50359, 50363], 50257
    else:
        prompt, eot = [3, 5, 7, 9], 2
    rng = np.random.Generator(np.random.PCG64(a.seed * 1000 + a.clip))
    forced = rng.integers(0, dims.vocab, size=a.steps - 1).tolist()
    enc0, rows0 = run(model, mel, prompt, forced, eot)
    F.linear, F.conv1d, torch.matmul = linear, conv1d, matmul
    torch.nn.functional.linear = linear
    try:
        enc1, rows1 = run(model, mel, prompt, forced, eot)
    finally:
        F.linear, F.conv1d, torch.matmul = _linear, _conv1d, _matmul
    d = np.abs(rows1 - rows0)
    top = np.sort(rows0, axis=1)[:, -2:]
    print(f"preset {a.preset} variant {a.variant!r} limb {a.limb} scale 2^{a.scale_a},2^{a.scale_b} terms {TERMS} round_only {ROUND_ONLY}: encoder max |d| {np.abs(enc1 - enc0).max():.3e} (|enc| max {np.abs(enc0).max():.2f}); "
          f"logits max |d| {d.max():.3e} mean {d.mean():.3e}; logit sigma {rows0.std():.2f}; min top-1 margin {np.min(top[:, 1] - top[:, 0]):.3e}; "
          f"argmax equal {int((rows1.argmax(1) == rows0.argmax(1)).sum())}/{len(rows0)}")


if __name__ == "__main__":
    main()
