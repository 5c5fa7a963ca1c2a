"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors.  Run with `-m gpu` on an MI355X.

Tolerances
  F32 mode (exact-f32 MFMA) and F16X3 mode (two fp16 limbs per operand, three fp16 MFMAs per product: the in-tolerance mode
  at matrix-core speed): mel 1e-4, encoder 1e-3, logits 1e-3 (north_star), tokens exact — the same tests, parametrised.
  BF16 mode: reported as error statistics; tokens must match wherever the oracle's top-1 margin
  exceeds the measured logit error bound (teacher-forced), see test_bf16_*.
"""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as orc
from whisper_rust_ort_amd import binding as wb
from whisper_rust_ort_amd import modelspec as ms

pytestmark = pytest.mark.gpu

MEL_TOL, ENC_TOL, LOGIT_TOL = 1e-4, 1e-3, 1e-3
EXACT_MODES = ["f32", "f16x3"]   # the two precisions held to north_star's bar (tokens identical, logits within 1e-3)


@pytest.fixture(scope="module")
def gpu():
    if wb.device_count() < 1:
        pytest.fail("no MI355X visible: the GPU suite has no fallback")
    return 0


class Bundle:
    def __init__(self, preset, seed, prec, max_batch=1):
        self.dims = ms.PRESETS[preset]
        self.model = wb.Model(f"synthetic:{preset}:{seed}", 0, prec)
        self.ctx = wb.Context(self.model, max_batch)
        self._w = None
        self.preset, self.seed = preset, seed

    @property
    def w(self):
        if self._w is None:
            self._w = ms.flatten_state_dict(self.dims, ms.synth_state_dict(self.dims, self.seed))
        return self._w


_cache = {}


def bundle(preset, seed, prec, max_batch=1):
    k = (preset, seed, prec, max_batch)
    if k not in _cache:
        if max_batch > 256:      # tens of GB of workspace each: keep one at a time
            for kk in [kk for kk in _cache if kk[3] > 256]:
                _cache.pop(kk).ctx.close()
        _cache[k] = Bundle(preset, seed, prec, max_batch)
    return _cache[k]


def small_prompt(dims):
    return ([50258, 50259, 50359, 50363], 50257) if dims.vocab > 50400 else ([3, 5, 7, 9], 2)


# ------------------------------------------------------------------------------------------------
# log-mel
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [480000, 479999, 160000, 16000, 12345, 400, 321, 320, 161, 160, 159, 2, 1])
def test_log_mel_matches_oracle(gpu, n):
    b = bundle("nano", 7, wb.WH_PREC_F32)
    pcm = ms.synth_clip(5)[:n]
    got = b.ctx.whisper_log_mel(pcm)
    ref = orc.log_mel(pcm, 80)
    assert got.shape == ref.shape == (80, max(1, n // 160) if n >= 160 else 1)
    np.testing.assert_allclose(got, ref, rtol=0, atol=MEL_TOL)


def test_log_mel_long_file_global_max(gpu):
    b = bundle("nano", 7, wb.WH_PREC_F32)
    pcm = np.concatenate([ms.synth_clip(1) * 0.01, ms.synth_clip(2), ms.synth_clip(3)[:123457] * 0.1])
    got = b.ctx.whisper_log_mel(pcm)
    ref = orc.log_mel(pcm, 80)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=MEL_TOL)


def test_log_mel_silence_and_golden(gpu, golden_dir):
    b = bundle("nano", 7, wb.WH_PREC_F32)
    mel = b.ctx.whisper_log_mel(np.zeros(16000, np.float32))
    np.testing.assert_allclose(mel, np.full((80, 100), -1.5, np.float32), rtol=0, atol=1e-6)  # device log10f: 1 ulp
    g = np.load(os.path.join(golden_dir, "nano_s7_c0.npz"))
    got = b.ctx.whisper_log_mel(ms.synth_clip(0))
    np.testing.assert_allclose(got[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)


def test_empty_audio_is_rejected(gpu):
    b = bundle("nano", 7, wb.WH_PREC_F32)
    with pytest.raises(wb.WhisperHipError) as ei:
        b.ctx.whisper_log_mel(np.zeros(0, np.float32))
    assert ei.value.code == 1 and "Empty audio" in str(ei.value)   # src/main.rs:414-416
    with pytest.raises(wb.WhisperHipError) as ei:
        b.ctx.transcribe_batch([np.zeros(0, np.float32)], wb.DecodeParams([3, 5, 7, 9], 4, 2))
    assert ei.value.code == 1


# ------------------------------------------------------------------------------------------------
# encoder + decode, exact-f32 mode, against the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec_name", EXACT_MODES)
@pytest.mark.parametrize("preset,seed,clip", [("nano", 7, 0), ("micro", 11, 2)])
def test_f32_full_path_matches_oracle(gpu, preset, seed, clip, prec_name):
    b = bundle(preset, seed, wb.PRECISIONS[prec_name])
    dims = b.dims
    pcm = ms.synth_clip(clip)
    mel = b.ctx.whisper_log_mel(pcm)
    mel_ref = orc.log_mel(pcm, dims.n_mels)
    np.testing.assert_allclose(mel, mel_ref, rtol=0, atol=MEL_TOL)
    enc = b.ctx.run_encoder(mel_ref)
    enc_ref = orc.encoder(dims, b.w, mel_ref)
    np.testing.assert_allclose(enc, enc_ref, rtol=0, atol=ENC_TOL)
    prompt, eot = small_prompt(dims)
    # (a) free-running greedy
    ta, la = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 24, eot), want_logits=True)
    ra, rla = orc.decode_greedy(dims, b.w, enc_ref, prompt, 24, eot, want_logits=True)
    assert ta.tolist() == ra.tolist()
    np.testing.assert_allclose(la, rla, rtol=0, atol=LOGIT_TOL)
    # (b) suppress + begin-suppress (src/main.rs:765-768): forbid the free-running choices
    sup = [eot, int(ra[len(prompt)])]
    bsup = [int(ra[len(prompt) + 1])]
    tb, lb = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 24, eot, sup, bsup), want_logits=True)
    rb, rlb = orc.decode_greedy(dims, b.w, enc_ref, prompt, 24, eot, sup, bsup, want_logits=True)
    assert tb.tolist() == rb.tolist()
    assert sup[1] not in tb[len(prompt):].tolist() and tb[len(prompt)] != bsup[0]
    np.testing.assert_allclose(lb, rlb, rtol=0, atol=LOGIT_TOL)
    # (c) teacher forced random history
    rng = np.random.Generator(np.random.PCG64(99))
    forced = rng.integers(0, dims.vocab, size=40).tolist()
    tc, lc = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 41, eot, forced=forced), want_logits=True)
    rc, rlc = orc.decode_greedy(dims, b.w, enc_ref, prompt, 41, eot, forced=forced, want_logits=True)
    assert tc.tolist() == rc.tolist() and len(tc) == len(prompt) + 41
    np.testing.assert_allclose(lc, rlc, rtol=0, atol=LOGIT_TOL)


@pytest.mark.parametrize("prec_name", EXACT_MODES)
def test_f32_matches_golden_vectors(gpu, golden_dir, prec_name):
    for name in ("nano_s7_c1.npz", "micro_s11_c2.npz"):
        g = np.load(os.path.join(golden_dir, name))
        b = bundle(str(g["preset"]), int(g["seed"]), wb.PRECISIONS[prec_name])
        pcm = ms.synth_clip(int(g["clip"]))
        mel = b.ctx.whisper_log_mel(pcm)
        np.testing.assert_allclose(mel[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)
        enc = b.ctx.run_encoder(mel)
        np.testing.assert_allclose(enc[g["enc_rows"]], g["enc_slice"], rtol=0, atol=ENC_TOL)
        prompt, eot, mx = g["prompt"].tolist(), int(g["eot"]), int(g["max_new"])
        ta, la = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, mx, eot), want_logits=True)
        assert ta.tolist() == g["tokens_a"].tolist()
        for i in range(len(la)):
            np.testing.assert_allclose(la[i][g["top_ids_a"][i]], g["top_vals_a"][i], rtol=0, atol=LOGIT_TOL)
        tb, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, mx, eot, g["suppress_b"].tolist(),
                                                               g["begin_suppress_b"].tolist()))
        assert tb.tolist() == g["tokens_b"].tolist()
        forced = g["forced_c"].tolist()
        tc, lc = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), want_logits=True)
        assert tc.tolist() == g["tokens_c"].tolist()
        for i in range(len(lc)):
            np.testing.assert_allclose(lc[i][g["top_ids_c"][i]], g["top_vals_c"][i], rtol=0, atol=LOGIT_TOL)


@pytest.mark.parametrize("prec_name", EXACT_MODES)
def test_f32_whisper_base_matches_golden(gpu, golden_dir, prec_name):
    """whisper-base dims, hash-seeded weights: token-for-token + logits within 1e-3 (configs[1])."""
    g = np.load(os.path.join(golden_dir, "base_s1234_c0.npz"))
    b = bundle("base", 1234, wb.PRECISIONS[prec_name])
    pcm = ms.synth_clip(0)
    mel = b.ctx.whisper_log_mel(pcm)
    np.testing.assert_allclose(mel[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)
    enc = b.ctx.run_encoder(mel)
    np.testing.assert_allclose(enc[g["enc_rows"]], g["enc_slice"], rtol=0, atol=ENC_TOL)
    np.testing.assert_allclose(enc.astype(np.float64).mean(0), g["enc_col_mean"], rtol=0, atol=ENC_TOL)
    prompt, eot = g["prompt"].tolist(), int(g["eot"])
    ta, la = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 128, eot), want_logits=True)
    assert ta.tolist() == g["tokens_a"].tolist()
    for i in range(len(la)):
        np.testing.assert_allclose(la[i][g["top_ids_a"][i]], g["top_vals_a"][i], rtol=0, atol=LOGIT_TOL)
    tb, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 128, eot, g["suppress_b"].tolist(), g["begin_suppress_b"].tolist()))
    assert tb.tolist() == g["tokens_b"].tolist()
    forced = g["forced_c"].tolist()
    tc, lc = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), want_logits=True)
    assert tc.tolist() == g["tokens_c"].tolist()
    for i in range(len(lc)):
        np.testing.assert_allclose(lc[i][g["top_ids_c"][i]], g["top_vals_c"][i], rtol=0, atol=LOGIT_TOL)
    np.testing.assert_allclose(lc[:4, :2048], g["logits_c_head"], rtol=0, atol=LOGIT_TOL)
    worst = max(float(np.abs(la[i][g["top_ids_a"][i]] - g["top_vals_a"][i]).max()) for i in range(len(la)))
    print(f"{prec_name} base, one clip: max |logit - golden| over 128 free-running rows {worst:.2e}; encoder max |d| {np.abs(enc[g['enc_rows']] - g['enc_slice']).max():.2e}")


# ------------------------------------------------------------------------------------------------
# fused batch entry, ragged clips, long-form
# ------------------------------------------------------------------------------------------------
def test_batch_equals_staged_calls_and_oracle(gpu):
    b = bundle("nano", 7, wb.WH_PREC_F32, max_batch=8)
    dims = b.dims
    prompt, eot = small_prompt(dims)
    lens = [480000, 480000, 300000, 16000, 161, 480000, 99999]
    clips = [ms.synth_clip(10 + i)[:n] for i, n in enumerate(lens)]
    params = wb.DecodeParams(prompt, 12, eot)
    got = b.ctx.transcribe_batch(clips, params)
    assert len(got) == len(clips)
    for pcm, toks in zip(clips, got):
        mel_full = orc.log_mel(pcm, dims.n_mels)
        mel = orc.window_mel(mel_full, 0, 3000)       # zero padding in normalised space (:899-905)
        enc = orc.encoder(dims, b.w, mel)
        ref, _ = orc.decode_greedy(dims, b.w, enc, prompt, 12, eot)
        assert toks.tolist() == ref.tolist()
    # single-clip batch == same clip inside a bigger batch
    one = b.ctx.transcribe_batch([clips[2]], params)
    assert one[0].tolist() == got[2].tolist()


def test_eot_stops_each_clip_independently(gpu):
    b = bundle("nano", 7, wb.WH_PREC_F32, max_batch=8)
    dims = b.dims
    prompt, _ = small_prompt(dims)
    clips = [ms.synth_clip(30 + i) for i in range(4)]
    free = b.ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 20, dims.vocab - 1))
    # choose as EOT the token clip 0 emits at generated index 3: clip 0 must stop there (EOT kept)
    eot = int(free[0][len(prompt) + 3])
    got = b.ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 20, eot))
    for f, t in zip(free, got):
        gen = f[len(prompt):].tolist()
        cut = gen.index(eot) + 1 if eot in gen else len(gen)
        assert t.tolist() == f[: len(prompt) + cut].tolist()
    assert len(got[0]) <= len(prompt) + 4 and got[0][-1] == eot


def test_longform_windows_match_oracle(gpu):
    b = bundle("nano", 7, wb.WH_PREC_F32, max_batch=8)
    dims = b.dims
    prompt, eot = small_prompt(dims)
    pcm = np.concatenate([ms.synth_clip(40), ms.synth_clip(41), ms.synth_clip(42)[:200000]])  # 72.5 s
    params = wb.DecodeParams(prompt, 8, eot)
    got = b.ctx.transcribe_longform(pcm, params)
    offs = wb.longform_plan(pcm.size)
    assert len(got) == len(offs) == 3
    mel_full = orc.log_mel(pcm, dims.n_mels)           # whole-file mel, global max (:870-872)
    for off, toks in zip(offs, got):
        mel = orc.window_mel(mel_full, off // 160, 3000)
        enc = orc.encoder(dims, b.w, mel)
        ref, _ = orc.decode_greedy(dims, b.w, enc, prompt, 8, eot)
        assert toks.tolist() == ref.tolist()


def test_decode_before_encode_is_state_error(gpu):
    m = wb.Model("synthetic:nano:7", 0, wb.WH_PREC_F32)
    c = wb.Context(m, 1)
    with pytest.raises(wb.WhisperHipError) as ei:
        c.greedy_decode_with_past(wb.DecodeParams([3, 5, 7, 9], 4, 2))
    assert ei.value.code == 3 and "Missing cached decoder input" in str(ei.value)   # src/main.rs:808-810
    with pytest.raises(wb.WhisperHipError) as ei:
        c.run_encoder(np.zeros((80, 2999), np.float32))
    assert ei.value.code == 2
    with pytest.raises(wb.WhisperHipError) as ei:
        c.run_encoder(np.zeros((80, 3000), np.float32), want_output=False)
        c.greedy_decode_with_past(wb.DecodeParams([3, 5, 7, 9], 500, 2))
    assert ei.value.code == 4


def test_argmax_edge_cases_via_suppress_all_but_one(gpu):
    """Masked argmax semantics (src/main.rs:709-735) through the LM-head kernel: with every id but one
    suppressed the survivor is chosen; with everything suppressed the answer is id 0."""
    b = bundle("nano", 7, wb.WH_PREC_F32)
    dims = b.dims
    prompt, _ = small_prompt(dims)
    b.ctx.run_encoder(orc.log_mel(ms.synth_clip(0), 80), want_output=False)
    keep = 777
    sup = [i for i in range(dims.vocab) if i != keep]
    t, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 3, dims.vocab + 5, sup))
    assert t[len(prompt):].tolist() == [keep] * 3
    t, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 2, dims.vocab + 5, list(range(dims.vocab))))
    assert t[len(prompt):].tolist() == [0, 0]
    # begin-suppress applies to the first generated token only
    t0, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 2, dims.vocab + 5))
    t1, _ = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 2, dims.vocab + 5, (), [int(t0[len(prompt)])]))
    assert t1[len(prompt)] != t0[len(prompt)]


# ------------------------------------------------------------------------------------------------
# bf16 mode
# ------------------------------------------------------------------------------------------------
# bf16 against the f32 golden vectors: ABSOLUTE bounds (round-3 verdict: a bound of 2 x the measured error moves with what it bounds).
# Measured on MI355X with the encoder LayerNorm fold: encoder states 0.070 (micro) / 0.081 (base), logits 0.080 (micro) / 0.132 (base) on
# logits of sigma 1.3; a token is "decided" where the f32 top-1 margin exceeds twice the logit bound — there the argmax must agree.
BF16_ENC_BOUND, BF16_LOGIT_BOUND = 0.10, 0.16
BF16_MIN_DECIDED = {"micro": 4, "base": 3}   # measured 5 and 4   # steps of the 24-step golden history whose f32 top-1 margin exceeds 2 x the logit bound


@pytest.mark.parametrize("preset,seed,clip", [("micro", 11, 2), ("base", 1234, 0)])
def test_bf16_teacher_forced_agreement(gpu, golden_dir, preset, seed, clip):
    """bf16 MFMA path vs the fp32 golden vectors under teacher forcing: encoder and logit errors inside fixed bounds, the argmax
    agrees on every step whose fp32 top-1 margin exceeds twice the logit bound, and the golden history has such steps."""
    g = np.load(os.path.join(golden_dir, f"{preset}_s{seed}_c{clip}.npz"))
    b = bundle(preset, seed, wb.WH_PREC_BF16)
    pcm = ms.synth_clip(clip)
    mel = b.ctx.whisper_log_mel(pcm)
    np.testing.assert_allclose(mel[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)  # mel is f64/f32 in both modes
    enc = b.ctx.run_encoder(mel)
    enc_err = np.abs(enc[g["enc_rows"]] - g["enc_slice"]).max()
    assert enc_err < BF16_ENC_BOUND, enc_err
    prompt, eot = g["prompt"].tolist(), int(g["eot"])
    forced = g["forced_c"].tolist()
    tc, lc = b.ctx.greedy_decode_with_past(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), want_logits=True)
    errs = [float(np.abs(lc[i][g["top_ids_c"][i]] - g["top_vals_c"][i]).max()) for i in range(len(lc))]
    decided = [i for i in range(len(lc)) if g["top_vals_c"][i][0] - g["top_vals_c"][i][1] > 2.0 * BF16_LOGIT_BOUND]
    agree = sum(int(tc[len(prompt) + i] == g["tokens_c"][len(prompt) + i]) for i in decided)
    print(f"{preset}: bf16 enc err {enc_err:.4f}; max logit err {max(errs):.4f}, mean {np.mean(errs):.4f}; decided {len(decided)}/{len(lc)} agree {agree}")
    assert max(errs) < BF16_LOGIT_BOUND
    assert agree == len(decided)
    assert len(decided) >= BF16_MIN_DECIDED[preset], (len(decided), "the golden history must keep steps with a clear f32 winner")


def test_gemm8_off_switch_keeps_the_result_or_fails_loudly(gpu, golden_dir, monkeypatch):
    """WH_GEMM8=0 (the documented A/B switch) on a bf16 context above the small-context size: every encoder GEMM goes to k_gemm, which has no
    LayerNorm fold, so the context must not take the fold path (round-3 advisor: the launches were skipped and WH_OK returned).  The result is
    held to the same golden bound as the default path; a GEMM that cannot run now fails the call instead of being skipped."""
    g = np.load(os.path.join(golden_dir, "base_s1234_c0.npz"))
    monkeypatch.setenv("WH_GEMM8", "0")
    model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    ctx = wb.Context(model, 32)
    prompt, eot = g["prompt"].tolist(), int(g["eot"])
    forced = g["forced_c"].tolist()
    ctx.transcribe_batch([ms.synth_clip(0)] + [ms.synth_clip(310 + i) for i in range(31)], wb.DecodeParams(prompt, 2, eot, [eot]))
    _, lg = ctx.greedy_decode_resident_rows(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), [0])
    err = max(float(np.abs(lg[0][i][g["top_ids_c"][i]] - g["top_vals_c"][i]).max()) for i in range(len(forced) + 1))
    print(f"WH_GEMM8=0, bf16 base, 32-clip context: max |logit - golden| {err:.4f}")
    assert np.isfinite(lg[0]).all() and err < BF16_LOGIT_BOUND
    ctx.close()


def test_one_launch_feed_forward_block_against_the_two_gemm_form(gpu, golden_dir, monkeypatch):
    """k_enc_mlp (wh_mlp.hip: a layer's fc1 + GELU + fc2 + residual in one launch, hidden activations in LDS) against the two k_gemm8 launches it
    replaces (WH_ENC_MLP=0 at context creation): the same roundings in the same places (h to bf16, f32 residual stream), so the encoder output
    differs by accumulation order only — held far inside the bf16 bound — and both forms meet the golden logit bound.  A one-clip call on the
    32-clip context exercises the tile that is cut by the end of the rows (1500 = 11 x 128 + 92)."""
    g = np.load(os.path.join(golden_dir, "base_s1234_c0.npz"))
    prompt, eot = g["prompt"].tolist(), int(g["eot"])
    forced = g["forced_c"].tolist()
    model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    clips = [ms.synth_clip(0)] + [ms.synth_clip(310 + i) for i in range(31)]
    enc, lgs, toks = {}, {}, {}
    for form in ("1", "0"):
        monkeypatch.setenv("WH_ENC_MLP", form)
        ctx = wb.Context(model, 32)
        enc[form] = ctx.run_encoder(ctx.whisper_log_mel(clips[0]))
        toks[form] = [t.tolist() for t in ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 8, eot, [eot]))]
        _, lg = ctx.greedy_decode_resident_rows(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), [0])
        lgs[form] = lg[0]
        ctx.close()
    d_enc = float(np.abs(enc["1"] - enc["0"]).max())
    err = {f: max(float(np.abs(lgs[f][i][g["top_ids_c"][i]] - g["top_vals_c"][i]).max()) for i in range(len(forced) + 1)) for f in lgs}
    same = sum(a == b for a, b in zip(toks["1"], toks["0"]))
    print(f"one-launch feed-forward block vs two GEMMs, bf16 base, 32-clip context: max |encoder out diff| {d_enc:.2e} (encoder outputs up to "
          f"{float(np.abs(enc['0']).max()):.1f}); max |logit - golden| {err['1']:.4f} vs {err['0']:.4f}; {same} / 32 clips with identical 8-token decodes")
    assert np.isfinite(enc["1"]).all() and d_enc < 0.25 * BF16_ENC_BOUND
    assert err["1"] < BF16_LOGIT_BOUND and err["0"] < BF16_LOGIT_BOUND


def test_feed_forward_kernel_matches_host_restatement(gpu):
    """tools/mlp_check: k_enc_mlp alone against a double-precision host restatement (fold, erf GELU, h rounded to bf16, second product, bias,
    residual; the bf16 copy and the LayerNorm partial sums of the new rows) at 128 / 300 / 1500 / 13500 rows — whole tiles, a cut tile, one
    clip, nine clips."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "mlp_check")
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() compiles it"
    r = subprocess.run([exe, "16"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr


def test_workspace_placement_step(gpu, monkeypatch):
    """wh_ctx_create_ex looks where the workspace of a large encoder-state context lies: it times the cross-attention kernel on the fresh workspace
    and, when that reads slow, builds a second one beside it and keeps the faster.  Whatever it finds, the context it returns
    decodes like any other; WH_PLACE=0 skips the step; small contexts never take it."""
    model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    prompt, eot = [50258, 50259, 50359, 50363], 50257
    clips = [ms.synth_clip(500 + i) for i in range(4)]
    small = wb.Context(model, 4)
    ref = [t.tolist() for t in small.transcribe_batch(clips, wb.DecodeParams(prompt, 6, eot, [eot]))]
    assert small.placement["workspaces_timed"] == 0
    small.close()
    monkeypatch.setenv("WH_PLACE_FRAC", "2.0")          # no workspace can read that fast: every allowed try is made and the fastest kept
    big = wb.Context(model, 1024)
    pl = big.placement
    print("placement step, 1024-clip bf16 context, every try forced:", pl)
    assert pl["workspaces_timed"] == 3 and 0.0 < pl["kept_us_per_launch"] <= pl["first_us_per_launch"]
    # 1024 clips x 1520 rows x 1 KB per launch: between a third of the HBM roof and the roof
    gbps = 1024 * 1520 * 1024 / pl["kept_us_per_launch"] * 1e-3
    assert 2500.0 < gbps < 8000.0, gbps
    got = [t.tolist() for t in big.transcribe_batch(clips, wb.DecodeParams(prompt, 6, eot, [eot]))]
    big.close()
    # more tries than workspaces fit: the search ends at the allocation that fails, and that failure must not surface later
    # (a sticky HIP error found by the next hipGetLastError() made the first transcribe call of such a context fail)
    monkeypatch.setenv("WH_PLACE_TRIES", "64")
    many = wb.Context(model, 1024)
    pl = many.placement
    print("placement step, tries until the memory is full:", pl)
    assert 3 <= pl["workspaces_timed"] < 64
    assert [t.tolist() for t in many.transcribe_batch(clips, wb.DecodeParams(prompt, 6, eot, [eot]))] == got
    many.close()
    monkeypatch.delenv("WH_PLACE_TRIES")
    monkeypatch.setenv("WH_PLACE", "0")
    off = wb.Context(model, 1024)
    assert off.placement["workspaces_timed"] == 0
    off.close()
    # bf16 rows of a 4-clip context (k_gemm small-context kernels, no fold) and of a 1024-clip one differ in rounding: compare the prompt echo and lengths only
    assert all(g[:4] == r[:4] and len(g) == len(r) for g, r in zip(got, ref))


def test_split_fp16_encoder_state_kernels(gpu, golden_dir):
    """The split-fp16 mode's cross-attention streams the encoder states as fp16 + an e4m3 remainder (k_dec_cross_attn_es3, 3 bytes per element; every
    f16x3 test of this file at 256 clips and more runs it).  Here: the kernel alone against a host restatement (tools/es3_check), and the form it
    replaced — two fp16 limbs, k_dec_cross_attn_es2, WH_ES3=0 — still held to the golden vectors, in a process of its own (the switch is read once)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "es3_check")
    assert os.path.exists(exe), f"{exe} missing: __graft_entry__.build() compiles it"
    r = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout + r.stderr
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from whisper_rust_ort_amd import binding as wb, modelspec as ms\n"
        f"g = np.load(os.path.join({golden_dir!r}, 'base_s1234_c0.npz'))\n"
        "prompt, eot, forced = g['prompt'].tolist(), int(g['eot']), g['forced_c'].tolist()\n"
        "m = wb.Model('synthetic:base:1234', 0, wb.WH_PREC_F16X3)\n"
        "c = wb.Context(m, 256)\n"
        "c.transcribe_batch([ms.synth_clip(0)] + [ms.synth_clip(300 + i) for i in range(255)], wb.DecodeParams(prompt, 2, eot, [eot]))\n"
        "_, lg = c.greedy_decode_resident_rows(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), [0])\n"
        "err = max(float(np.abs(lg[0][i][g['top_ids_c'][i]] - g['top_vals_c'][i]).max()) for i in range(len(forced) + 1))\n"
        "print('es2 form: max |logit - golden|', err)\n"
        "assert err < 1e-3, err\n")
    env = dict(os.environ, WH_ES3="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr[-400:])
    assert r.returncode == 0, r.stdout + r.stderr


def test_bf16_batch_is_deterministic_and_permutation_invariant(gpu):
    b = bundle("micro", 11, wb.WH_PREC_BF16, max_batch=16)
    prompt, eot = small_prompt(b.dims)
    clips = [ms.synth_clip(60 + i) for i in range(16)]
    params = wb.DecodeParams(prompt, 16, eot, [eot])
    a = b.ctx.transcribe_batch(clips, params)
    a2 = b.ctx.transcribe_batch(clips, params)
    assert [t.tolist() for t in a] == [t.tolist() for t in a2]
    perm = np.random.Generator(np.random.PCG64(5)).permutation(16)
    c = b.ctx.transcribe_batch([clips[i] for i in perm], params)
    for j, i in enumerate(perm):
        assert c[j].tolist() == a[i].tolist()
    assert all(len(t) == len(prompt) + 16 for t in a)   # EOT suppressed → full length


# ------------------------------------------------------------------------------------------------
# full-size properties (BASELINE configs[2] shard: whisper-base, bf16, 64 clips per batch, 128 tokens)
# ------------------------------------------------------------------------------------------------
def test_base_bf16_full_batch_properties(gpu):
    """At the benchmark's own size the oracle is too slow to be the checker, so size-independent properties
    are: the 64-clip batch equals the same clips run in two 32-clip batches and one by one (clips are
    independent units), duplicates give identical rows, the run is deterministic, and with EOT suppressed
    every clip emits exactly max_new_tokens."""
    b64 = bundle("base", 1234, wb.WH_PREC_BF16, max_batch=64)
    prompt, eot = small_prompt(b64.dims)
    clips = [ms.synth_clip(100 + i) for i in range(62)] + [ms.synth_clip(100), ms.synth_clip(101)]   # two duplicates
    params = wb.DecodeParams(prompt, 128, eot, [eot])
    full = b64.ctx.transcribe_batch(clips, params)
    again = b64.ctx.transcribe_batch(clips, params)
    assert [t.tolist() for t in full] == [t.tolist() for t in again]
    assert all(len(t) == len(prompt) + 128 for t in full) and all(eot not in t[len(prompt):] for t in full)
    assert full[62].tolist() == full[0].tolist() and full[63].tolist() == full[1].tolist()
    halves = b64.ctx.transcribe_batch(clips[:32], params) + b64.ctx.transcribe_batch(clips[32:], params)
    assert [t.tolist() for t in halves] == [t.tolist() for t in full]
    for i in (0, 17, 63):
        assert b64.ctx.transcribe_batch([clips[i]], params)[0].tolist() == full[i].tolist()
    # free-running (EOT allowed): each row is a prefix-consistent cut of the suppressed run up to its EOT
    free = b64.ctx.transcribe_batch(clips[:8], wb.DecodeParams(prompt, 128, eot))
    for t in free:
        assert len(t) <= len(prompt) + 128 and (eot not in t[len(prompt):-1].tolist())


def test_launch_modes_agree(gpu):
    """The token loop has three launch modes — hipGraph replay (default), eager with per-launch events
    (profile on), and sampled (graph replay with every stride-th position eager + timed).  They launch
    the same kernels on the same device state, so token ids must be identical; the sampled mode reports
    only the launches it bracketed."""
    b = bundle("nano", 7, wb.WH_PREC_F32, max_batch=8)
    prompt, eot = small_prompt(b.dims)
    clips = [ms.synth_clip(40 + i) for i in range(3)]
    params = wb.DecodeParams(prompt, 24, eot, [eot])
    b.ctx.profile_enable(False)
    ref = [t.tolist() for t in b.ctx.transcribe_batch(clips, params)]
    b.ctx.profile_enable(True)
    eager = [t.tolist() for t in b.ctx.transcribe_batch(clips, params)]
    p_all = b.ctx.profile_get()
    b.ctx.profile_enable(["dec_cross_attn"], stride=4)
    sampled = [t.tolist() for t in b.ctx.transcribe_batch(clips, params)]
    p_smp = b.ctx.profile_get()
    b.ctx.profile_enable(False)
    assert eager == ref and sampled == ref
    n_layers = b.dims.dec_layers
    steps = len(prompt) + 24 - 1                      # decoder positions run (the last token needs no step)
    assert p_all["dec_cross_attn"]["launches"] == n_layers * steps
    # prompt positions are always eager; of the remaining graph positions r = 0..R-1 those with r % 4 == 2
    rem = steps - len(prompt)
    assert p_smp["dec_cross_attn"]["launches"] == n_layers * (len(prompt) + len([r for r in range(rem) if r % 4 == 2]))
    assert p_smp.get("dec_self_attn", {"launches": 0})["launches"] == 0 and p_smp["dec_cross_attn"]["ms"] > 0


def _ctx_logit_compare(prec, golden_dir, label, big=256):
    """Teacher-forced logits of the same 32 clips decoded on a 256-clip (or, big=1024, the largest) context (one key range per clip, the attention
    kernel writes its own output, non-temporal K/V stream, 64-row GEMM groups) and on a 64-clip context (four key ranges
    merged in the out-projection GEMM), both against each other and against the f32 golden vectors of clips 0 and 3.
    With the token history forced nothing accumulates: what remains is each configuration's own rounding."""
    g0 = np.load(os.path.join(golden_dir, "base_s1234_c0.npz"))
    g3 = np.load(os.path.join(golden_dir, "base_s1234_c3.npz"))
    prompt, eot = g0["prompt"].tolist(), int(g0["eot"])
    forced = g0["forced_c"].tolist()                             # the whole 23-step forced history at every context size
    distinct = [ms.synth_clip(0), ms.synth_clip(3)] + [ms.synth_clip(300 + i) for i in range(30)]
    b256 = bundle("base", 1234, prec, max_batch=big)
    b64 = bundle("base", 1234, prec, max_batch=64)
    fp = wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced)
    b256.ctx.transcribe_batch([distinct[i % 32] for i in range(big)], wb.DecodeParams(prompt, 2, eot, [eot]))
    if big > 256:
        # wh_decode_greedy_rows: logits of the 32 distinct clips and of a few of their duplicates only (all rows would be 10 GB at 2048)
        dup = [32, 33, big // 2 + 7, big - 32, big - 1]
        t256, lsel = b256.ctx.greedy_decode_resident_rows(fp, list(range(32)) + dup)
        l256 = np.stack(lsel[:32])
        for j, r in enumerate(dup):
            assert np.array_equal(lsel[32 + j], l256[r % 32]), r
        for i in range(32, big):                                 # duplicates inside one context: identical tokens everywhere
            assert t256[i].tolist() == t256[i % 32].tolist(), i
    else:
        t256, l256 = b256.ctx.greedy_decode_resident_batch(fp, want_logits=True)
        l256 = np.stack(l256)                                    # [big][rows][V]
        for i in range(32, big):                                 # duplicates inside one context: identical arithmetic
            assert np.array_equal(l256[i], l256[i % 32]), i
            assert t256[i].tolist() == t256[i % 32].tolist()
    b64.ctx.transcribe_batch(distinct, wb.DecodeParams(prompt, 2, eot, [eot]))
    t64, l64 = b64.ctx.greedy_decode_resident_batch(fp, want_logits=True)
    l64 = np.stack(l64)                                          # [32][24][V]
    d_ctx = np.abs(l256[:32] - l64).max(axis=(1, 2))             # per clip
    # against the f32 golden vectors (clip 0 under its own forced history; clip 3's golden used another history, so
    # only its first row — the prompt step, history-free — is comparable)
    e256 = max(np.abs(l256[0][i][g0["top_ids_c"][i]] - g0["top_vals_c"][i]).max() for i in range(len(forced) + 1))
    e64 = max(np.abs(l64[0][i][g0["top_ids_c"][i]] - g0["top_vals_c"][i]).max() for i in range(len(forced) + 1))
    e3 = max(np.abs(l256[1][0][g3["top_ids_c"][0]] - g3["top_vals_c"][0]).max(),
             np.abs(l64[1][0][g3["top_ids_c"][0]] - g3["top_vals_c"][0]).max())
    print(f"{label}: {big}-clip vs 64-clip context, teacher-forced: max |dlogit| per clip: max {d_ctx.max():.4f} median {np.median(d_ctx):.4f}; "
          f"vs f32 golden: 256-ctx {e256:.4f}, 64-ctx {e64:.4f}, clip 3 row 0 {e3:.4f}; logit std {l64.std():.3f}")
    # argmax agreement between the contexts wherever the top-1 margin exceeds twice their measured difference
    srt = np.sort(l64, axis=2)
    margin = srt[:, :, -1] - srt[:, :, -2]
    a256 = l256[:32].argmax(axis=2)
    a64 = l64.argmax(axis=2)
    decided = margin > 2.0 * d_ctx[:, None]
    assert (a256[decided] == a64[decided]).all()
    print(f"{label}: argmax agrees on all {int(decided.sum())} of {decided.size} decided positions; "
          f"{int((a256 != a64).sum())} undecided positions differ")
    return d_ctx, e256, e64, e3


@pytest.mark.parametrize("big", [256, 1024, 2048])
def test_base_bf16_256_vs_64_clip_context_logit_bound(gpu, golden_dir, big):
    """Replaces the former 'at most 4 of 32 clips may diverge' allowance by a measured, per-row logit bound."""
    d_ctx, e256, e64, e3 = _ctx_logit_compare(wb.WH_PREC_BF16, golden_dir, "bf16", big)
    assert d_ctx.max() < 0.12          # two bf16 summation orders / two forms of the cross-attention on logits of sigma 1.3 (measured 0.06-0.08)
    assert max(e256, e64, e3) < BF16_LOGIT_BOUND   # the bf16-vs-f32 bound of test_bf16_teacher_forced_agreement, over the whole forced history


# ------------------------------------------------------------------------------------------------
# the batched kernel variants bench.py times, against the f32 golden vectors (HF-pinned)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec_name", EXACT_MODES)
@pytest.mark.parametrize("nb", [64, 256, 1024, 2048])
def test_f32_base_batched_contexts_match_golden(gpu, golden_dir, nb, prec_name):
    """whisper-base dims, exact-f32 mode and the split-fp16 mode, 64-, 256-, 1024- and 2048-clip contexts (cross_splits 4 / 1, merged vs direct attention
    output, non-temporal K/V loads from 256 up, the row-group variants of the decode GEMMs and the LM head; 2048 = the
    library's largest batch and bench.py's per-step workload): golden clips 0 and
    3 sit at several batch rows among filler clips.  Tokens identical to the golden free-running streams, top-k logits
    within 1e-3 — the same bar as the one-clip path."""
    g = {0: np.load(os.path.join(golden_dir, "base_s1234_c0.npz")), 3: np.load(os.path.join(golden_dir, "base_s1234_c3.npz"))}
    b = bundle("base", 1234, wb.PRECISIONS[prec_name], max_batch=nb)
    prompt, eot = g[0]["prompt"].tolist(), int(g[0]["eot"])
    rows = {0: 0, 1: 3, 17: 0, nb // 2: 3, nb - 2: 3, nb - 1: 0}
    pcm = {0: ms.synth_clip(0), 3: ms.synth_clip(3)}
    clips = [pcm[rows[i]] if i in rows else ms.synth_clip(500 + (i % 23)) for i in range(nb)]
    # (a) free-running greedy, 128 new tokens, through the throughput entry
    got = b.ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 128, eot))
    for r, c in rows.items():
        assert got[r].tolist() == g[c]["tokens_a"].tolist(), (r, c)
    # (b) the same with suppress sets (src/main.rs:765-768) — clip 0's sets
    gb = b.ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 128, eot, g[0]["suppress_b"].tolist(), g[0]["begin_suppress_b"].tolist()))
    for r, c in rows.items():
        if c == 0:
            assert gb[r].tolist() == g[0]["tokens_b"].tolist(), r
    # (c) logits of the batched decode, read back for the golden rows only (wh_decode_greedy_rows: every context size sees the same
    # 32 free-running steps and the whole teacher-forced history): first 32 free-running rows, then each golden clip's forced history
    sel = sorted(rows)
    ta, la = b.ctx.greedy_decode_resident_rows(wb.DecodeParams(prompt, 32, eot), sel)
    worst = 0.0
    for j, r in enumerate(sel):
        c, n = rows[r], len(la[j])
        assert n == 32 and ta[r].tolist() == g[c]["tokens_a"][: len(prompt) + n].tolist()
        for i in range(n):
            worst = max(worst, float(np.abs(la[j][i][g[c]["top_ids_a"][i]] - g[c]["top_vals_a"][i]).max()))
    for c in (0, 3):
        forced = g[c]["forced_c"].tolist()
        tc, lc = b.ctx.greedy_decode_resident_rows(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), sel)
        for j, r in enumerate(sel):
            if rows[r] != c:
                continue
            assert tc[r].tolist() == g[c]["tokens_c"].tolist(), (r, c)
            for i in range(len(lc[j])):
                worst = max(worst, float(np.abs(lc[j][i][g[c]["top_ids_c"][i]] - g[c]["top_vals_c"][i]).max()))
            np.testing.assert_allclose(lc[j][:4, :2048], g[c]["logits_c_head"], rtol=0, atol=LOGIT_TOL)
    print(f"{prec_name} base, {nb}-clip context: max |logit - golden| over all compared rows {worst:.2e}")
    assert worst <= LOGIT_TOL


@pytest.mark.parametrize("prec_name", ["bf16", "fp8", "f16x3"])
def test_wide_batch_decode_gemm_is_bit_identical(gpu, monkeypatch, prec_name):
    """k_dec_gemm_wide (several 16-column tiles per workgroup) and k_lm_head_tile (the LM head on 256 x 256 LDS-DMA tiles), both
    chosen from the batch size, against k_dec_gemm / k_lm_head on the same 512-clip context: same K split, same summation
    order, same epilogue arithmetic — tokens and every logit bit-identical, whichever kernels run."""
    prec = wb.PRECISIONS[prec_name]
    model = wb.Model("synthetic:base:1234", 0, prec)
    prompt, eot = [50258, 50259, 50359, 50363], 50257
    clips = [ms.synth_clip(700 + i) for i in range(8)]
    forced = np.random.Generator(np.random.PCG64(11)).integers(0, 50257, size=5).tolist()
    res = {}
    modes = ("0", "-1", "2", "4", "lm")   # k_dec_gemm + k_lm_head only | heuristics (wide GEMM, tile LM head) | NT = 2 / 4 forced | only the LM head switched
    for wide in modes:
        monkeypatch.setenv("WH_DEC_WIDE", "0" if wide == "lm" else wide)
        monkeypatch.setenv("WH_LM_TILE_MIN_ROWS", "0" if wide == "0" else "256")
        ctx = wb.Context(model, 512)
        toks = [t.tolist() for t in ctx.transcribe_batch([clips[i % 8] for i in range(512)], wb.DecodeParams(prompt, 24, eot, [eot]))]
        _, lg = ctx.greedy_decode_resident_batch(wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced), want_logits=True)
        res[wide] = (toks, np.stack(lg))
        ctx.close()
    for wide in modes[1:]:
        assert res[wide][0] == res["0"][0], wide
        assert np.array_equal(res[wide][1], res["0"][1]), wide


@pytest.mark.parametrize("prec_name", ["bf16", "fp8"])
def test_largest_batch_rows_equal_small_calls_on_the_same_context(gpu, prec_name):
    """One 1024-clip context: a full batch (wide decode GEMM tiles, one key range per clip, 64-row LM-head groups) against
    the same clips submitted alone, in a batch of 3 and in a batch of 100 (k_dec_gemm, other row-group shapes) — a clip's
    tokens never depend on what shares its batch."""
    b = bundle("base", 1234, wb.PRECISIONS[prec_name], max_batch=1024)
    prompt, eot = small_prompt(b.dims)
    uniq = [ms.synth_clip(900 + i) for i in range(24)]
    clips = [uniq[(i * 5) % 24] for i in range(1024)]
    p = wb.DecodeParams(prompt, 48, eot, [eot])
    full = [t.tolist() for t in b.ctx.transcribe_batch(clips, p)]
    for i in range(24, 1024):                       # duplicates inside the batch
        assert full[i] == full[i % 24], i
    for i in (0, 7, 1023):
        assert b.ctx.transcribe_batch([clips[i]], p)[0].tolist() == full[i]
    assert [t.tolist() for t in b.ctx.transcribe_batch(clips[5:8], p)] == full[5:8]
    assert [t.tolist() for t in b.ctx.transcribe_batch(clips[300:400], p)] == full[300:400]


# ------------------------------------------------------------------------------------------------
# the entry bench.py times (PCM resident in HBM) and its pipelined form
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset,seed,prec_name,nb", [("micro", 11, "f32", 6), ("base", 1234, "bf16", 64), ("base", 1234, "bf16", 1024)])
def test_device_entry_equals_host_entry(gpu, preset, seed, prec_name, nb):
    """wh_transcribe_batch_device (device-resident [n][480000] f32, every clip exactly 30 s — what bench.py times) returns
    exactly what wh_transcribe_batch returns for the same clips from host memory, and the f32 case is held to the oracle.
    Covers the device entry's own code: no H2D, the constant sample / frame counts, the caller's PCM pointer."""
    b = bundle(preset, seed, wb.PRECISIONS[prec_name], max_batch=nb)
    prompt, eot = small_prompt(b.dims)
    uniq = [ms.synth_clip(1200 + i) for i in range(min(nb, 16))]
    clips = [uniq[(i * 7) % len(uniq)] for i in range(nb)]
    p = wb.DecodeParams(prompt, 12, eot, [eot])
    host = [t.tolist() for t in b.ctx.transcribe_batch(clips, p)]
    hip = wb.HipRuntime()
    d_pcm = hip.upload(0, np.stack(clips))
    try:
        dev = [t.tolist() for t in b.ctx.transcribe_batch_device(d_pcm, nb, p)]
        assert dev == host
        # a sub-batch starting in the middle of the device array
        k = nb // 2
        assert [t.tolist() for t in b.ctx.transcribe_batch_device(d_pcm + k * 480000 * 4, nb - k, p)] == host[k:]
    finally:
        hip.free(d_pcm)
    if prec_name == "f32":
        for i in (0, nb - 1):
            mel = orc.window_mel(orc.log_mel(clips[i], b.dims.n_mels), 0, 3000)
            ref, _ = orc.decode_greedy(b.dims, b.w, orc.encoder(b.dims, b.w, mel), prompt, 12, eot, [eot])
            assert host[i] == ref.tolist(), i


@pytest.mark.parametrize("masks", ["two_streams", "cu_masks"])
def test_pipelined_device_entry_equals_plain(gpu, masks):
    """wh_transcribe_batch_device_next on a two-stream context (with and without CU masks): batch A with B's encoder pass
    prefetched beside A's token loop, then B from the prefetched states, then A again without a prefetch — identical to the
    one-stream context's results for A and B; a call for another batch than the prefetched one recomputes; stage timings
    stay positive."""
    b = bundle("base", 1234, wb.WH_PREC_BF16, max_batch=64)
    prompt, eot = small_prompt(b.dims)
    p = wb.DecodeParams(prompt, 24, eot, [eot])
    A = [ms.synth_clip(1300 + i) for i in range(64)]
    B = [ms.synth_clip(1400 + i) for i in range(48)]
    ref_a = [t.tolist() for t in b.ctx.transcribe_batch(A, p)]
    ref_b = [t.tolist() for t in b.ctx.transcribe_batch(B, p)]
    if masks == "cu_masks":
        ctx = wb.Context(b.model, 64, enc_cu_mask=wb.cu_mask(0, 64), dec_cu_mask=wb.cu_mask(64, 192))
    else:
        ctx = wb.Context(b.model, 64, two_streams=True)
    hip = wb.HipRuntime()
    da, db = hip.upload(0, np.stack(A)), hip.upload(0, np.stack(B))
    try:
        got_a = [t.tolist() for t in ctx.transcribe_batch_device(da, 64, p, next_ptr=db, next_n=48)]
        got_b = [t.tolist() for t in ctx.transcribe_batch_device(db, 48, p, next_ptr=da, next_n=64)]   # prefetched; prefetches A
        tm = ctx.timings()
        got_a2 = [t.tolist() for t in ctx.transcribe_batch_device(da, 64, p)]                         # prefetched, no further prefetch
        assert got_a == ref_a and got_b == ref_b and got_a2 == ref_a
        assert tm["preprocess_s"] > 0 and tm["encode_s"] > 0 and tm["decode_s"] > 0
        # prefetch B, then ask for A's second half instead: the prefetched states are not used, the result is A's
        ctx.transcribe_batch_device(da, 64, p, next_ptr=db, next_n=48)
        assert [t.tolist() for t in ctx.transcribe_batch_device(da + 32 * 480000 * 4, 32, p)] == ref_a[32:]
        # the staged API and the host entry still work on a two-stream context
        assert [t.tolist() for t in ctx.transcribe_batch(B, p)] == ref_b
        mel = ctx.whisper_log_mel(A[0])
        ctx.run_encoder(mel)
        one, _ = ctx.greedy_decode_with_past(p)
        assert one.tolist() == ref_a[0]
    finally:
        ctx.close()
        hip.free(da)
        hip.free(db)


# ------------------------------------------------------------------------------------------------
# cross-attention on the encoder states (wh_cross_es.hip): the same attention as the projected K / V form
# (reference: present.{i}.encoder.{key,value} computed once per clip, src/main.rs:771-787, read by every token, :798-812)
# ------------------------------------------------------------------------------------------------
def test_cross_mode_rule_and_flags(gpu):
    """What a context's token loop streams is decided from the model and the context only: bf16 whisper-base geometry and
    max_batch >= 256 -> the encoder states; everything else -> the projected K / V cache.  The flags override the size rule;
    forcing the encoder-state form onto a model without the geometry is refused."""
    base16 = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    for mb, want in ((1, 0), (64, 0), (256, 1)):
        c = wb.Context(base16, mb)
        assert c.cross_mode == want, (mb, c.cross_mode)
        c.close()
    c = wb.Context(base16, 2, cross_es=True)
    assert c.cross_mode == 1
    c.close()
    c = wb.Context(base16, 256, cross_es=False)
    assert c.cross_mode == 0
    c.close()
    nano = wb.Model("synthetic:nano:7", 0, wb.WH_PREC_BF16)
    with pytest.raises(wb.WhisperHipError):
        wb.Context(nano, 4, cross_es=True)
    base32 = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_F32)
    c = wb.Context(base32, 256)
    assert c.cross_mode == 0        # the exact-f32 mode keeps the reference's K / V form
    c.close()
    basex3 = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_F16X3)
    for mb, want in ((64, 0), (256, 1)):   # the split-fp16 mode streams the states as fp16 limb planes (k_dec_cross_attn_es2)
        c = wb.Context(basex3, mb)
        assert c.cross_mode == want, (mb, c.cross_mode)
        c.close()
    base8 = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_FP8)
    for mb, want in ((64, 0), (256, 1)):   # the fp8 mode streams them as e4m3 rows (k_dec_cross_attn_es8)
        c = wb.Context(base8, mb)
        assert c.cross_mode == want, (mb, c.cross_mode)
        c.close()


@pytest.mark.parametrize("nb", [32, 288])
def test_cross_es_matches_projected_kv_and_golden(gpu, golden_dir, nb):
    """Teacher-forced logits of the same clips on two contexts of one bf16 whisper-base model — cross-attention on the encoder
    states vs on the projected K / V — against each other and against the f32 golden vectors (clip 0 at row 0).  32 clips: one
    clip per workgroup; 288: the persistent form, 32 workgroups walk two clips each (ring and query prefetch across a clip
    boundary), every clip repeated nine times must give identical rows.  The two forms are the same arithmetic up to bf16
    rounding: their difference stays below each one's distance from the f32 vectors."""
    g0 = np.load(os.path.join(golden_dir, "base_s1234_c0.npz"))
    prompt, eot = g0["prompt"].tolist(), int(g0["eot"])
    forced = g0["forced_c"].tolist()
    model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    distinct = [ms.synth_clip(0), ms.synth_clip(3)] + [ms.synth_clip(300 + i) for i in range(30)]
    clips = [distinct[i % 32] for i in range(nb)]
    fp = wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced)
    out = {}
    for name, es in (("es", True), ("kv", False)):
        ctx = wb.Context(model, nb, cross_es=es)
        assert ctx.cross_mode == int(es)
        ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 2, eot, [eot]))
        t, l = ctx.greedy_decode_resident_batch(fp, want_logits=True)
        l = np.stack(l)
        for i in range(32, nb):          # duplicates inside one context: identical arithmetic wherever the clip sits
            assert np.array_equal(l[i], l[i % 32]), (name, i)
        out[name] = l[:32]
        ctx.close()
    assert np.isfinite(out["es"]).all()
    d = np.abs(out["es"] - out["kv"]).max(axis=(1, 2))
    e_es = max(np.abs(out["es"][0][i][g0["top_ids_c"][i]] - g0["top_vals_c"][i]).max() for i in range(len(forced) + 1))
    e_kv = max(np.abs(out["kv"][0][i][g0["top_ids_c"][i]] - g0["top_vals_c"][i]).max() for i in range(len(forced) + 1))
    print(f"{nb} clips: encoder-state vs K/V form max |dlogit| {d.max():.4f} (median {np.median(d):.4f}); vs f32 golden: es {e_es:.4f}, kv {e_kv:.4f}")
    assert d.max() < 0.12                 # measured 0.07-0.08: two bf16 roundings of the same attention (logit std 1.3)
    assert max(e_es, e_kv) < BF16_LOGIT_BOUND   # the bf16-vs-f32 bound of test_bf16_teacher_forced_agreement
    srt = np.sort(out["kv"], axis=2)
    decided = (srt[:, :, -1] - srt[:, :, -2]) > 2.0 * d[:, None]
    assert (out["es"].argmax(axis=2)[decided] == out["kv"].argmax(axis=2)[decided]).all()


def test_cross_es_longform_equals_staged_calls(gpu):
    """Long-form entry and the staged calls (whisper_log_mel -> run_encoder -> greedy_decode_with_past) on a context whose token
    loop attends over the encoder states: the windows of a 72.5 s file decoded together must give, token for token, what each window
    gives alone on the same context (same kernels, results independent of the batch; reference loop src/main.rs:870-915)."""
    model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
    ctx = wb.Context(model, 4, cross_es=True)
    assert ctx.cross_mode == 1
    prompt, eot = small_prompt(ms.PRESETS["base"])
    pcm = np.concatenate([ms.synth_clip(40), ms.synth_clip(41), ms.synth_clip(42)[:200000]])  # 72.5 s
    params = wb.DecodeParams(prompt, 12, eot, [eot])
    got = ctx.transcribe_longform(pcm, params)
    offs = wb.longform_plan(pcm.size)
    assert len(got) == len(offs) == 3
    mel_full = ctx.whisper_log_mel(pcm)                 # whole-file mel, global max (:870-872)
    for off, toks in zip(offs, got):
        ctx.run_encoder(orc.window_mel(mel_full, off // 160, 3000), want_output=False)
        alone, _ = ctx.greedy_decode_with_past(params)
        assert toks.tolist() == alone.tolist()
    ctx.close()


# ------------------------------------------------------------------------------------------------
# LayerNorm folded around the GEMMs (bf16): rows whose mean is large against their spread, outlier channels
# ------------------------------------------------------------------------------------------------
def _offset_model(prec, enc_offset, dec_offset, outlier):
    """micro-size model whose residual streams carry a common per-row offset and a few outlier channels: the synthetic weights with
    `enc_offset` / `dec_offset` added to every entry of the encoder / decoder position table and three channels of it scaled by `outlier`."""
    dims = ms.PRESETS["micro"]
    sd = ms.synth_state_dict(dims, 11)
    for name, off in (("model.encoder.embed_positions.weight", enc_offset), ("model.decoder.embed_positions.weight", dec_offset)):
        t = sd[name].copy()
        t[:, [5, 77, 200]] *= outlier
        sd[name] = (t + np.float32(off)).astype(np.float32)
    return dims, wb.Model.from_weights(dims, ms.flatten_state_dict(dims, sd), 0, prec)


@pytest.mark.parametrize("enc_offset,dec_offset,outlier", [(0.0, 0.0, 1.0), (25.0, 0.0, 1.0), (0.0, 0.0, 300.0), (25.0, 25.0, 40.0)])
def test_folded_layernorm_with_offset_rows_and_outlier_channels(gpu, monkeypatch, enc_offset, dec_offset, outlier):
    """Round-3 advisor: the folded encoder LayerNorm rounds the RAW residual rows to bf16 and takes the variance as E[x^2] - mean^2; a row whose
    mean dwarfs its spread (or with outlier channels) could lose what the unfolded LayerNorm kernel keeps.  Encoder states and teacher-forced
    logits of a bf16 context with the fold against the same context without it (WH_NO_ENC_FOLD=1), both against the exact-f32 mode of the same
    weights: the fold must stay within 1.5 x the unfolded path's error (+ a small floor)."""
    pcm = ms.synth_clip(2)
    forced = np.random.Generator(np.random.PCG64(17)).integers(0, 4099, size=15).tolist()
    prompt, eot = [3, 5, 7, 9], 2
    p = wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced)

    def run(prec, fold):
        if fold:
            monkeypatch.delenv("WH_NO_ENC_FOLD", raising=False)
        else:
            monkeypatch.setenv("WH_NO_ENC_FOLD", "1")
        dims, model = _offset_model(prec, enc_offset, dec_offset, outlier)
        ctx = wb.Context(model, 16)              # above the small-context size: the LDS-DMA GEMMs (and, unless switched off, the fold)
        enc = ctx.run_encoder(ctx.whisper_log_mel(pcm))
        _, lg = ctx.greedy_decode_with_past(p, want_logits=True)
        ctx.close()
        return enc, lg

    e32, l32 = run(wb.WH_PREC_F32, False)
    ef, lf = run(wb.WH_PREC_BF16, True)
    en, ln_ = run(wb.WH_PREC_BF16, False)
    assert np.isfinite(ef).all() and np.isfinite(lf).all()
    d_ef, d_en = np.abs(ef - e32).max(), np.abs(en - e32).max()
    d_lf, d_ln = np.abs(lf - l32).max(), np.abs(ln_ - l32).max()
    print(f"offset enc {enc_offset} dec {dec_offset} outlier x{outlier}: encoder err fold {d_ef:.4f} / no fold {d_en:.4f}; logit err fold {d_lf:.4f} / no fold {d_ln:.4f}; "
          f"|enc| max {np.abs(e32).max():.2f}, logit sigma {l32.std():.2f}")
    assert d_ef <= 1.5 * d_en + 0.02 and d_lf <= 1.5 * d_ln + 0.03
