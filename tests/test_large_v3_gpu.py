"""whisper-large-v3 geometry (BASELINE configs[3]: d_model 1280, 20 heads, 32+32 layers, ffn 5120, 128 mel
bins, vocab 51866) with hash-seeded weights: the HIP path against the CPU oracle.  The oracle needs a
couple of minutes of host time for the 2.3 TFLOP encoder, so the test only runs when WH_TEST_LARGE=1."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from whisper_rust_ort_amd import binding as wb
from whisper_rust_ort_amd import modelspec as ms

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.environ.get("WH_TEST_LARGE"), reason="set WH_TEST_LARGE=1 (needs ~20 GB host RAM and minutes of CPU)")
def test_large_v3_f32_against_oracle():
    dims = ms.PRESETS["large-v3"]
    w = wb.synthetic_weights("large-v3", 5)                       # the C++ generator (bit-identical to numpy's)
    model = wb.Model.from_weights(dims, w, 0, wb.WH_PREC_F32)
    ctx = wb.Context(model, 1)
    pcm = ms.synth_clip(7)
    mel = ctx.whisper_log_mel(pcm)
    mel_ref = orc.log_mel(pcm, 128)
    assert mel.shape == (128, 3000)
    np.testing.assert_allclose(mel, mel_ref, rtol=0, atol=1e-4)
    enc = ctx.run_encoder(mel_ref)
    enc_ref = orc.encoder(dims, w, mel_ref)
    print("large-v3 encoder max abs err", np.abs(enc - enc_ref).max())
    np.testing.assert_allclose(enc, enc_ref, rtol=0, atol=2e-3)
    prompt, eot = [50258, 50259, 50360, 50364], 50257
    forced = np.random.Generator(np.random.PCG64(3)).integers(0, dims.vocab, size=7).tolist()
    tg, lg = ctx.greedy_decode_with_past(wb.DecodeParams(prompt, 8, eot, forced=forced), want_logits=True)
    tr, lr = orc.decode_greedy(dims, w, enc_ref, prompt, 8, eot, forced=forced, want_logits=True)
    print("large-v3 logits max abs err", np.abs(lg - lr).max())
    assert tg.tolist() == tr.tolist()
    np.testing.assert_allclose(lg, lr, rtol=0, atol=2e-3)


@pytest.mark.skipif(not os.environ.get("WH_TEST_LARGE"), reason="set WH_TEST_LARGE=1")
def test_large_v3_bf16_runs_and_is_deterministic():
    model = wb.Model("synthetic:large-v3:5", 0, wb.WH_PREC_BF16)
    ctx = wb.Context(model, 4)
    clips = [ms.synth_clip(20 + i) for i in range(4)]
    p = wb.DecodeParams([50258, 50259, 50360, 50364], 16, 50257, [50257])
    a = ctx.transcribe_batch(clips, p)
    b = ctx.transcribe_batch(clips, p)
    assert [t.tolist() for t in a] == [t.tolist() for t in b]
    assert all(len(t) == 20 for t in a)
    print("large-v3 bf16 timings", ctx.timings())


@pytest.mark.skipif(not os.environ.get("WH_TEST_LARGE"), reason="set WH_TEST_LARGE=1")
def test_large_v3_fp8_runs_and_tracks_bf16():
    """d_model 1280 exercises the general-geometry variants of the fp8 kernels (two chunks per lane in the e4m3 cross
    attention, 8-code weight loads where a wave's K share is not a multiple of 64, 20 heads in the K/V scale tables)."""
    prompt, eot = [50258, 50259, 50360, 50364], 50257
    clips = [ms.synth_clip(20 + i) for i in range(3)]
    p = wb.DecodeParams(prompt, 12, eot, [eot])
    m8 = wb.Model("synthetic:large-v3:5", 0, wb.WH_PREC_FP8)
    c8 = wb.Context(m8, 4)
    a = c8.transcribe_batch(clips, p)
    b = c8.transcribe_batch(clips, p)
    assert [t.tolist() for t in a] == [t.tolist() for t in b]
    assert all(len(t) == 16 for t in a)
    assert c8.transcribe_batch([clips[1]], p)[0].tolist() == a[1].tolist()
    # teacher-forced on the fp8 tokens: bf16 logits of the same prefixes stay close (weights differ by the e4m3 rounding only)
    c8.run_encoder(c8.whisper_log_mel(clips[0]))
    forced = a[0][4:-1].tolist()
    _, l8 = c8.greedy_decode_with_past(wb.DecodeParams(prompt, 12, eot, [eot], forced=forced), want_logits=True)
    del c8, m8
    mb = wb.Model("synthetic:large-v3:5", 0, wb.WH_PREC_BF16)
    cb = wb.Context(mb, 1)
    cb.run_encoder(cb.whisper_log_mel(clips[0]))
    _, lb = cb.greedy_decode_with_past(wb.DecodeParams(prompt, 12, eot, [eot], forced=forced), want_logits=True)
    diff = np.abs(l8 - lb)
    print("large-v3 fp8 vs bf16 teacher-forced: mean |dlogit|", diff.mean(), "max", diff.max(), "logit scale", np.abs(lb).mean())
    assert np.isfinite(l8).all() and diff.mean() < 0.25 * np.abs(lb).mean()
