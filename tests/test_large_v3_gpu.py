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
