"""CPU oracle (oracle/whisper_oracle.c) held to the committed golden vectors.

The vectors come from tests/golden/make_golden.py (HF `transformers` Whisper classes with the
hash-seeded weights); tolerances are fp32 round-off between two summation orders.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as orc
from whisper_rust_ort_amd import modelspec as ms

MEL_TOL = 1e-4      # f32 filterbank + f64 DFT vs HF's numpy f64 pipeline (4.7e-5 measured)
ENC_TOL = 2e-4
LOGIT_TOL = 1e-3    # north_star: logits within 1e-3 fp32


def _load(golden_dir, name):
    p = os.path.join(golden_dir, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not generated")
    return np.load(p)


def _model(g):
    dims = ms.PRESETS[str(g["preset"])]
    sd = ms.synth_state_dict(dims, int(g["seed"]))
    sums = np.array([sd[n].astype(np.float64).sum() for n, _ in ms.tensor_table(dims)])
    np.testing.assert_allclose(sums, g["weight_sums"], rtol=0, atol=1e-6)
    return dims, ms.flatten_state_dict(dims, sd)


def _clip(g):
    pcm = ms.synth_clip(int(g["clip"]))
    np.testing.assert_array_equal(pcm[:64], g["pcm_head"])
    assert hashlib.sha256(pcm.tobytes()).hexdigest() == str(g["pcm_sha256"])
    return pcm


def _check_rows(logits, top_ids, top_vals, tol):
    assert logits.shape[0] == top_ids.shape[0]
    for i in range(logits.shape[0]):
        np.testing.assert_allclose(logits[i][top_ids[i]], top_vals[i], rtol=0, atol=tol)


@pytest.mark.parametrize("name", ["nano_s7_c0.npz", "nano_s7_c1.npz", "micro_s11_c2.npz"])
def test_oracle_full_path_small(golden_dir, name):
    g = _load(golden_dir, name)
    dims, w = _model(g)
    pcm = _clip(g)
    mel = orc.log_mel(pcm, dims.n_mels)
    assert mel.shape == (dims.n_mels, 3000)
    np.testing.assert_allclose(mel[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)
    assert abs(mel.astype(np.float64).sum() - float(g["mel_sum"])) < 0.5
    enc = orc.encoder(dims, w, mel)
    np.testing.assert_allclose(enc[g["enc_rows"]], g["enc_slice"], rtol=0, atol=ENC_TOL)
    np.testing.assert_allclose(enc.astype(np.float64).mean(0), g["enc_col_mean"], rtol=0, atol=ENC_TOL)
    prompt, eot, max_new = g["prompt"].tolist(), int(g["eot"]), int(g["max_new"])
    ta, la = orc.decode_greedy(dims, w, enc, prompt, max_new, eot, want_logits=True)
    assert ta.tolist() == g["tokens_a"].tolist()
    _check_rows(la, g["top_ids_a"], g["top_vals_a"], LOGIT_TOL)
    tb, lb = orc.decode_greedy(dims, w, enc, prompt, max_new, eot, g["suppress_b"].tolist(),
                               g["begin_suppress_b"].tolist(), want_logits=True)
    assert tb.tolist() == g["tokens_b"].tolist()
    _check_rows(lb, g["top_ids_b"], g["top_vals_b"], LOGIT_TOL)
    forced = g["forced_c"].tolist()
    tc, lc = orc.decode_greedy(dims, w, enc, prompt, len(forced) + 1, eot, forced=forced, want_logits=True)
    assert tc.tolist() == g["tokens_c"].tolist()
    _check_rows(lc, g["top_ids_c"], g["top_vals_c"], LOGIT_TOL)
    n = g["logits_c_head"].shape
    np.testing.assert_allclose(lc[: n[0], : n[1]], g["logits_c_head"][:, : lc.shape[1]], rtol=0, atol=LOGIT_TOL)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["base_s1234_c0.npz"])
def test_oracle_full_path_base(golden_dir, name):
    """whisper-base dims (SURVEY §8c golden vectors 1-5): mel, encoder slices, per-step top-k, tokens."""
    g = _load(golden_dir, name)
    dims, w = _model(g)
    pcm = _clip(g)
    mel = orc.log_mel(pcm, dims.n_mels)
    np.testing.assert_allclose(mel[:, ::25], g["mel_slice"], rtol=0, atol=MEL_TOL)
    enc = orc.encoder(dims, w, mel)
    np.testing.assert_allclose(enc[g["enc_rows"]], g["enc_slice"], rtol=0, atol=ENC_TOL)
    prompt, eot = g["prompt"].tolist(), int(g["eot"])
    assert prompt == [50258, 50259, 50359, 50363]  # reference src/main.rs:549-566
    forced = g["forced_c"].tolist()
    tc, lc = orc.decode_greedy(dims, w, enc, prompt, len(forced) + 1, eot, forced=forced, want_logits=True)
    assert tc.tolist() == g["tokens_c"].tolist()
    _check_rows(lc, g["top_ids_c"], g["top_vals_c"], LOGIT_TOL)
    ta, la = orc.decode_greedy(dims, w, enc, prompt, 32, eot, want_logits=True)
    assert ta.tolist() == g["tokens_a"][: len(ta)].tolist()
    _check_rows(la, g["top_ids_a"][: len(la)], g["top_vals_a"][: len(la)], LOGIT_TOL)
