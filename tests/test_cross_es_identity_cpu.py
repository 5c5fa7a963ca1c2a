"""The algebra behind wh_cross_es.hip, checked in float64 on the CPU: attention over the projected keys and values
(present.{i}.encoder.{key,value} of the reference's decoder graphs, src/main.rs:771-787, read by every later token, :798-812) equals
attention over the encoder states themselves with the K projection moved to the query side and the V projection behind the
weighted sum — the form k_dec_qexpand / k_dec_cross_attn_es / the grouped V projection compute."""
import numpy as np


def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def test_attention_on_encoder_states_equals_projected_kv():
    rng = np.random.Generator(np.random.PCG64(11))
    S, d, H, hd = 97, 128, 2, 64          # keys, d_model, heads, head_dim (whisper: hd = 64)
    E = rng.standard_normal((S, d))
    Wq, Wk, Wv, Wo = (rng.standard_normal((d, d)) / np.sqrt(d) for _ in range(4))
    bq, bv, bo = rng.standard_normal(d), rng.standard_normal(d), rng.standard_normal(d)
    x = rng.standard_normal(d)
    q = (Wq @ x + bq) * hd ** -0.5
    # the reference's form: K = E Wk^T (no bias), V = E Wv^T + bv, per-head softmax(q_h K_h^T) V_h, out-projection
    K, V = E @ Wk.T, E @ Wv.T + bv
    att = np.concatenate([_softmax(q[h * hd:(h + 1) * hd] @ K[:, h * hd:(h + 1) * hd].T) @ V[:, h * hd:(h + 1) * hd] for h in range(H)])
    ref = Wo @ att + bo
    # the encoder-state form: qe_h = Wk_h^T q_h (d values per head), p_h = softmax(qe_h . E), ctx_h = p_h E, out_h = Wv_h ctx_h + bv_h
    out = np.empty(d)
    for h in range(H):
        rows = slice(h * hd, (h + 1) * hd)
        qe = Wk[rows].T @ q[rows]
        p = _softmax(E @ qe)
        ctx = p @ E
        out[rows] = Wv[rows] @ ctx + bv[rows]
    got = Wo @ out + bo
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10)
    # bytes the token loop streams per clip, layer and token: S d for E against 2 S d for K and V
    assert E.size * 2 == K.size + V.size
