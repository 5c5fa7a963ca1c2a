"""e4m3 quantisation: the three implementations (numpy in modelspec, C in the oracle, C++ in the library) agree bit
for bit, and agree with torch's float8_e4m3fn where torch is the independent definition."""
import numpy as np
import pytest

from oracle import oracle as orc
from whisper_rust_ort_amd import binding as wb
from whisper_rust_ort_amd import modelspec as ms


def _samples():
    rng = np.random.default_rng(3)
    parts = [rng.standard_normal(50000).astype(np.float32) * s for s in (1e-3, 1e-2, 0.1, 1, 10, 100, 300)]
    edge = np.array([0, -0.0, 448, -448, 449, 464, 480, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 3 * 2.0 ** -10,
                     2.0 ** -6, 0.0175, 17.0, 18.0, 19.0, 20.0], np.float32)   # incl. round-half-to-even ties
    return np.concatenate(parts + [edge])


def test_quantise_three_way_bit_identical():
    x = _samples()
    q_np, q_c, q_cpp = ms.quantize_e4m3(x), orc.e4m3_quantize(x), wb.e4m3_quantize(x)
    assert np.array_equal(q_np, q_c) and np.array_equal(q_np, q_cpp)
    codes = np.arange(256, dtype=np.uint8)
    d_np, d_c, d_cpp = ms.dequantize_e4m3(codes), orc.e4m3_dequantize(codes), wb.e4m3_dequantize(codes)
    assert np.array_equal(d_np, d_c, equal_nan=True) and np.array_equal(d_np, d_cpp, equal_nan=True)
    ok = ~np.isnan(d_np)
    assert np.array_equal(ms.quantize_e4m3(d_np[ok]), codes[ok])   # every finite code is a fixed point
    assert d_np[0x7E] == 448.0 and d_np[0x01] == 2.0 ** -9 and np.isnan(d_np[0x7F]) and np.isnan(d_np[0xFF])


def test_quantise_matches_torch_float8():
    torch = pytest.importorskip("torch")
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch without float8_e4m3fn")
    x = _samples()
    t = torch.from_numpy(np.clip(x, -448, 448)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    q = ms.quantize_e4m3(x)
    fin = ~np.isnan(x)
    assert np.array_equal(q[fin], t[fin])


def test_quantize_linear_rows():
    rng = np.random.default_rng(4)
    w = rng.standard_normal((7, 64)).astype(np.float32) * np.array([1e-3, 0.1, 1, 5, 0, 30, 2], np.float32)[:, None]
    q, s = ms.quantize_linear(w)
    assert s[4] == 1.0 and not (q[4] & 0x7F).any()              # zero row: scale 1, codes +-0
    amax = np.abs(w).max(axis=1)
    live = amax > 0
    assert np.all(np.abs(ms.dequantize_e4m3(q)).max(axis=1)[live] == 448.0)   # the row maximum lands on the top code
    deq = ms.dequantize_e4m3(q) * s[:, None]
    assert np.all(np.abs(deq - w)[live] <= (amax[live] / 448.0 * 16.0)[:, None] + 1e-12)   # half a top-binade step
    fq = ms.fake_quant_state_dict({"a.q_proj.weight": w, "a.q_proj.bias": w[0], "x.embed_tokens.weight": w})
    assert np.array_equal(fq["a.q_proj.weight"], deq.astype(np.float32))
    assert fq["a.q_proj.bias"] is w[0] or np.array_equal(fq["a.q_proj.bias"], w[0])
    assert np.array_equal(fq["x.embed_tokens.weight"], w)       # embeddings are not Linear weights
