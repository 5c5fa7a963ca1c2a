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


def _write_nano_checkpoint(d, sd, dims):
    import json
    import struct
    d.mkdir()
    hdr, blobs, off = {}, [], 0
    for name, arr in sd.items():
        b = arr.astype("<f4").tobytes()
        hdr[name] = {"dtype": "F32", "shape": list(arr.shape), "data_offsets": [off, off + len(b)]}
        blobs.append(b)
        off += len(b)
    hj = json.dumps(hdr).encode()
    (d / "model.safetensors").write_bytes(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))
    (d / "config.json").write_text(json.dumps({
        "num_mel_bins": 80, "d_model": dims.d_model, "encoder_attention_heads": dims.n_heads, "decoder_attention_heads": dims.n_heads,
        "encoder_layers": dims.enc_layers, "decoder_layers": dims.dec_layers, "encoder_ffn_dim": dims.ffn, "decoder_ffn_dim": dims.ffn,
        "vocab_size": dims.vocab, "max_source_positions": 1500, "max_target_positions": 448}))
    (d / "generation_config.json").write_text("{}")


def test_quantize_fp8_tool(tmp_path):
    """The offline converter (reference analogue: quantize_onnx_int8.py) rewrites only Linear weights, as e4m3 codes +
    per-row scales that equal modelspec.quantize_linear, and carries the side files over."""
    import subprocess
    import sys
    import os
    from whisper_rust_ort_amd import quantize_fp8 as qt
    dims = ms.PRESETS["nano"]
    sd = ms.synth_state_dict(dims, 7)
    src, dst = tmp_path / "src", tmp_path / "dst"
    _write_nano_checkpoint(src, sd, dims)
    tool = os.path.join(os.path.dirname(os.path.abspath(qt.__file__)), "quantize_fp8.py")
    r = subprocess.run([sys.executable, tool, "--src_dir", str(src), "--dst_dir", str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[-2:] == ["DONE", f"Output dir: {dst}"]
    assert (dst / "config.json").read_text() == (src / "config.json").read_text() and (dst / "generation_config.json").is_file()
    out, meta = qt.read_safetensors(dst / "model.safetensors")
    assert "e4m3" in meta["quantization"]
    n_lin = 0
    for name, arr in sd.items():
        dtype, shape, data = out[name]
        if ms.is_linear_weight(name):
            n_lin += 1
            q, s = ms.quantize_linear(arr)
            assert dtype == "F8_E4M3" and shape == list(arr.shape) and data == q.tobytes()
            sdt, sshape, sdata = out[name + "_scale"]
            assert sdt == "F32" and sshape == [arr.shape[0]] and sdata == s.astype("<f4").tobytes()
        else:
            assert dtype == "F32" and data == arr.astype("<f4").tobytes() and (name + "_scale") not in out
    assert n_lin == 6 * dims.enc_layers + 10 * dims.dec_layers
    missing = subprocess.run([sys.executable, tool, "--src_dir", str(tmp_path / "nope"), "--dst_dir", str(dst)], capture_output=True, text=True)
    assert missing.returncode != 0 and "Missing source dir" in missing.stderr
