#!/usr/bin/env python3
"""Offline weight conversion for WH_PREC_FP8: a Whisper checkpoint directory (config.json + model.safetensors in
F32 / F16 / BF16) -> the same directory layout with every Linear / QKV weight stored as OCP e4m3 codes (safetensors
dtype F8_E4M3, shape unchanged) plus one F32 scale per output channel under "<name>_scale".

This is the fp8 counterpart of the reference's weights-only INT8 step (quantize_onnx_int8.py:14-46: `--src_dir`,
`--dst_dir`, copies config.json / generation_config.json / tokenizer.json, quantises MatMul/Gemm weights only —
convolutions, LayerNorms, biases and embeddings keep their precision; same here).  Loading the result with
`--precision fp8` uses the stored codes and scales as they are; with bf16 / f32 it runs the dequantised weights.

    python whisper-rust-ort_amd/quantize_fp8.py --src_dir whisper-base --dst_dir whisper-base-fp8
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import struct
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

_NP = {"F32": "<f4", "F16": "<f2"}


def read_safetensors(path: Path):
    """-> (ordered {name: (dtype, shape, bytes)}, metadata)"""
    raw = path.read_bytes()
    (hlen,) = struct.unpack("<Q", raw[:8])
    hdr = json.loads(raw[8:8 + hlen])
    meta = hdr.pop("__metadata__", None)
    out = {}
    for name, e in sorted(hdr.items(), key=lambda kv: kv[1]["data_offsets"][0]):
        b0, b1 = e["data_offsets"]
        out[name] = (e["dtype"], list(e["shape"]), raw[8 + hlen + b0: 8 + hlen + b1])
    return out, meta


def write_safetensors(path: Path, tensors, meta=None) -> None:
    hdr, blobs, off = {}, [], 0
    for name, (dtype, shape, data) in tensors.items():
        hdr[name] = {"dtype": dtype, "shape": shape, "data_offsets": [off, off + len(data)]}
        blobs.append(data)
        off += len(data)
    if meta is not None:
        hdr["__metadata__"] = meta
    hj = json.dumps(hdr).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)   # keep the data section 8-byte aligned
    path.write_bytes(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))


def to_f32(dtype: str, data: bytes, shape) -> np.ndarray:
    if dtype == "BF16":
        u = np.frombuffer(data, "<u2").astype(np.uint32) << 16
        return u.view(np.float32).reshape(shape)
    if dtype in _NP:
        return np.frombuffer(data, _NP[dtype]).astype(np.float32).reshape(shape)
    raise SystemExit(f"unsupported source dtype {dtype}")


def quantize_dir(src: Path, dst: Path) -> dict:
    if not src.is_dir():
        raise SystemExit(f"Missing source dir: {src}")
    st = src / "model.safetensors"
    if not st.is_file():
        raise SystemExit(f"Missing weights file: {st}")
    dst.mkdir(parents=True, exist_ok=True)
    for name in ["config.json", "generation_config.json", "tokenizer.json"]:
        if (src / name).is_file():
            shutil.copy2(src / name, dst / name)
    tensors, meta = read_safetensors(st)
    out, n_q, bytes_in, bytes_out = {}, 0, 0, 0
    for name, (dtype, shape, data) in tensors.items():
        bytes_in += len(data)
        if ms.is_linear_weight(name) and len(shape) == 2 and dtype != "F8_E4M3":
            codes, scale = ms.quantize_linear(to_f32(dtype, data, shape))
            out[name] = ("F8_E4M3", shape, codes.tobytes())
            out[name + "_scale"] = ("F32", [shape[0]], scale.astype("<f4").tobytes())
            bytes_out += codes.nbytes + scale.nbytes
            n_q += 1
        else:
            out[name] = (dtype, shape, data)
            bytes_out += len(data)
    meta = dict(meta or {})
    meta["quantization"] = "e4m3fn weights, per-output-channel f32 scale (<name>_scale), Linear/QKV only"
    write_safetensors(dst / "model.safetensors", out, meta)
    return {"quantized_tensors": n_q, "tensors": len(tensors), "bytes_in": bytes_in, "bytes_out": bytes_out}


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--src_dir", default="whisper-base")
    ap.add_argument("--dst_dir", default="whisper-base-fp8")
    a = ap.parse_args()
    src, dst = Path(a.src_dir), Path(a.dst_dir)
    print(f"Quantizing {src / 'model.safetensors'} -> {dst / 'model.safetensors'}")
    r = quantize_dir(src, dst)
    print(f"{r['quantized_tensors']} of {r['tensors']} tensors to e4m3; {r['bytes_in']} -> {r['bytes_out']} bytes")
    print("DONE")
    print("Output dir:", dst)


if __name__ == "__main__":
    main()
