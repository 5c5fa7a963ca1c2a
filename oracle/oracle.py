"""ctypes loader for oracle/libwhisper_oracle.so.

TEST INFRASTRUCTURE ONLY (see whisper_oracle.c).  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libwhisper_oracle.so")


class OrcDims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_mels", "d_model", "n_heads", "enc_layers", "dec_layers",
                                       "ffn", "vocab", "n_audio_ctx", "n_text_ctx")]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "whisper_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libwhisper_oracle.so"])
    return _SO


_lib: Optional[C.CDLL] = None


def cpu_budget(cap: int = 16) -> int:
    """Usable host cores: min(affinity mask, cgroup CPU quota, cap).  The GPU box exposes every
    hardware thread of the host but only grants a share of them; an OpenMP team sized to the
    hardware count would oversubscribe that share by an order of magnitude."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        L = C.CDLL(_SO)
        f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.orc_mel_frames.restype = C.c_size_t
        L.orc_mel_frames.argtypes = [C.c_size_t]
        L.orc_mel_filterbank.argtypes = [C.c_int, f32p]
        L.orc_log_mel.argtypes = [f32p, C.c_size_t, C.c_int, f32p]
        L.orc_window_mel.argtypes = [f32p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t, f32p]
        L.orc_n_params.restype = C.c_size_t
        L.orc_n_params.argtypes = [C.POINTER(OrcDims)]
        L.orc_encoder.argtypes = [C.POINTER(OrcDims), f32p, f32p, f32p]
        L.orc_argmax_last_row.argtypes = [i64p, C.c_size_t, f32p, C.c_size_t, i64p, C.c_size_t, i64p]
        L.orc_decode_greedy.argtypes = [C.POINTER(OrcDims), f32p, f32p, i64p, C.c_size_t, C.c_size_t,
                                        C.c_int64, i64p, C.c_size_t, i64p, C.c_size_t, i64p, C.c_size_t,
                                        i64p, C.POINTER(C.c_size_t), f32p]
        u8p = C.POINTER(C.c_uint8)
        L.orc_e4m3_quantize.argtypes = [f32p, C.c_size_t, u8p]
        L.orc_e4m3_dequantize.argtypes = [u8p, C.c_size_t, f32p]
        L.orc_set_kv_fp8.argtypes = [C.c_int]
        L.orc_set_act_mx.argtypes = [C.c_int]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(int(os.environ["OMP_NUM_THREADS"]))
        _lib = L
    return _lib


def _f32(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def dims_struct(dims) -> OrcDims:
    return OrcDims(dims.n_mels, dims.d_model, dims.n_heads, dims.enc_layers, dims.dec_layers,
                   dims.ffn, dims.vocab, dims.n_audio_ctx, dims.n_text_ctx)


def mel_frames(n: int) -> int:
    return int(lib().orc_mel_frames(n))


def mel_filterbank(n_mels: int) -> np.ndarray:
    fb = np.empty((n_mels, 201), np.float32)
    lib().orc_mel_filterbank(n_mels, _f32(fb))
    return fb


def log_mel(pcm: np.ndarray, n_mels: int = 80) -> np.ndarray:
    pcm = np.ascontiguousarray(pcm, np.float32)
    if pcm.size == 0:
        raise ValueError("Empty audio")  # reference src/main.rs:414-416
    nf = mel_frames(pcm.size)
    out = np.empty((n_mels, nf), np.float32)
    rc = lib().orc_log_mel(_f32(pcm), pcm.size, n_mels, _f32(out))
    if rc:
        raise RuntimeError(f"orc_log_mel rc={rc}")
    return out


def window_mel(mel_full: np.ndarray, frame_start: int, win_frames: int = 3000) -> np.ndarray:
    mel_full = np.ascontiguousarray(mel_full, np.float32)
    out = np.empty((mel_full.shape[0], win_frames), np.float32)
    lib().orc_window_mel(_f32(mel_full), mel_full.shape[1], mel_full.shape[0], frame_start, win_frames, _f32(out))
    return out


def mx_applies(dims) -> bool:
    """The build's fp8 mode uses MX activations (fp8 MFMA) when every contraction length is a multiple of 128 from 256 on and
    d_model is a width its MX LayerNorm exists for (whisper-rust-ort_amd/csrc/wh_kernels.h: wh_mx_ln_width)."""
    return (dims.d_model in (256, 512, 1024, 1280, 1536, 2048) and dims.ffn >= 256 and dims.ffn % 128 == 0
            and dims.n_audio_ctx >= 256)


def encoder(dims, wflat: np.ndarray, mel: np.ndarray, act_mx: bool = False) -> np.ndarray:
    """act_mx: LayerNorm and GELU outputs pass through MX (block-32 power-of-two scaled e4m3) — the build's fp8 mode."""
    mel = np.ascontiguousarray(mel, np.float32)
    assert mel.shape == (dims.n_mels, 2 * dims.n_audio_ctx), mel.shape
    assert wflat.dtype == np.float32 and wflat.size == int(lib().orc_n_params(C.byref(dims_struct(dims))))
    out = np.empty((dims.n_audio_ctx, dims.d_model), np.float32)
    lib().orc_set_act_mx(1 if act_mx else 0)
    try:
        rc = lib().orc_encoder(C.byref(dims_struct(dims)), _f32(wflat), _f32(mel), _f32(out))
    finally:
        lib().orc_set_act_mx(0)
    if rc:
        raise RuntimeError(f"orc_encoder rc={rc}")
    return out


def argmax_last_row(logits: np.ndarray, suppress: Sequence[int] = ()) -> int:
    logits = np.ascontiguousarray(logits, np.float32)
    shape = np.asarray(logits.shape, np.int64)
    sup = np.asarray(list(suppress), np.int64)
    out = C.c_int64(0)
    rc = lib().orc_argmax_last_row(_i64(shape), shape.size, _f32(logits), logits.size, _i64(sup), sup.size,
                                   C.byref(out))
    if rc:
        raise ValueError(f"Unexpected logits shape: {logits.shape}")  # src/main.rs:710-716
    return int(out.value)


def decode_greedy(dims, wflat: np.ndarray, enc: np.ndarray, prompt: Sequence[int], max_new: int, eot: int,
                  suppress: Sequence[int] = (), begin_suppress: Sequence[int] = (),
                  forced: Optional[Sequence[int]] = None, want_logits: bool = False, kv_fp8: bool = False,
                  act_mx: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """kv_fp8: cross-attention K/V pass through e4m3 with one scale per (layer, K|V, head) — the build's fp8 mode;
    act_mx: the encoder states enter the cross K/V projections in MX form (see encoder)."""
    enc = np.ascontiguousarray(enc, np.float32)
    pr = np.asarray(list(prompt), np.int64)
    sup = np.asarray(list(suppress), np.int64)
    bsup = np.asarray(list(begin_suppress), np.int64)
    fo = np.asarray(list(forced) if forced is not None else [], np.int64)
    toks = np.zeros(pr.size + max_new, np.int64)
    n_out = C.c_size_t(0)
    logits = np.zeros((max_new, dims.vocab), np.float32) if want_logits else None
    lib().orc_set_kv_fp8(1 if kv_fp8 else 0)
    lib().orc_set_act_mx(1 if act_mx else 0)
    try:
        rc = lib().orc_decode_greedy(C.byref(dims_struct(dims)), _f32(wflat), _f32(enc), _i64(pr), pr.size, max_new,
                                     eot, _i64(sup), sup.size, _i64(bsup), bsup.size,
                                     _i64(fo) if fo.size else None, fo.size, _i64(toks), C.byref(n_out),
                                     _f32(logits) if want_logits else None)
    finally:
        lib().orc_set_kv_fp8(0)
        lib().orc_set_act_mx(0)
    if rc:
        raise RuntimeError(f"orc_decode_greedy rc={rc}")
    n = int(n_out.value)
    return toks[:n].copy(), (logits[: n - pr.size].copy() if want_logits else None)


def e4m3_quantize(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(x.shape, np.uint8)
    lib().orc_e4m3_quantize(_f32(x), x.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def e4m3_dequantize(c: np.ndarray) -> np.ndarray:
    c = np.ascontiguousarray(c, np.uint8)
    out = np.zeros(c.shape, np.float32)
    lib().orc_e4m3_dequantize(c.ctypes.data_as(C.POINTER(C.c_uint8)), c.size, _f32(out))
    return out


def num_threads() -> int:
    return int(lib().orc_num_threads())
