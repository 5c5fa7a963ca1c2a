"""ctypes binding of libwhisper_hip.so — the host-side mirror of the reference's seam.

Names follow reference src/main.rs: `whisper_log_mel` (:407), `run_encoder` (:698),
`greedy_decode_with_past` (:753), plus the batch entry that replaces the per-window body of
`transcribe_longform_chunked` (:870-915).  Errors surface as `WhisperHipError` carrying the C
status code and the library's message (the reference propagates `anyhow` errors, :1104-1108).

There is no fallback of any kind: if the shared library or a GPU is missing, loading or the first
call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwhisper_hip.so")

WH_PREC_F32, WH_PREC_BF16, WH_PREC_FP8, WH_PREC_F16X3 = 0, 1, 2, 3
PRECISIONS = {"f32": WH_PREC_F32, "bf16": WH_PREC_BF16, "fp8": WH_PREC_FP8, "f16x3": WH_PREC_F16X3}
WH_N_FRAMES, WH_CLIP_SAMPLES = 3000, 480000
KG_NAMES = ("mel", "enc_gemm", "enc_attn", "dec_cross_attn", "dec_gemm", "dec_other")

STATUS = {0: "WH_OK", 1: "WH_ERR_EMPTY_AUDIO", 2: "WH_ERR_BAD_SHAPE", 3: "WH_ERR_STATE", 4: "WH_ERR_ARG",
          5: "WH_ERR_NOMEM", 6: "WH_ERR_HIP", 7: "WH_ERR_IO", 8: "WH_ERR_UNSUPPORTED"}

# every symbol include/whisper_hip.h declares
EXPORTS = ("wh_model_load", "wh_model_create", "wh_model_free", "wh_model_get_dims", "wh_model_precision",
           "wh_model_export_tensor", "wh_ctx_create", "wh_ctx_create_ex", "wh_ctx_free", "wh_ctx_cross_mode", "wh_ctx_placement", "wh_last_error", "wh_get_timings",
           "wh_mel_frames", "wh_log_mel", "wh_encode", "wh_decode_greedy", "wh_decode_greedy_batch", "wh_decode_greedy_rows", "wh_transcribe_batch",
           "wh_transcribe_batch_next", "wh_transcribe_batch_device", "wh_transcribe_batch_device_next", "wh_longform_plan", "wh_transcribe_longform", "wh_profile_enable",
           "wh_profile_get", "wh_synthetic_weights", "wh_e4m3_quantize", "wh_e4m3_dequantize", "wh_abi_version",
           "wh_device_count")


class WhisperHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class WhDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "d_model", "n_heads", "enc_layers", "dec_layers", "ffn", "vocab",
                                         "n_audio_ctx", "n_text_ctx")]


class WhTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("preprocess_s", "encode_s", "decode_s", "total_s", "h2d_s", "d2h_s")]


class WhDecodeParams(C.Structure):
    _fields_ = [("prompt", C.POINTER(C.c_int64)), ("n_prompt", C.c_size_t), ("max_new_tokens", C.c_size_t),
                ("eot", C.c_int64), ("suppress", C.POINTER(C.c_int64)), ("n_suppress", C.c_size_t),
                ("begin_suppress", C.POINTER(C.c_int64)), ("n_begin_suppress", C.c_size_t),
                ("forced", C.POINTER(C.c_int64)), ("n_forced", C.c_size_t)]


class WhClip(C.Structure):
    _fields_ = [("pcm", C.POINTER(C.c_float)), ("n_samples", C.c_size_t)]


class WhCtxOpts(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("max_batch", C.c_int), ("flags", C.c_int),
                ("enc_cu_mask", C.POINTER(C.c_uint32)), ("enc_cu_mask_words", C.c_size_t),
                ("dec_cu_mask", C.POINTER(C.c_uint32)), ("dec_cu_mask_words", C.c_size_t)]


WH_CTX_TWO_STREAMS = 1
WH_CTX_CROSS_ES_ON = 2
WH_CTX_CROSS_ES_OFF = 4
N_CUS = 256   # MI355X: 8 XCDs x 32 compute units


def cu_mask(first: int, count: int, total: int = N_CUS) -> np.ndarray:
    """CU mask words with bits first .. first+count-1 set.  On MI355X bit i is compute unit i // 8 of XCD i % 8
    (tools/cu_mask_probe.hip), so a contiguous bit range whose ends are multiples of 8 takes the same number of
    compute units from every XCD."""
    if first < 0 or count < 1 or first + count > total:
        raise ValueError(f"CU range [{first}, {first + count}) outside 0..{total}")
    w = np.zeros((total + 31) // 32, np.uint32)
    for i in range(first, first + count):
        w[i >> 5] |= np.uint32(1 << (i & 31))
    return w


_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                                "there is no CPU fallback")
    L = C.CDLL(path)
    vp, f32p, i64p, szp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_size_t)
    L.wh_model_load.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.wh_model_create.argtypes = [C.POINTER(WhDims), f32p, C.c_size_t, C.c_int, C.c_int, C.POINTER(vp)]
    L.wh_model_free.argtypes = [vp]
    L.wh_model_free.restype = None
    L.wh_model_get_dims.argtypes = [vp, C.POINTER(WhDims)]
    L.wh_model_precision.argtypes = [vp]
    L.wh_model_export_tensor.argtypes = [vp, C.c_char_p, f32p, C.c_size_t, szp]
    L.wh_ctx_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.wh_ctx_create_ex.argtypes = [vp, C.POINTER(WhCtxOpts), C.POINTER(vp)]
    L.wh_ctx_free.argtypes = [vp]
    L.wh_ctx_free.restype = None
    L.wh_ctx_cross_mode.argtypes = [vp]
    L.wh_ctx_placement.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.wh_last_error.argtypes = [vp]
    L.wh_last_error.restype = C.c_char_p
    L.wh_get_timings.argtypes = [vp, C.POINTER(WhTiming)]
    L.wh_mel_frames.argtypes = [C.c_size_t]
    L.wh_mel_frames.restype = C.c_size_t
    L.wh_log_mel.argtypes = [vp, f32p, C.c_size_t, f32p, C.c_size_t, szp]
    L.wh_encode.argtypes = [vp, f32p, f32p]
    L.wh_decode_greedy.argtypes = [vp, C.POINTER(WhDecodeParams), i64p, C.c_size_t, szp, f32p, C.c_size_t]
    L.wh_decode_greedy_batch.argtypes = [vp, C.POINTER(WhDecodeParams), i64p, C.c_size_t, szp, C.c_size_t, szp, f32p, C.c_size_t]
    L.wh_decode_greedy_rows.argtypes = [vp, C.POINTER(WhDecodeParams), C.POINTER(C.c_int32), C.c_size_t, i64p, C.c_size_t, szp, C.c_size_t, szp, f32p, C.c_size_t]
    L.wh_transcribe_batch.argtypes = [vp, C.POINTER(WhClip), C.c_size_t, C.POINTER(WhDecodeParams), i64p, szp]
    L.wh_transcribe_batch_next.argtypes = [vp, C.POINTER(WhClip), C.c_size_t, C.POINTER(WhClip), C.c_size_t, C.POINTER(WhDecodeParams), i64p, szp]
    L.wh_transcribe_batch_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(WhDecodeParams), i64p, szp]
    L.wh_transcribe_batch_device_next.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(WhDecodeParams), i64p, szp]
    L.wh_longform_plan.argtypes = [C.c_size_t, C.c_double, C.c_double, szp, C.c_size_t, szp]
    L.wh_transcribe_longform.argtypes = [vp, f32p, C.c_size_t, C.c_double, C.c_double, C.POINTER(WhDecodeParams), i64p,
                                         szp, C.c_size_t, szp]
    L.wh_profile_enable.argtypes = [vp, C.c_int]
    L.wh_profile_get.argtypes = [vp, C.POINTER(C.c_double), i64p]
    L.wh_synthetic_weights.argtypes = [C.c_char_p, C.c_uint64, f32p, C.c_size_t, szp]
    L.wh_e4m3_quantize.argtypes = [f32p, C.c_size_t, C.POINTER(C.c_uint8)]
    L.wh_e4m3_quantize.restype = None
    L.wh_e4m3_dequantize.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, f32p]
    L.wh_e4m3_dequantize.restype = None
    _lib = L
    return L


def _f32(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


@dataclass
class DecodeParams:
    """Arguments of greedy_decode_with_past (reference src/main.rs:753-761) + GenerationCfg (:102-106)."""
    prompt: Sequence[int]
    max_new_tokens: int = 128
    eot: int = 50257
    suppress_tokens: Sequence[int] = ()
    begin_suppress_tokens: Sequence[int] = ()
    forced: Optional[Sequence[int]] = None

    def to_c(self):
        keep = [np.asarray(list(self.prompt), np.int64), np.asarray(list(self.suppress_tokens), np.int64),
                np.asarray(list(self.begin_suppress_tokens), np.int64),
                np.asarray(list(self.forced) if self.forced is not None else [], np.int64)]
        p = WhDecodeParams(_i64(keep[0]), keep[0].size, self.max_new_tokens, self.eot,
                           _i64(keep[1]) if keep[1].size else None, keep[1].size,
                           _i64(keep[2]) if keep[2].size else None, keep[2].size,
                           _i64(keep[3]) if keep[3].size else None, keep[3].size)
        return p, keep


class Model:
    """Replaces the three `ort::Session`s (reference src/main.rs:1103-1108)."""

    def __init__(self, spec_or_dir: str, device: int = 0, precision: int = WH_PREC_BF16):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.wh_model_load(spec_or_dir.encode(), device, precision, C.byref(h))
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(None) or b"").decode())
        self.h = h
        d = WhDims()
        self.lib.wh_model_get_dims(self.h, C.byref(d))
        self.dims = d
        self.precision = precision
        self.device = device

    @classmethod
    def from_weights(cls, dims, wflat: np.ndarray, device: int = 0, precision: int = WH_PREC_BF16) -> "Model":
        self = cls.__new__(cls)
        self.lib = load_library()
        wflat = np.ascontiguousarray(wflat, np.float32)
        cd = WhDims(dims.n_mels, dims.d_model, dims.n_heads, dims.enc_layers, dims.dec_layers, dims.ffn, dims.vocab,
                    dims.n_audio_ctx, dims.n_text_ctx)
        h = C.c_void_p()
        rc = self.lib.wh_model_create(C.byref(cd), _f32(wflat), wflat.size, device, precision, C.byref(h))
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(None) or b"").decode())
        self.h, self.dims, self.precision, self.device = h, cd, precision, device
        return self

    def export_tensor(self, name: str) -> np.ndarray:
        n = C.c_size_t(0)
        rc = self.lib.wh_model_export_tensor(self.h, name.encode(), None, 0, C.byref(n))
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(None) or b"").decode())
        out = np.empty(n.value, np.float32)
        rc = self.lib.wh_model_export_tensor(self.h, name.encode(), _f32(out), out.size, C.byref(n))
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(None) or b"").decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.wh_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One stream's workspace + KV cache (reference: per-thread IoBinding + `past`, src/main.rs:786-791)."""

    def __init__(self, model: Model, max_batch: int = 1, enc_cu_mask: Optional[np.ndarray] = None,
                 dec_cu_mask: Optional[np.ndarray] = None, two_streams: bool = False, cross_es: Optional[bool] = None):
        """enc_cu_mask / dec_cu_mask (uint32 words, see cu_mask()) or two_streams: the chip-partition form of the context
        (wh_ctx_create_ex) — log-mel + encoder on their own stream, beside the token loop of the previous batch.
        cross_es: force the token loop's cross-attention onto the encoder states (True) or onto the projected K / V (False);
        None = the library's rule (bf16 whisper-base geometry, max_batch >= 256)."""
        self.model, self.lib, self.max_batch = model, model.lib, max_batch
        h = C.c_void_p()
        flags = (WH_CTX_TWO_STREAMS if two_streams else 0) | (0 if cross_es is None else (WH_CTX_CROSS_ES_ON if cross_es else WH_CTX_CROSS_ES_OFF))
        if enc_cu_mask is None and dec_cu_mask is None and not flags:
            rc = self.lib.wh_ctx_create(model.h, max_batch, C.byref(h))
        else:
            em = np.ascontiguousarray(enc_cu_mask, np.uint32) if enc_cu_mask is not None else None
            dm = np.ascontiguousarray(dec_cu_mask, np.uint32) if dec_cu_mask is not None else None
            u32p = C.POINTER(C.c_uint32)
            o = WhCtxOpts(C.sizeof(WhCtxOpts), max_batch, flags,
                          em.ctypes.data_as(u32p) if em is not None else None, em.size if em is not None else 0,
                          dm.ctypes.data_as(u32p) if dm is not None else None, dm.size if dm is not None else 0)
            rc = self.lib.wh_ctx_create_ex(model.h, C.byref(o), C.byref(h))
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(None) or b"").decode())
        self.h = h

    def _check(self, rc: int):
        if rc:
            raise WhisperHipError(rc, (self.lib.wh_last_error(self.h) or b"").decode())

    @property
    def cross_mode(self) -> int:
        """0: the token loop streams the projected cross K / V of every layer; 1: the encoder states (wh_cross_es.hip)."""
        return int(self.lib.wh_ctx_cross_mode(self.h))

    @property
    def placement(self) -> dict:
        """wh_ctx_placement: how many workspaces wh_ctx_create_ex timed (0: step not taken) and the cross-attention launch time on the first / kept one."""
        a, b = C.c_float(0), C.c_float(0)
        n = int(self.lib.wh_ctx_placement(self.h, C.byref(a), C.byref(b)))
        return {"workspaces_timed": n, "first_us_per_launch": float(a.value), "kept_us_per_launch": float(b.value)}

    # --- the reference's three functions -----------------------------------------------------
    def whisper_log_mel(self, audio_16k: np.ndarray) -> np.ndarray:
        """whisper_log_mel_80 (src/main.rs:407-509) → [n_mels, n_frames] f32."""
        pcm = np.ascontiguousarray(audio_16k, np.float32)
        nf = int(self.lib.wh_mel_frames(pcm.size)) if pcm.size else 1
        out = np.empty((self.model.dims.n_mels, nf), np.float32)
        got = C.c_size_t(0)
        self._check(self.lib.wh_log_mel(self.h, _f32(pcm), pcm.size, _f32(out), nf, C.byref(got)))
        return out

    def run_encoder(self, input_features: np.ndarray, want_output: bool = True) -> Optional[np.ndarray]:
        """run_encoder (src/main.rs:698-707): [1?, n_mels, 3000] → [1500, d_model] f32."""
        mel = np.ascontiguousarray(input_features, np.float32)
        if mel.ndim == 3 and mel.shape[0] == 1:
            mel = mel[0]
        if mel.shape != (self.model.dims.n_mels, WH_N_FRAMES):
            raise WhisperHipError(2, f"expected input_features [{self.model.dims.n_mels}, {WH_N_FRAMES}], got {mel.shape}")
        out = np.empty((self.model.dims.n_audio_ctx, self.model.dims.d_model), np.float32) if want_output else None
        self._check(self.lib.wh_encode(self.h, _f32(mel), _f32(out) if want_output else None))
        return out

    def greedy_decode_with_past(self, params: DecodeParams, want_logits: bool = False
                                ) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """greedy_decode_with_past (src/main.rs:753-829) on the encoder states held by this context."""
        p, keep = params.to_c()
        cap = len(params.prompt) + params.max_new_tokens
        toks = np.zeros(cap, np.int64)
        n = C.c_size_t(0)
        logits = np.zeros((params.max_new_tokens, self.model.dims.vocab), np.float32) if want_logits else None
        self._check(self.lib.wh_decode_greedy(self.h, C.byref(p), _i64(toks), cap, C.byref(n),
                                              _f32(logits) if want_logits else None, params.max_new_tokens))
        nt = int(n.value)
        return toks[:nt].copy(), (logits[: nt - len(params.prompt)].copy() if want_logits else None)

    def greedy_decode_resident_batch(self, params: DecodeParams, want_logits: bool = False
                                     ) -> Tuple[List[np.ndarray], Optional[List[np.ndarray]]]:
        """Parity harness: greedy_decode_with_past over every clip whose encoder states are resident (the last
        transcribe_batch / run_encoder), decoded as ONE batch, optionally with the logits rows of every clip."""
        p, keep = params.to_c()
        cap = len(params.prompt) + params.max_new_tokens
        nb = self.max_batch
        toks = np.zeros((nb, cap), np.int64)
        n = (C.c_size_t * nb)()
        got = C.c_size_t(0)
        logits = np.zeros((nb, params.max_new_tokens, self.model.dims.vocab), np.float32) if want_logits else None
        self._check(self.lib.wh_decode_greedy_batch(self.h, C.byref(p), _i64(toks), cap, n, nb, C.byref(got),
                                                    _f32(logits) if want_logits else None, params.max_new_tokens))
        k = int(got.value)
        out = [toks[i, : n[i]].copy() for i in range(k)]
        lg = [logits[i, : n[i] - len(params.prompt)] for i in range(k)] if want_logits else None
        return out, lg

    def greedy_decode_resident_rows(self, params: DecodeParams, rows: Sequence[int]) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """The same batch decode with the logits of the chosen batch rows only (wh_decode_greedy_rows): tokens of every clip,
        logits [len(rows)][generated][vocab]."""
        p, keep = params.to_c()
        cap = len(params.prompt) + params.max_new_tokens
        nb = self.max_batch
        toks = np.zeros((nb, cap), np.int64)
        n = (C.c_size_t * nb)()
        got = C.c_size_t(0)
        sel = np.ascontiguousarray(rows, np.int32)
        logits = np.zeros((len(sel), params.max_new_tokens, self.model.dims.vocab), np.float32)
        self._check(self.lib.wh_decode_greedy_rows(self.h, C.byref(p), sel.ctypes.data_as(C.POINTER(C.c_int32)), len(sel), _i64(toks), cap, n, nb,
                                                   C.byref(got), _f32(logits), params.max_new_tokens))
        k = int(got.value)
        out = [toks[i, : n[i]].copy() for i in range(k)]
        return out, [logits[j, : n[int(r)] - len(params.prompt)] for j, r in enumerate(sel)]

    # --- fused batch entries ----------------------------------------------------------------------
    def transcribe_batch_next(self, clips: Sequence[np.ndarray], params: DecodeParams, next_clips: Optional[Sequence[np.ndarray]] = None
                              ) -> List[np.ndarray]:
        """wh_transcribe_batch_next: `clips` transcribed like transcribe_batch while `next_clips` (arrays that stay alive and unchanged
        until the call that transcribes them — views of page-locked memory make the copy asynchronous) are copied to the device."""
        arrs = [np.ascontiguousarray(c, np.float32) for c in clips]
        cl = (WhClip * len(arrs))(*[WhClip(_f32(a), a.size) for a in arrs])
        nxt = [np.ascontiguousarray(c, np.float32) for c in next_clips] if next_clips is not None else None
        ncl = (WhClip * len(nxt))(*[WhClip(_f32(a), a.size) for a in nxt]) if nxt else None
        p, keep = params.to_c()
        stride = len(params.prompt) + params.max_new_tokens
        toks = np.zeros((len(arrs), stride), np.int64)
        n = (C.c_size_t * len(arrs))()
        self._check(self.lib.wh_transcribe_batch_next(self.h, cl, len(arrs), ncl, len(nxt) if nxt else 0, C.byref(p), _i64(toks), n))
        self._keep_next = nxt   # (contiguous copies, if any were made, must outlive the asynchronous copy)
        return [toks[i, : n[i]].copy() for i in range(len(arrs))]

    def transcribe_batch(self, clips: Sequence[np.ndarray], params: DecodeParams) -> List[np.ndarray]:
        arrs = [np.ascontiguousarray(c, np.float32) for c in clips]
        cl = (WhClip * len(arrs))(*[WhClip(_f32(a), a.size) for a in arrs])
        p, keep = params.to_c()
        stride = len(params.prompt) + params.max_new_tokens
        toks = np.zeros((len(arrs), stride), np.int64)
        n = (C.c_size_t * len(arrs))()
        self._check(self.lib.wh_transcribe_batch(self.h, cl, len(arrs), C.byref(p), _i64(toks), n))
        return [toks[i, : n[i]].copy() for i in range(len(arrs))]

    def transcribe_batch_device(self, d_pcm_ptr: int, n_clips: int, params: DecodeParams, next_ptr: Optional[int] = None,
                                next_n: int = 0) -> List[np.ndarray]:
        """PCM already resident in HBM: `d_pcm_ptr` = device address of [n_clips][480000] f32.  next_ptr / next_n: the
        batch whose log-mel + encoder are to run beside this batch's token loop (wh_transcribe_batch_device_next)."""
        p, keep = params.to_c()
        stride = len(params.prompt) + params.max_new_tokens
        toks = np.zeros((n_clips, stride), np.int64)
        n = (C.c_size_t * n_clips)()
        if next_ptr is None:
            self._check(self.lib.wh_transcribe_batch_device(self.h, C.c_void_p(d_pcm_ptr), n_clips, C.byref(p), _i64(toks), n))
        else:
            self._check(self.lib.wh_transcribe_batch_device_next(self.h, C.c_void_p(d_pcm_ptr), n_clips, C.c_void_p(next_ptr), next_n,
                                                                 C.byref(p), _i64(toks), n))
        return [toks[i, : n[i]].copy() for i in range(n_clips)]

    def transcribe_longform(self, audio_16k: np.ndarray, params: DecodeParams, chunk_length_s: float = 30.0,
                            overlap_s: float = 5.0) -> List[np.ndarray]:
        pcm = np.ascontiguousarray(audio_16k, np.float32)
        nch = C.c_size_t(0)
        self.lib.wh_longform_plan(pcm.size, chunk_length_s, overlap_s, None, 0, C.byref(nch))
        k = max(1, int(nch.value))
        p, keep = params.to_c()
        stride = len(params.prompt) + params.max_new_tokens
        toks = np.zeros((k, stride), np.int64)
        n = (C.c_size_t * k)()
        got = C.c_size_t(0)
        self._check(self.lib.wh_transcribe_longform(self.h, _f32(pcm), pcm.size, chunk_length_s, overlap_s, C.byref(p),
                                                    _i64(toks), n, k, C.byref(got)))
        return [toks[i, : n[i]].copy() for i in range(int(got.value))]

    # --- measurement -------------------------------------------------------------------------------
    def timings(self) -> dict:
        t = WhTiming()
        self.lib.wh_get_timings(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in WhTiming._fields_}

    def profile_enable(self, groups=True, stride: int = 0):
        """groups: True/False (all/none) or an iterable of KG_NAMES to event-time; stride > 1 samples every
        stride-th decoder position (the rest replay the hipGraph)."""
        if groups is True:
            mask = (1 << len(KG_NAMES)) - 1
        elif not groups:
            mask = 0
        else:
            mask = sum(1 << KG_NAMES.index(g) for g in groups)
        self.lib.wh_profile_enable(self.h, mask | ((stride & 0xFFFF) << 16))

    def profile_get(self) -> dict:
        ms = (C.c_double * len(KG_NAMES))()
        ln = (C.c_int64 * len(KG_NAMES))()
        self.lib.wh_profile_get(self.h, ms, ln)
        return {k: {"ms": ms[i], "launches": int(ln[i])} for i, k in enumerate(KG_NAMES)}

    def close(self):
        if getattr(self, "h", None):
            self.lib.wh_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def longform_plan(n_samples: int, chunk_length_s: float = 30.0, overlap_s: float = 5.0) -> List[int]:
    L = load_library()
    n = C.c_size_t(0)
    L.wh_longform_plan(n_samples, chunk_length_s, overlap_s, None, 0, C.byref(n))
    offs = (C.c_size_t * max(1, n.value))()
    L.wh_longform_plan(n_samples, chunk_length_s, overlap_s, offs, n.value, C.byref(n))
    return [int(offs[i]) for i in range(n.value)]


def synthetic_weights(preset: str, seed: int) -> np.ndarray:
    """Host-only: the C++ generator's f32 blob in canonical order."""
    L = load_library()
    n = C.c_size_t(0)
    rc = L.wh_synthetic_weights(preset.encode(), seed, None, 0, C.byref(n))
    if rc:
        raise WhisperHipError(rc, (L.wh_last_error(None) or b"").decode())
    out = np.empty(n.value, np.float32)
    rc = L.wh_synthetic_weights(preset.encode(), seed, _f32(out), out.size, C.byref(n))
    if rc:
        raise WhisperHipError(rc, (L.wh_last_error(None) or b"").decode())
    return out


def e4m3_quantize(x: np.ndarray) -> np.ndarray:
    """Host-only: f32 -> OCP e4m3fn codes exactly as WH_PREC_FP8 stores weights."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    load_library().wh_e4m3_quantize(_f32(x), x.size, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def e4m3_dequantize(codes: np.ndarray) -> np.ndarray:
    codes = np.ascontiguousarray(codes, np.uint8)
    out = np.empty(codes.shape, np.float32)
    load_library().wh_e4m3_dequantize(codes.ctypes.data_as(C.POINTER(C.c_uint8)), codes.size, _f32(out))
    return out


def device_count() -> int:
    return int(load_library().wh_device_count())


class HipRuntime:
    """Just enough of the HIP runtime (the instance libwhisper_hip.so already loaded) to keep PCM resident in HBM for
    wh_transcribe_batch_device: bench.py and the GPU tests of that entry."""

    def __init__(self):
        load_library()
        self.lib = C.CDLL("libamdhip64.so.7", mode=C.RTLD_GLOBAL)
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipSetDevice.argtypes = [C.c_int]

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with hipError {rc}")

    def upload(self, dev: int, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.hipSetDevice(dev), "hipSetDevice")
        p = C.c_void_p()
        self.check(self.lib.hipMalloc(C.byref(p), arr.nbytes), "hipMalloc")
        self.check(self.lib.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1), "hipMemcpy H2D")
        self.check(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")
        return p.value

    def malloc(self, dev: int, nbytes: int) -> int:
        """Uninitialised device memory (probe programs: a placeholder that makes the next allocation land elsewhere)."""
        self.check(self.lib.hipSetDevice(dev), "hipSetDevice")
        p = C.c_void_p()
        self.check(self.lib.hipMalloc(C.byref(p), int(nbytes)), "hipMalloc")
        return p.value

    def free(self, ptr: int):
        self.check(self.lib.hipFree(C.c_void_p(ptr)), "hipFree")

    def host_alloc(self, shape, dtype=np.float32) -> np.ndarray:
        """Page-locked host memory as a numpy array (hipHostMalloc): the source of truly asynchronous host-to-device copies."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.lib.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        self.check(self.lib.hipHostMalloc(C.byref(p), n, 0), "hipHostMalloc")
        buf = (C.c_char * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        self.lib.hipHostFree.argtypes = [C.c_void_p]
        self.check(self.lib.hipHostFree(C.c_void_p(self._pinned.pop(arr.ctypes.data))), "hipHostFree")

    def sync(self):
        self.check(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")
