"""Model geometry, canonical tensor table and the hash-seeded synthetic weights.

No Whisper checkpoint exists in this pipeline (reference `.gitignore:13-14` keeps the
ONNX/weight directories out of the tree, and there is no network), so every run uses
weights defined by an integer hash of (tensor name, seed, flat index).  The same
definition is implemented in C++ inside the HIP library (`csrc/wh_weights.cpp`); the
test-suite checks the two bit-for-bit.

Tensor names follow the HF `WhisperForConditionalGeneration` state-dict
(`model.encoder.conv1.weight`, `model.decoder.layers.N.encoder_attn.k_proj.weight`, ...),
i.e. the layout of a real `model.safetensors`, which the library's loader also accepts.
Those graphs are what the reference executes through ONNX Runtime
(reference `src/main.rs:1099-1108`, exported by `scripts/export_onnx_whisper.py:19-28`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Iterator, List, Tuple

import numpy as np

N_FRAMES = 3000  # 30 s window, hop 160 @16 kHz (reference src/main.rs:896-905)


@dataclass(frozen=True)
class WhisperDims:
    n_mels: int
    d_model: int
    n_heads: int
    enc_layers: int
    dec_layers: int
    ffn: int
    vocab: int
    n_audio_ctx: int = 1500
    n_text_ctx: int = 448

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    def as_dict(self) -> dict:
        return asdict(self)


PRESETS: Dict[str, WhisperDims] = {
    # small enough that the scalar C oracle finishes in well under a second
    "nano": WhisperDims(n_mels=80, d_model=128, n_heads=2, enc_layers=2, dec_layers=2,
                        ffn=256, vocab=1024),
    # same shape family as base but ~10x cheaper: used for wider parity sweeps
    "micro": WhisperDims(n_mels=80, d_model=256, n_heads=4, enc_layers=2, dec_layers=3,
                         ffn=1024, vocab=4099),
    # openai/whisper-base
    "base": WhisperDims(n_mels=80, d_model=512, n_heads=8, enc_layers=6, dec_layers=6,
                        ffn=2048, vocab=51865),
    # openai/whisper-large-v3 (128 mel bins, vocab 51866)
    "large-v3": WhisperDims(n_mels=128, d_model=1280, n_heads=20, enc_layers=32,
                            dec_layers=32, ffn=5120, vocab=51866),
}


def tensor_table(dims: WhisperDims) -> List[Tuple[str, Tuple[int, ...]]]:
    """Canonical (name, shape) list. The flat weight blob used by the oracle and the
    library's internal arena both follow exactly this order."""
    d, F = dims.d_model, dims.ffn
    t: List[Tuple[str, Tuple[int, ...]]] = []

    def attn(prefix: str) -> None:
        t.append((f"{prefix}.q_proj.weight", (d, d)))
        t.append((f"{prefix}.q_proj.bias", (d,)))
        t.append((f"{prefix}.k_proj.weight", (d, d)))  # k_proj has no bias
        t.append((f"{prefix}.v_proj.weight", (d, d)))
        t.append((f"{prefix}.v_proj.bias", (d,)))
        t.append((f"{prefix}.out_proj.weight", (d, d)))
        t.append((f"{prefix}.out_proj.bias", (d,)))

    def ln(prefix: str) -> None:
        t.append((f"{prefix}.weight", (d,)))
        t.append((f"{prefix}.bias", (d,)))

    def mlp(prefix: str) -> None:
        t.append((f"{prefix}.fc1.weight", (F, d)))
        t.append((f"{prefix}.fc1.bias", (F,)))
        t.append((f"{prefix}.fc2.weight", (d, F)))
        t.append((f"{prefix}.fc2.bias", (d,)))

    e = "model.encoder"
    t.append((f"{e}.conv1.weight", (d, dims.n_mels, 3)))
    t.append((f"{e}.conv1.bias", (d,)))
    t.append((f"{e}.conv2.weight", (d, d, 3)))
    t.append((f"{e}.conv2.bias", (d,)))
    t.append((f"{e}.embed_positions.weight", (dims.n_audio_ctx, d)))
    for i in range(dims.enc_layers):
        p = f"{e}.layers.{i}"
        attn(f"{p}.self_attn")
        ln(f"{p}.self_attn_layer_norm")
        mlp(p)
        ln(f"{p}.final_layer_norm")
    ln(f"{e}.layer_norm")

    dd = "model.decoder"
    t.append((f"{dd}.embed_tokens.weight", (dims.vocab, d)))
    t.append((f"{dd}.embed_positions.weight", (dims.n_text_ctx, d)))
    for i in range(dims.dec_layers):
        p = f"{dd}.layers.{i}"
        attn(f"{p}.self_attn")
        ln(f"{p}.self_attn_layer_norm")
        attn(f"{p}.encoder_attn")
        ln(f"{p}.encoder_attn_layer_norm")
        mlp(p)
        ln(f"{p}.final_layer_norm")
    ln(f"{dd}.layer_norm")
    return t


def n_params(dims: WhisperDims) -> int:
    return sum(int(np.prod(s)) for _, s in tensor_table(dims))


# ----------------------------------------------------------------------------------
# hash-seeded values
# ----------------------------------------------------------------------------------
_GOLD = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix_u24(key: int, n: int) -> np.ndarray:
    """24 uniform bits per flat index i: splitmix64 finaliser of key + i*GOLD."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        z = np.uint64(key) + i * np.uint64(_GOLD)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        return (z >> np.uint64(40)).astype(np.uint32)


def sinusoid_positions(length: int, channels: int) -> np.ndarray:
    """[3P] transformers modeling_whisper.py `sinusoids` (:55-64): [sin | cos] halves,
    timescale increment log(10000)/(channels/2 - 1).  Evaluated in float64, stored f32."""
    half = channels // 2
    inc = math.log(10000.0) / (half - 1)
    inv = np.exp(-inc * np.arange(half, dtype=np.float64))
    st = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)


# Scales chosen so that a random-weight model still behaves like a language model under greedy decoding
# (SURVEY §7 hard part 1): with unit-gain layers and a large tied embedding the residual stream is dominated
# by the input token's own embedding, the tied LM head then picks that token again and every free-running
# stream degenerates into one repeated id.  Small embeddings, residual-branch gain 4 in the decoder, q/k
# projections x2.5 (peaked, context-dependent attention) and a final-LayerNorm gain of 2 (logit std ~1.3) give
# ~26 distinct ids per 48 generated tokens at whisper-base size with top-1 margins >= 1e-2, while two f32
# implementations (this repo's oracle vs HF eager) still agree to ~1e-5 on the logits.  q/k x3 is already in a
# chaotic regime: near-one-hot softmaxes amplify f32 summation-order noise to ~1e-3 on the logits.
RES_GAIN, QK_GAIN, FINAL_LN_GAIN = 4.0, 2.5, 2.0


def value_rule(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(offset, amplitude): value = offset + amplitude * U[-1,1)."""
    if name == "model.decoder.layer_norm.weight":
        return FINAL_LN_GAIN, 0.2
    if name.endswith("layer_norm.weight"):
        return 1.0, 0.1
    if name.endswith(".bias"):
        return 0.0, 0.1
    if name.endswith("embed_tokens.weight"):
        return 0.0, 0.05
    if name.endswith("decoder.embed_positions.weight"):
        return 0.0, 0.05
    fan_in = int(np.prod(shape[1:]))
    gain = 1.0
    if name.startswith("model.decoder.") and (name.endswith("out_proj.weight") or name.endswith("fc2.weight")):
        gain = RES_GAIN
    elif name.endswith("q_proj.weight") or name.endswith("k_proj.weight"):
        gain = QK_GAIN
    return 0.0, float(np.float32(math.sqrt(3.0 / fan_in) * gain))


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> np.ndarray:
    if name.endswith("encoder.embed_positions.weight"):
        return sinusoid_positions(shape[0], shape[1])
    n = int(np.prod(shape))
    key = (fnv1a64(name) ^ ((seed * _GOLD) & _M64)) & _M64
    u = _splitmix_u24(key, n)
    off, amp = value_rule(name, shape)
    # (u - 2^23) is exact in f32; one rounding in the multiply, one in the add
    v = (u.astype(np.float32) - np.float32(8388608.0)) * (np.float32(amp) / np.float32(8388608.0))
    if off != 0.0:
        v = np.float32(off) + v
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(dims: WhisperDims, seed: int) -> Dict[str, np.ndarray]:
    return {n: synth_tensor(n, s, seed) for n, s in tensor_table(dims)}


def flatten_state_dict(dims: WhisperDims, sd: Dict[str, np.ndarray]) -> np.ndarray:
    """One contiguous f32 blob in `tensor_table` order (what `oracle/` consumes)."""
    parts = []
    for name, shape in tensor_table(dims):
        a = np.ascontiguousarray(sd[name], dtype=np.float32)
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def iter_offsets(dims: WhisperDims) -> Iterator[Tuple[str, Tuple[int, ...], int]]:
    off = 0
    for name, shape in tensor_table(dims):
        yield name, shape, off
        off += int(np.prod(shape))


# ----------------------------------------------------------------------------------
# fp8 weights: OCP e4m3fn, one f32 scale per output channel, for the Linear /
# QKV weights only — the scope of the reference's weights-only MatMul/Gemm quantisation
# (quantize_onnx_int8.py:37-42); convolutions, LayerNorms, biases and both embeddings stay as they are.
# Bit-identical twins: wh_quantize_e4m3 / wh_dequantize_e4m3 in csrc/wh_model.cpp, orc_e4m3_* in oracle/.
# ----------------------------------------------------------------------------------
E4M3_MAX = 448.0


def is_linear_weight(name: str) -> bool:
    return name.endswith("_proj.weight") or name.endswith(".fc1.weight") or name.endswith(".fc2.weight")


def quantize_e4m3(x: np.ndarray) -> np.ndarray:
    """f32 -> e4m3fn code (uint8): round to nearest even, saturating at +-448, NaN -> 0x7F."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    sign = ((x.view(np.uint32) >> np.uint32(24)) & np.uint32(0x80)).astype(np.uint8)
    a = np.minimum(np.abs(x), np.float32(E4M3_MAX))
    u = a.view(np.uint32).astype(np.uint64)
    r = u + np.uint64(0x7FFFF) + ((u >> np.uint64(20)) & np.uint64(1))          # RNE at mantissa bit 20
    normal = (((r >> np.uint64(23)) - np.uint64(120)) << np.uint64(3)) | ((r >> np.uint64(20)) & np.uint64(7))
    sub = np.rint(np.nan_to_num(a).astype(np.float64) * 512.0).astype(np.uint64)              # multiples of 2^-9; 8 = first normal
    code = np.where(a >= np.float32(2.0 ** -6), np.minimum(normal, np.uint64(0x7E)), sub).astype(np.uint8)
    code = np.where(np.isnan(x), np.uint8(0x7F), code)
    return (code | sign).astype(np.uint8)


def dequantize_e4m3(code: np.ndarray) -> np.ndarray:
    c = np.asarray(code, dtype=np.uint8).astype(np.int32)
    e, m = (c >> 3) & 15, c & 7
    mag = np.where(e == 0, m * 2.0 ** -9, (8 + m) * np.exp2(e.astype(np.float64) - 10.0))
    mag = np.where((c & 0x7F) == 0x7F, np.nan, mag)
    return np.where(c & 0x80, -mag, mag).astype(np.float32)


def quantize_linear(w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """[N, K] f32 -> (codes uint8 [N, K], scale f32 [N]); scale[n] = max|w[n, :]| / 448 (1 for a zero row)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    amax = np.max(np.abs(w), axis=1)
    scale = np.where(amax > 0, amax / np.float32(E4M3_MAX), np.float32(1.0)).astype(np.float32)
    return quantize_e4m3(w / scale[:, None]), scale


def fake_quant_state_dict(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Every Linear weight replaced by dequant(quant(w)) in f32 — what an fp8-weight model computes with."""
    out = {}
    for name, w in sd.items():
        if is_linear_weight(name):
            q, s = quantize_linear(w)
            out[name] = (dequantize_e4m3(q) * s[:, None]).astype(np.float32)
        else:
            out[name] = w
    return out


# ----------------------------------------------------------------------------------
# synthetic audio
# ----------------------------------------------------------------------------------
def synth_clip(index: int, base_seed: int = 1000, n: int = 480000) -> np.ndarray:
    """Clip i: three enveloped sinusoids (80-4000 Hz) + N(0, 0.02^2) noise, 16 kHz mono,
    clipped to [-1, 1].  Deterministic for a given numpy version's PCG64."""
    rng = np.random.Generator(np.random.PCG64(base_seed + index))
    t = np.arange(n, dtype=np.float64) / 16000.0
    f = rng.uniform(80.0, 4000.0, size=3)
    ph = rng.uniform(0.0, 2 * math.pi, size=3)
    env = 0.5 - 0.5 * np.cos(2 * math.pi * 4.0 * t)
    x = 0.25 * sum(np.sin(2 * math.pi * f[j] * t + ph[j]) for j in range(3)) * env
    x = x + 0.02 * rng.standard_normal(n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)
