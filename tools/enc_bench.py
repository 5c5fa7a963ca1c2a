#!/usr/bin/env python3
"""Encoder-side micro benchmark: mel + encoder (+ cross-KV, 4 prompt positions) at bench.py's batch shape, event-timed per
kernel group.  Used while tuning the encoder GEMMs (not a judged number; bench.py is)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import binding as wb, modelspec as ms
import bench as B

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--preset", default="base")
ap.add_argument("--precision", default="bf16")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dims = ms.PRESETS[a.preset]
model = wb.Model(f"synthetic:{a.preset}:1234", 0, wb.PRECISIONS[a.precision])
ctx = wb.Context(model, a.clips)
hip = B.Hip()
pcm = np.stack([ms.synth_clip(i % 16) for i in range(a.clips)])
d_pcm = hip.upload(0, pcm)
prompt, eot = ([50258, 50259, 50359, 50363], 50257) if dims.vocab > 50400 else ([3, 5, 7, 9], 2)
params = wb.DecodeParams(prompt, 1, eot, suppress_tokens=[eot])
ctx.transcribe_batch_device(d_pcm, a.clips, params)
ctx.profile_enable(True)
best = None
for _ in range(a.reps):
    ctx.transcribe_batch_device(d_pcm, a.clips, params)
    g = ctx.profile_get()
    t = ctx.timings()
    if best is None or g["enc_gemm"]["ms"] < best[0]["enc_gemm"]["ms"]:
        best = (g, t)
g, t = best
w = B.algorithmic_work(dims, a.clips, 4, 1, 2)
attn_flop = dims.enc_layers * 4 * dims.n_audio_ctx ** 2 * dims.d_model * a.clips
gemm_flop = w["enc_flop_per_clip"] * a.clips - attn_flop
print({k: (round(v["ms"], 3), v["launches"]) for k, v in g.items()})
print(f"encode_s {t['encode_s']*1e3:.2f} ms; enc_gemm {gemm_flop / g['enc_gemm']['ms'] / 1e9:.0f} TF/s; enc_attn {attn_flop / g['enc_attn']['ms'] / 1e9:.0f} TF/s; "
      f"mel {w['mel_bytes_per_clip'] * a.clips / g['mel']['ms'] / 1e6:.0f} GB/s")
