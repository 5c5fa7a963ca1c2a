"""The two launch-time states of the encoder-state cross-attention against WHERE the context's workspace lies: one
process; the context is created and timed, then — while it still exists — a placeholder of `hold` GB is allocated, the context destroyed and
created again (its workspace cannot take the same memory), the placeholder freed, and the new context timed.  Consecutive processes of one
box alternate between the states (tools/runs/gpu_r04y.sh), which points at the physical placement; this asks whether a process can move.

    python tools/es_place_probe.py [clips=2048] [rounds=4] [hold_gb=80] [precision=bf16]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hold_gb = float(sys.argv[3]) if len(sys.argv) > 3 else 80.0
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
model = wb.Model("synthetic:base:1234", 0, wb.PRECISIONS[prec])
hip = wb.HipRuntime()
base = np.stack([ms.synth_clip(i) for i in range(8)])
d_pcm = hip.upload(0, np.concatenate([base] * (nb // 8)))
params = wb.DecodeParams([50258, 50259, 50359, 50363], 32, 50257, [50257])


def timed(ctx, tag):
    ctx.transcribe_batch_device(d_pcm, nb, params)
    ctx.profile_enable(["dec_cross_attn"], stride=4)
    t0 = time.perf_counter()
    ctx.transcribe_batch_device(d_pcm, nb, params)
    dt = time.perf_counter() - t0
    pg = ctx.profile_get()["dec_cross_attn"]
    us = pg["ms"] / pg["launches"] * 1e3
    print(f"{tag}: cross-attention {us:.1f} us per launch (step with 32 tokens {dt * 1e3:.0f} ms)", flush=True)
    return us


ctx = wb.Context(model, nb)
timed(ctx, "context 0 (first allocation of the process)")
for r in range(1, rounds + 1):
    hold = hip.malloc(0, int(hold_gb * (1 << 30)))   # taken while the old workspace is still there
    ctx.close()
    ctx = wb.Context(model, nb)                         # ... so the new one lies somewhere else
    hip.free(hold)
    timed(ctx, f"context {r} (re-created beside a {hold_gb:.0f} GB placeholder)")
ctx.close()
