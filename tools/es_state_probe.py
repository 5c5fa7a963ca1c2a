"""Do the two launch-time groups of k_dec_cross_attn_es follow the process or the context's allocation?  One process, the
same model and device-resident PCM; the context is created, timed and destroyed several times (optionally with a dummy allocation of a
varying size made first, so that the workspace lands elsewhere)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
hip = wb.HipRuntime()
base = np.stack([ms.synth_clip(i) for i in range(8)])
pcm = np.concatenate([base] * (nb // 8))
d_pcm = hip.upload(0, pcm)
params = wb.DecodeParams([50258, 50259, 50359, 50363], 128, 50257, [50257])
for r in range(reps):
    pad = hip.upload(0, np.zeros((r * 37 + 1) * (1 << 20), np.uint8)) if len(sys.argv) > 3 else None
    ctx = wb.Context(model, nb)
    ctx.transcribe_batch_device(d_pcm, nb, params)
    ctx.profile_enable(["dec_cross_attn"], stride=16)
    t0 = time.perf_counter()
    ctx.transcribe_batch_device(d_pcm, nb, params)
    ctx.transcribe_batch_device(d_pcm, nb, params)
    dt = (time.perf_counter() - t0) / 2
    pg = ctx.profile_get()["dec_cross_attn"]
    print(f"context {r}: step {dt * 1e3:.1f} ms, k_dec_cross_attn_es {pg['ms'] / pg['launches'] * 1e3:.1f} us per launch", flush=True)
    ctx.close()
    if pad is not None:
        hip.free(pad)
