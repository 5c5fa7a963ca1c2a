"""k_dec_cross_attn_es launch times fall into two groups from context to context.  Is there a clip pitch (rows of padding
between the clips' encoder states) that is fast whatever physical memory the context got?  One process: the context is created several times
(each creation may land in either group); inside each context the pitch is swept with the probe hook wh_debug_set_es_pad."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WH_ES_PAD_MAX"] = "600"
from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
pads = [20, 0, 4, 12, 36, 52, 84, 148, 276, 532, 20]
model = wb.Model("synthetic:base:1234", 0, wb.PRECISIONS[prec])
lib = model.lib
lib.wh_debug_set_es_pad.argtypes = [C.c_void_p, C.c_int]
hip = wb.HipRuntime()
base = np.stack([ms.synth_clip(i) for i in range(8)])
d_pcm = hip.upload(0, np.concatenate([base] * (nb // 8)))
params = wb.DecodeParams([50258, 50259, 50359, 50363], 24, 50257, [50257])
for r in range(reps):
    ctx = wb.Context(model, nb)
    row = []
    for pad in pads:
        assert lib.wh_debug_set_es_pad(ctx.h, pad) == 0
        ctx.profile_enable(False)
        ctx.transcribe_batch_device(d_pcm, nb, params)
        ctx.profile_enable(["dec_cross_attn"])
        ctx.transcribe_batch_device(d_pcm, nb, params)
        pg = ctx.profile_get()["dec_cross_attn"]
        row.append(pg["ms"] / pg["launches"] * 1e3)
    print(f"context {r} ({prec}, {nb} clips): us per launch by pad " + "  ".join(f"{p}:{u:.0f}" for p, u in zip(pads, row)), flush=True)
    ctx.close()
