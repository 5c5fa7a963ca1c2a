"""bench.py's in_tolerance object reports 511 / 512 clips token-identical between WH_PREC_F16X3 and the exact-f32 mode over 128 free-running tokens.
Where does the one clip fork, and how close were the two candidates?  Both precisions decode the same 512 benchmark clips; for every clip that
differs, the f32 tokens up to the fork are teacher-forced in both modes and the logits of the fork row are compared."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prompt, eot = [50258, 50259, 50359, 50363], 50257
clips = [ms.synth_clip(i) for i in range(n)]
params = wb.DecodeParams(prompt, 128, eot, [eot])
toks = {}
for name in ("f32", "f16x3"):
    ctx = wb.Context(wb.Model("synthetic:base:1234", 0, wb.PRECISIONS[name]), n)
    toks[name] = [t.tolist() for t in ctx.transcribe_batch(clips, params)]
    ctx.close()
bad = [i for i in range(n) if toks["f32"][i] != toks["f16x3"][i]]
print(f"{n} clips: {len(bad)} differ: {bad}")
for i in bad:
    a, b = toks["f32"][i], toks["f16x3"][i]
    k = next(j for j in range(len(a)) if a[j] != b[j])
    gen = k - len(prompt)                      # index of the generated token at which they fork
    rows = {}
    for name in ("f32", "f16x3"):
        one = wb.Context(wb.Model("synthetic:base:1234", 0, wb.PRECISIONS[name]), 1)
        one.run_encoder(one.whisper_log_mel(clips[i]), want_output=False)
        _, lg = one.greedy_decode_with_past(wb.DecodeParams(prompt, gen + 1, eot, [eot], forced=a[len(prompt):k]), want_logits=True)
        rows[name] = lg[gen]
        one.close()
    r32, rx = rows["f32"], rows["f16x3"]
    top = np.argsort(-r32)[:2]
    print(f"clip {i}: fork at generated token {gen}: f32 chose {a[k]}, f16x3 chose {b[k]}; f32 logits of the two candidates {r32[a[k]]:.6f} / {r32[b[k]]:.6f} "
          f"(margin {r32[a[k]] - r32[b[k]]:.2e}); f16x3 {rx[a[k]]:.6f} / {rx[b[k]]:.6f} (margin {rx[a[k]] - rx[b[k]]:.2e}); max |f16x3 - f32| over the row {np.abs(rx - r32).max():.2e}; "
          f"f32 top-2 ids {top.tolist()}")
