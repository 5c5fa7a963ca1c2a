"""Device batches beyond 256 clips: rows of the same clip must decode identically wherever they sit in the batch, and equal the
same clip decoded in a 16-clip context (F32: exact arithmetic, so token equality is the bar)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from whisper_rust_ort_amd import binding as wb, modelspec as ms
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = {"f32": wb.WH_PREC_F32, "bf16": wb.WH_PREC_BF16, "fp8": wb.WH_PREC_FP8}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
model = wb.Model("synthetic:base:1234", 0, prec)
uniq = [ms.synth_clip(100 + i) for i in range(16)]
p = wb.DecodeParams([50258, 50259, 50359, 50363], 32, 50257, [50257])
small = wb.Context(model, 16)
ref = [t.tolist() for t in small.transcribe_batch(uniq, p)]
del small
ctx = wb.Context(model, B)
clips = [uniq[(i * 7) % 16] for i in range(B)]
t0 = time.time(); out = ctx.transcribe_batch(clips, p); dt = time.time() - t0
bad = sum(out[i].tolist() != ref[(i * 7) % 16] for i in range(B))
print(f"B={B}: {bad} of {B} rows differ from the 16-clip context ({dt:.2f} s)", ctx.timings())
sys.exit(1 if bad and prec == wb.WH_PREC_F32 else 0)
