"""Cross-attention on the encoder states (wh_cross_es.hip) against the projected-K/V kernels and the f32 golden vectors:
teacher-forced logits of the same clips on two contexts of one bf16 whisper-base model."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g0 = np.load(os.path.join(ROOT, "tests", "golden", "base_s1234_c0.npz"))
prompt, eot = g0["prompt"].tolist(), int(g0["eot"])
forced = g0["forced_c"].tolist()
if nb > 256:
    forced = forced[:7]
t0 = time.time()
model = wb.Model("synthetic:base:1234", 0, wb.WH_PREC_BF16)
print(f"model build {time.time() - t0:.1f} s", flush=True)
clips = [ms.synth_clip(0), ms.synth_clip(3)] + [ms.synth_clip(300 + i) for i in range(30)]
clips = [clips[i % 32] for i in range(nb)]
fp = wb.DecodeParams(prompt, len(forced) + 1, eot, forced=forced)
res = {}
for name, es in (("es", True), ("kv", False)):
    ctx = wb.Context(model, nb, cross_es=es)
    assert ctx.cross_mode == int(es)
    ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 2, eot, [eot]))
    t, l = ctx.greedy_decode_resident_batch(fp, want_logits=True)
    res[name] = (t, np.stack(l[:32]))
    free = ctx.transcribe_batch(clips, wb.DecodeParams(prompt, 32, eot, [eot]))
    res[name + "_free"] = [x.tolist() for x in free[:32]]
    del ctx
les, lkv = res["es"][1], res["kv"][1]
d = np.abs(les - lkv).max(axis=(1, 2))
print(f"es vs kv: max |dlogit| per clip: max {d.max():.4f} median {np.median(d):.4f}; logit std {lkv.std():.3f}")
for name in ("es", "kv"):
    l = res[name][1]
    e = [np.abs(l[0][i][g0["top_ids_c"][i]] - g0["top_vals_c"][i]).max() for i in range(len(forced) + 1)]
    print(f"{name} vs f32 golden (clip 0, teacher-forced): max {max(e):.4f} mean {np.mean(e):.4f}")
same = sum(a == b for a, b in zip(res["es_free"], res["kv_free"]))
print(f"free-running 32 tokens: {same}/32 clips identical between the two modes")
assert np.isfinite(les).all()
assert d.max() < 0.2
