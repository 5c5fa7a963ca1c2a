#!/usr/bin/env python3
"""Encoder-dominated run for counter collection: N clips, 1 new token (mel + encoder + cross-KV + one decoder step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisper_rust_ort_amd import binding as wb, modelspec as ms
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = wb.PRECISIONS[sys.argv[2]] if len(sys.argv) > 2 else wb.WH_PREC_BF16
model = wb.Model("synthetic:base:1234", 0, prec)
ctx = wb.Context(model, n)
clips = [ms.synth_clip(i) for i in range(n)]
p = wb.DecodeParams([50258, 50259, 50359, 50363], 1, 50257, [50257])
for _ in range(2):
    out = ctx.transcribe_batch(clips, p)
print("ok", len(out))
