#!/bin/bash
# round 4: folded LayerNorms on centred rows (encoder and decoder)
set -o pipefail
mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "folded_layernorm" > gpurun_out/r04p/pytest_fold.log 2>&1
grep -E "offset enc|passed|failed" gpurun_out/r04p/pytest_fold.log | head -20
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04p/pytest_all.log 2>&1; rc=$?
tail -3 gpurun_out/r04p/pytest_all.log; [ $rc -eq 0 ] || { tail -60 gpurun_out/r04p/pytest_all.log; exit $rc; }
for p in bf16 f16x3; do
timeout -k 10 600 python bench.py --precision $p --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04p/bench_$p.json 2> gpurun_out/r04p/bench_$p.err || { tail -5 gpurun_out/r04p/bench_$p.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r04p/bench_$p.json'))
print('$p', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
PY
done
