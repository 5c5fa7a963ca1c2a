#!/bin/bash
# round 4: the split-fp16 mode on fp16 + e4m3 encoder states — its tests, then the f16x3 bench line, A/B against WH_ES3=0 (two fp16 limbs)
set -o pipefail
mkdir -p gpurun_out/r04al
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py tests/test_large_v3_gpu.py -m gpu -x -q -s -k "f16x3 or split_fp16 or cross_mode or placement" > gpurun_out/r04al/pytest.log 2>&1; rc=$?
grep -E "es2 form|es3|passed|failed|max \|" gpurun_out/r04al/pytest.log | tail -14 | cut -c1-260
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04al/pytest.log; exit $rc; }
for v in 1 0; do
WH_ES3=$v timeout -k 10 500 python bench.py --precision f16x3 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04al/bench_x3_es3_$v.json 2> gpurun_out/r04al/bench_x3_es3_$v.err || { tail -20 gpurun_out/r04al/bench_x3_es3_$v.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04al/bench_x3_es3_$v.json'))
print('WH_ES3=$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
