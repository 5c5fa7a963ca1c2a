#!/bin/bash
# round 4: the workspace placement step — its test, then four bench.py processes in a row (the alternating states of tools/runs/gpu_r04y.sh) with it on
set -o pipefail
mkdir -p gpurun_out/r04aa
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "placement" > gpurun_out/r04aa/pytest.log 2>&1; rc=$?
grep -E "placement step|passed|failed" gpurun_out/r04aa/pytest.log | tail -4
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04aa/pytest.log; exit $rc; }
for i in 1 2 3 4 5 6; do
timeout -k 10 500 python bench.py --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04aa/bench_$i.json 2> gpurun_out/r04aa/bench_$i.err || { tail -20 gpurun_out/r04aa/bench_$i.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04aa/bench_$i.json'))
print('process $i', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
for i in 1 2; do
timeout -k 10 500 python bench.py --precision f16x3 --clips 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04aa/bench_x3_$i.json 2> gpurun_out/r04aa/bench_x3_$i.err || { tail -20 gpurun_out/r04aa/bench_x3_$i.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04aa/bench_x3_$i.json'))
print('f16x3 process $i', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
