#!/bin/bash
# round 4: the fp8 mode on e4m3 encoder states — its tests, the other fp8 tests, the cross-mode rule, then the fp8 bench line (A/B: WH_NO_ES8=1 = the K / V form)
set -o pipefail
mkdir -p gpurun_out/r04ai
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_hip_parity.py -m gpu -x -q -s -k "fp8 or cross_mode" > gpurun_out/r04ai/pytest.log 2>&1; rc=$?
grep -E "fp8|passed|failed|Error" gpurun_out/r04ai/pytest.log | tail -16 | cut -c1-300
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04ai/pytest.log; exit $rc; }
for v in 0 1; do
if [ $v = 1 ]; then export WH_NO_ES8=1; fi
timeout -k 10 500 python bench.py --precision fp8 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04ai/bench_fp8_noes$v.json 2> gpurun_out/r04ai/bench_fp8_noes$v.err || { tail -20 gpurun_out/r04ai/bench_fp8_noes$v.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04ai/bench_fp8_noes$v.json'))
print('WH_NO_ES8=$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
