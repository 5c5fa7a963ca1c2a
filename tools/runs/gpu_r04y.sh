#!/bin/bash
# round 4: decode tile GEMM column-tile width (WH_DEC_TILE_BN) in the f16x3 mode, 2048 clips: parity with the narrow tiles, then A/B
set -o pipefail
mkdir -p gpurun_out/r04y
WH_DEC_TILE_BN=33 timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "f16x3 and (1024 or 2048 or logit_bound)" > gpurun_out/r04y/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r04y/pytest.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04y/pytest.log; exit $rc; }
for v in 0 32 33 64; do
WH_DEC_TILE_BN=$v timeout -k 10 500 python bench.py --precision f16x3 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04y/bench_bn$v.json 2> gpurun_out/r04y/bench_bn$v.err || { tail -20 gpurun_out/r04y/bench_bn$v.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04y/bench_bn$v.json'))
print('bn=$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
P
done
