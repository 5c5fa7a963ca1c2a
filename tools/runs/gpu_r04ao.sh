#!/bin/bash
# round 4: fp8 bench line with one / two loader waves in k_dec_cross_attn_es8 (one call)
set -o pipefail
mkdir -p gpurun_out/r04ao
for nl in 1 2 1 2; do
WH_ES8_LOADERS=$nl timeout -k 10 500 python bench.py --precision fp8 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04ao/bench_fp8_nl$nl.json 2> gpurun_out/r04ao/bench_fp8_nl$nl.err || { tail -20 gpurun_out/r04ao/bench_fp8_nl$nl.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04ao/bench_fp8_nl$nl.json'))
print('loaders=$nl', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step']['dec_cross_attn'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
