#!/bin/bash
# round 4: f16x3 bench processes in a row with the placement threshold of the 3-byte form at 0.75
set -o pipefail
mkdir -p gpurun_out/r04an
for i in 1 2 3; do
timeout -k 10 500 python bench.py --precision f16x3 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04an/bench_x3_$i.json 2> gpurun_out/r04an/bench_x3_$i.err || { tail -20 gpurun_out/r04an/bench_x3_$i.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04an/bench_x3_$i.json'))
print('process $i', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step']['dec_cross_attn'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
