#!/bin/bash
# which processes land in the slow group of k_dec_cross_attn_es?  bench.py (context allocated before the PCM) vs the probe (PCM first), several processes each
set -o pipefail
mkdir -p gpurun_out/r04n
for i in 1 2 3 4 5 6; do
timeout -k 10 300 python bench.py --clips 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > gpurun_out/r04n/b$i.json 2> gpurun_out/r04n/b$i.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r04n/b$i.json')); print('bench $i', round(d['ms_per_step'],1), round(d['roofline']['avg_launch_us'],1), d['kernel_group_ms_per_step']['dec_cross_attn'])"
timeout -k 10 300 python tools/es_pitch_probe.py 2048 1 bf16 | cut -c1-90
done
