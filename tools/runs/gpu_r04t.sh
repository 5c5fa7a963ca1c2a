#!/bin/bash
# round 4: the two waves of a SIMD in opposite k-step orders (k_gemm8): harness (host-reference check + ablations), then the bench
set -o pipefail
mkdir -p gpurun_out/r04t
timeout -k 10 300 ./tools/gemm8_ablate > gpurun_out/r04t/ablate.txt 2>&1 || { tail -20 gpurun_out/r04t/ablate.txt; exit 1; }
cut -c1-460 gpurun_out/r04t/ablate.txt | tail -30
timeout -k 10 600 python bench.py --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04t/bench.json 2> gpurun_out/r04t/bench.err || { tail -20 gpurun_out/r04t/bench.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/r04t/bench.json'))
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
P
