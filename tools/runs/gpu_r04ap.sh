#!/bin/bash
# round 4: the driver-shaped bench line and the fp8 line on the final build
set -o pipefail
mkdir -p gpurun_out/r04ap
timeout -k 10 600 python bench.py > gpurun_out/r04ap/bench_n1.json 2> gpurun_out/r04ap/bench_n1.err || { tail -20 gpurun_out/r04ap/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ap/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', round(d['in_tolerance']['rtfx']), round(d['in_tolerance']['ms_per_step'],1), d['in_tolerance'].get('tokens_identical_to_exact_f32'), 'host_resident', round(d['host_resident']['rtfx']), 'batch1', round(d['batch1']['p95_ms_per_clip'],1))
P
timeout -k 10 500 python bench.py --precision fp8 --no-cpu-baseline --no-batch1 > gpurun_out/r04ap/bench_fp8.json 2> gpurun_out/r04ap/bench_fp8.err || { tail -20 gpurun_out/r04ap/bench_fp8.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ap/bench_fp8.json').read().strip().splitlines()[-1])
print('fp8', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
