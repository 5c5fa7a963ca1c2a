#!/bin/bash
# round 4: the driver-shaped bench line with the fp8 side object
set -o pipefail
mkdir -p gpurun_out/r04at
timeout -k 10 600 python bench.py > gpurun_out/r04at/bench_n1.json 2> gpurun_out/r04at/bench_n1.err || { tail -20 gpurun_out/r04at/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04at/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', round(d['in_tolerance']['rtfx']), 'fp8', json.dumps(d['fp8'])[:260])
P
