#!/bin/bash
# round 4, records 2/2: the bench lines of the end-of-round build
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04r; mkdir -p $O
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/$name.json'))
print('$name', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['roofline']['kernel'], (d.get('in_tolerance') or {}).get('rtfx'), (d.get('host_resident') or {}).get('rtfx'))
PY
}
run bench_n1
run bench_n1_f16x3 --precision f16x3 --no-cpu-baseline
run bench_n1_fp8 --precision fp8 --no-cpu-baseline
run bench_n1_f32 --precision f32 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1
run bench_large_v3 --preset large-v3 --steps 2 --warmup 1 --no-cpu-baseline
run bench_large_v3_f16x3 --preset large-v3 --precision f16x3 --steps 2 --warmup 1 --no-cpu-baseline
