#!/bin/bash
# round 4: full GPU suite + the driver-shaped bench on the build with k_enc_mlp
set -o pipefail
mkdir -p gpurun_out/r04x
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04x/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04x/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04x/pytest.log; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/r04x/bench_n1.json 2> gpurun_out/r04x/bench_n1.err || { tail -20 gpurun_out/r04x/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04x/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], d['roofline']['frac'], d.get('in_tolerance',{}).get('value'), d.get('host_resident',{}).get('value'))
P
