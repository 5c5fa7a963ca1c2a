#!/bin/bash
# round 4: final build — full GPU suite, smoke, the driver-shaped bench line
set -o pipefail
mkdir -p gpurun_out/r04ag
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04ag/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04ag/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04ag/pytest.log; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r04ag/bench_n1.json 2> gpurun_out/r04ag/bench_n1.err || { tail -20 gpurun_out/r04ag/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ag/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', round(d['in_tolerance']['rtfx']), round(d['in_tolerance']['ms_per_step'],1), d['in_tolerance'].get('tokens_identical_to_exact_f32'), 'host_resident', round(d['host_resident']['rtfx']))
P
