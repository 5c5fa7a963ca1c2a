#!/bin/bash
# round 4, final records: full GPU suite, smoke, the driver-shaped bench line, f16x3 kernel trace + HBM counters (k_dec_cross_attn_es3)
set -o pipefail
mkdir -p gpurun_out/r04am
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04am/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04am/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04am/pytest.log; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > gpurun_out/r04am/bench_n1.json 2> gpurun_out/r04am/bench_n1.err || { tail -20 gpurun_out/r04am/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04am/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', round(d['in_tolerance']['rtfx']), round(d['in_tolerance']['ms_per_step'],1), d['in_tolerance'].get('tokens_identical_to_exact_f32'), 'host_resident', round(d['host_resident']['rtfx']))
P
bash profiles/collect.sh f16x3 r04 2048 base > gpurun_out/r04am/collect_f16x3.log 2>&1 || { tail -30 gpurun_out/r04am/collect_f16x3.log; exit 1; }
tail -32 gpurun_out/r04am/collect_f16x3.log | head -12
tail -8 gpurun_out/r04am/collect_f16x3.log
