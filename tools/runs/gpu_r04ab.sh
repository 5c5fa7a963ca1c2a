#!/bin/bash
# round 4, end-of-round records on the final build: full GPU suite, the driver-shaped bench line, kernel trace + HBM counters (bf16, f16x3) and MFMA utilisation (bf16) at 2048 clips
set -o pipefail
mkdir -p gpurun_out/r04ab
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04ab/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04ab/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04ab/pytest.log; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/r04ab/bench_n1.json 2> gpurun_out/r04ab/bench_n1.err || { tail -20 gpurun_out/r04ab/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ab/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', round(d['in_tolerance']['rtfx']), round(d['in_tolerance']['ms_per_step'],1), 'host_resident', round(d['host_resident']['rtfx']))
P
for p in bf16 f16x3; do
bash profiles/collect.sh $p r04 2048 base > gpurun_out/r04ab/collect_$p.log 2>&1 || { tail -30 gpurun_out/r04ab/collect_$p.log; exit 1; }
tail -32 gpurun_out/r04ab/collect_$p.log | head -26
done
bash profiles/collect_mfma.sh bf16 r04 2048 base > gpurun_out/r04ab/mfma_bf16.log 2>&1 || { tail -30 gpurun_out/r04ab/mfma_bf16.log; exit 1; }
tail -16 gpurun_out/r04ab/mfma_bf16.log
