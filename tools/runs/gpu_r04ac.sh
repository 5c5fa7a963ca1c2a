#!/bin/bash
# round 4, end-of-round records 2/2 (after the sticky-error fix of the placement step): its test, the driver-shaped bench line, f16x3 trace + counters, bf16 MFMA utilisation
set -o pipefail
mkdir -p gpurun_out/r04ac
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "placement" > gpurun_out/r04ac/pytest.log 2>&1; rc=$?
grep -E "placement step|passed|failed" gpurun_out/r04ac/pytest.log | tail -4
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04ac/pytest.log; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/r04ac/bench_n1.json 2> gpurun_out/r04ac/bench_n1.err || { tail -20 gpurun_out/r04ac/bench_n1.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ac/bench_n1.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
print('in_tolerance', json.dumps(d['in_tolerance'])[:300])
print('host_resident', round(d['host_resident']['rtfx']))
P
bash profiles/collect.sh f16x3 r04 2048 base > gpurun_out/r04ac/collect_f16x3.log 2>&1 || { tail -30 gpurun_out/r04ac/collect_f16x3.log; exit 1; }
tail -32 gpurun_out/r04ac/collect_f16x3.log | head -26
bash profiles/collect_mfma.sh bf16 r04 2048 base > gpurun_out/r04ac/mfma_bf16.log 2>&1 || { tail -30 gpurun_out/r04ac/mfma_bf16.log; exit 1; }
tail -16 gpurun_out/r04ac/mfma_bf16.log
for p in f16x3 fp8; do
timeout -k 10 500 python bench.py --precision $p --no-cpu-baseline --no-batch1 > gpurun_out/r04ac/bench_$p.json 2> gpurun_out/r04ac/bench_$p.err || { tail -20 gpurun_out/r04ac/bench_$p.err; exit 1; }
python - <<P
import json
d=json.loads(open('gpurun_out/r04ac/bench_$p.json').read().strip().splitlines()[-1])
print('$p', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
