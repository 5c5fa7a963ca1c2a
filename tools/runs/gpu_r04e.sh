#!/bin/bash
# round 4: hardened parity tests (absolute bf16 bounds, full forced history at 2048 clips through the row selector) + the new bench line
set -o pipefail
mkdir -p gpurun_out/r04e
timeout -k 10 1100 python -m pytest tests/test_hip_parity.py tests/test_large_v3_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -s -k "bf16 or fp8 or batched or cross_es or gemm8_off or large_v3" > gpurun_out/r04e/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04e/pytest.log
grep -E "decided|max \||passed|failed|rc |clip vs|WH_GEMM8" gpurun_out/r04e/pytest.log | tail -30
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04e/pytest.log; exit $rc; }
timeout -k 10 900 python bench.py > gpurun_out/r04e/bench_default.json 2> gpurun_out/r04e/bench_default.err || { tail -20 gpurun_out/r04e/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/bench_default.json'))
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
for k in ('host_resident','in_tolerance','scaling_strong','scaling_weak','batch1','batch64'):
    print(k, json.dumps(d.get(k))[:600])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
