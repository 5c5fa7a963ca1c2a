#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04g
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "wide_batch or (f16x3 and (batched or whisper_base))" > gpurun_out/r04g/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04g/pytest.log
grep -E "max \||passed|failed|rc " gpurun_out/r04g/pytest.log | tail -12
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04g/pytest.log; exit $rc; }
timeout -k 10 600 python bench.py --precision f16x3 --clips 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04g/bench_f16x3_b2048.json 2> gpurun_out/r04g/bench_f16x3_b2048.err || { tail -5 gpurun_out/r04g/bench_f16x3_b2048.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench_f16x3_b2048.json'))
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))
PY
