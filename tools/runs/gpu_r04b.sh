#!/bin/bash
# round 4: decoder operands as fp16 limbs (h2) + register-budgeted decode GEMMs
set -o pipefail
mkdir -p gpurun_out/r04b
python -m pytest tests/test_hip_parity.py tests/test_large_v3_gpu.py -m gpu -x -q -s -k "f16x3 or test_f32_full_path or test_f32_matches_golden or test_f32_whisper_base or test_batch_equals or test_longform_windows or test_eot_stops" > gpurun_out/r04b/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04b/pytest.log
grep -E "max \||passed|failed|rc " gpurun_out/r04b/pytest.log | tail -12
[ $rc -eq 0 ] || exit $rc
for p in f16x3 f32; do
timeout -k 10 600 python bench.py --precision $p --clips 512 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04b/bench_${p}_b512.json 2> gpurun_out/r04b/bench_${p}_b512.err || exit 1
python - <<PY
import json
d=json.load(open('gpurun_out/r04b/bench_${p}_b512.json'))
print('$p', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
PY
done
