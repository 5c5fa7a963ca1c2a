#!/bin/bash
# round 4, first GPU call: the split-fp16 mode (WH_PREC_F16X3) against the golden vectors + a first timing
set -o pipefail
mkdir -p gpurun_out/r04a
python -m pytest tests/test_hip_parity.py tests/test_large_v3_gpu.py -m gpu -x -q -s -k "f16x3 or test_f32_full_path or test_f32_matches_golden or test_f32_whisper_base" > gpurun_out/r04a/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
timeout -k 10 600 python bench.py --precision f16x3 --clips 512 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04a/bench_f16x3_b512.json 2> gpurun_out/r04a/bench_f16x3_b512.err
echo "bench rc $?"
tail -c 600 gpurun_out/r04a/bench_f16x3_b512.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a/bench_f16x3_b512.json'))
print(d['value'], d['ms_per_step'], d['kernel_group_ms_per_step'], d['roofline']['frac'])
PY
