#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04i
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "f16x3 and (whisper_base or full_path or (batched and 256))" > gpurun_out/r04i/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04i/pytest.log
grep -E "max \||passed|failed|rc " gpurun_out/r04i/pytest.log | tail -6
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04i/pytest.log; exit $rc; }
for w in 1 0; do
WH_ENC_ATTN_W4=$w timeout -k 10 600 python bench.py --precision f16x3 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04i/bench_f16x3_w4_$w.json 2> gpurun_out/r04i/bench_f16x3_w4_$w.err || { tail -5 gpurun_out/r04i/bench_f16x3_w4_$w.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r04i/bench_f16x3_w4_$w.json'))
print('w4=$w', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
PY
done
