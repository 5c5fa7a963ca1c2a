#!/bin/bash
# round 4: cross-attention on the encoder states as fp16 limb planes (k_dec_cross_attn_es2)
set -o pipefail
mkdir -p gpurun_out/r04c
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "(f16x3 and batched) or cross_mode_rule" > gpurun_out/r04c/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04c/pytest.log
grep -E "max \||passed|failed|rc |Error|error" gpurun_out/r04c/pytest.log | tail -12
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04c/pytest.log; exit $rc; }
for n in 512 2048; do
timeout -k 10 600 python bench.py --precision f16x3 --clips $n --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04c/bench_f16x3_b$n.json 2> gpurun_out/r04c/bench_f16x3_b$n.err || { tail -5 gpurun_out/r04c/bench_f16x3_b$n.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r04c/bench_f16x3_b$n.json'))
print($n, round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))
PY
done
