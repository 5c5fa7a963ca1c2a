#!/bin/bash
# round 4: decode GEMMs as LDS-DMA tile GEMMs on contexts of >= 1024 clips (k_dec_tile), A/B against k_dec_gemm_wide
set -o pipefail
mkdir -p gpurun_out/r04h
timeout -k 10 1000 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "(batched and (1024 or 2048)) or (logit_bound and (1024 or 2048)) or largest_batch or (device_entry and 1024)" > gpurun_out/r04h/pytest.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04h/pytest.log
grep -E "max \||passed|failed|rc |clip vs" gpurun_out/r04h/pytest.log | tail -14
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04h/pytest.log; exit $rc; }
for p in bf16 f16x3; do for t in 0 1; do
WH_DEC_TILE=$t timeout -k 10 600 python bench.py --precision $p --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04h/bench_${p}_tile$t.json 2> gpurun_out/r04h/bench_${p}_tile$t.err || { tail -5 gpurun_out/r04h/bench_${p}_tile$t.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r04h/bench_${p}_tile$t.json'))
print('$p tile=$t', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
PY
done; done
