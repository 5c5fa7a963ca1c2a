#!/bin/bash
# round 4: LDS-DMA issue moved behind the MFMAs in the tile GEMMs (k_gemm8, k_gemm8x, k_lm_head_tile*, k_dec_tile)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_fp8_gpu.py -m gpu -x -q -k "whisper_base or (batched and (256 or 2048)) or wide_batch or logit_bound or bf16_teacher or mx_kernels" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; [ $rc -eq 0 ] || { tail -60 $O/pytest.log; exit $rc; }
for p in bf16 f16x3 fp8; do
timeout -k 10 600 python bench.py --precision $p --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > $O/bench_$p.json 2> $O/bench_$p.err || { tail -5 $O/bench_$p.err; exit 1; }
python - <<PY
import json
d=json.load(open('gpurun_out/r04l/bench_$p.json'))
print('$p', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
PY
done
