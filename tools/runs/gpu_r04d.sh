#!/bin/bash
# round 4: every split-fp16 (f16x3) test incl. small presets (k_gemm<h2>, k_dec_gemm<h2>) and whisper-large-v3; then the whole GPU suite
set -o pipefail
mkdir -p gpurun_out/r04d
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04d/pytest_all.log 2>&1
rc=$?
echo "pytest rc $rc" >> gpurun_out/r04d/pytest_all.log
grep -E "f16x3|passed|failed|rc |Error" gpurun_out/r04d/pytest_all.log | tail -14
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04d/pytest_all.log; exit $rc; }
