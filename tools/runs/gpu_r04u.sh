#!/bin/bash
# round 4: in-kernel stamps of k_gemm8 (tools/gemm8_stamps.hip)
set -o pipefail
mkdir -p gpurun_out/r04u
timeout -k 10 300 ./tools/gemm8_stamps > gpurun_out/r04u/stamps.txt 2>&1 || { tail -20 gpurun_out/r04u/stamps.txt; exit 1; }
cat gpurun_out/r04u/stamps.txt
