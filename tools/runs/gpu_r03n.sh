#!/bin/bash
# round 3, end: the default bench line (with the CPU leg), kernel trace + HBM counters + MFMA-utilisation counters of the same build
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-200 $O/bench_default.json
timeout -k 10 400 python bench.py --precision fp8 --no-cpu-baseline > $O/bench_fp8.json 2> $O/bench_fp8.err; echo "bench fp8 rc $?"; cut -c1-200 $O/bench_fp8.json
bash profiles/collect.sh bf16 r03 1024 base > gpurun_out/collect_r03_base_bf16.log 2>&1; echo "collect base bf16 rc $?"; tail -24 gpurun_out/collect_r03_base_bf16.log | cut -c1-170
cp gpurun_out/prof_r03_base_bf16_b1024/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
bash profiles/collect.sh fp8 r03 1024 base > gpurun_out/collect_r03_base_fp8.log 2>&1; echo "collect base fp8 rc $?"; tail -4 gpurun_out/collect_r03_base_fp8.log | cut -c1-170
bash profiles/collect_mfma.sh bf16 r03 1024 base > gpurun_out/collect_mfma_r03_base_bf16.log 2>&1; echo "mfma rc $?"; tail -14 gpurun_out/collect_mfma_r03_base_bf16.log | cut -c1-170
timeout -k 10 400 python bench.py --preset large-v3 --clips 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_lv3_b32.json 2>/dev/null; echo "lv3 b32 rc $?"; cut -c1-200 $O/bench_lv3_b32.json
timeout -k 10 600 python bench.py --preset large-v3 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > $O/bench_lv3_b256.json 2>/dev/null; echo "lv3 b256 rc $?"; cut -c1-200 $O/bench_lv3_b256.json
