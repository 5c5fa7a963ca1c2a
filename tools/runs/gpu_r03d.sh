#!/bin/bash
# round 3: rocprofv3 evidence of the committed build — kernel traces + HBM byte counters (base bf16 / fp8 at 1024 clips, whisper-large-v3
# bf16 at 32 clips, now with counters) and a whisper-large-v3 fp8 trace (k_gemm8_mx at K = 1280 / 5120)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash profiles/collect.sh bf16 r03 1024 base > gpurun_out/collect_r03_base_bf16.log 2>&1; echo "base bf16 rc $?"; tail -25 gpurun_out/collect_r03_base_bf16.log | cut -c1-200
cp gpurun_out/prof_r03_base_bf16_b1024/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
bash profiles/collect.sh bf16 r03 32 large-v3 > gpurun_out/collect_r03_lv3_bf16.log 2>&1; echo "large-v3 bf16 rc $?"; tail -22 gpurun_out/collect_r03_lv3_bf16.log | cut -c1-200
cp gpurun_out/prof_r03_large-v3_bf16_b32/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp; export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$R/gpurun_out/prof_r03_large-v3_fp8_b32; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --graph-timed --precision fp8 --clips 32 --preset large-v3 > $O/trace.log 2>&1; echo "large-v3 fp8 trace rc $?"
python3 $R/profiles/summarize_kernel_stats.py "$(ls $O/trace/*/*kernel_stats.csv | head -1)" 4 > $O/kernel_stats.txt; cp "$(ls $O/trace/*/*kernel_stats.csv | head -1)" $O/kernel_stats.csv; rm -rf $O/trace
tail -1 $O/trace.log | cut -c1-250; cat $O/kernel_stats.txt | cut -c1-170
