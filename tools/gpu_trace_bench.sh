#!/bin/bash
# kernel trace of one bench.py configuration (graph replay kept alive under the profiler): bash tools/gpu_trace_bench.sh <tag> <bench args...>
set -eo pipefail
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --graph-timed "$@" > "$OUT/trace.log" 2>&1
python3 $R/profiles/summarize_kernel_stats.py "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" 4 > "$OUT/kernel_stats.txt"
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/trace"
cat "$OUT/kernel_stats.txt"
