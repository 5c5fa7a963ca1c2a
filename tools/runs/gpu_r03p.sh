#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for c in 64 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --graph-timed --clips $c > $O/trace$c.log 2>&1; echo "trace $c rc $?"
python3 $R/profiles/summarize_kernel_stats.py "$(ls $O/trace$c/*/*kernel_stats.csv | head -1)" 6 > $O/kernel_stats_b$c.txt; cp "$(ls $O/trace$c/*/*kernel_stats.csv | head -1)" $O/kernel_stats_b$c.csv; rm -rf $O/trace$c
tail -1 $O/trace$c.log | cut -c1-200; cat $O/kernel_stats_b$c.txt | cut -c1-150
done
