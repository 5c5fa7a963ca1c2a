#!/bin/bash
# rocprofv3 --kernel-trace --stats of a short bench.py run (graph replay kept: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0).
# usage (through gpurun): bash tools/gpu_prof_bench.sh <tag> <bench args...>
set -o pipefail
TAG=${1:-x}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-batch1 "$@" > $OUT/bench_prof.log 2>&1 || { tail -20 $OUT/bench_prof.log; exit 1; }
python3 $R/profiles/summarize_kernel_stats.py "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" ${NBATCH:-4} > $OUT/kernel_stats.txt
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace
tail -1 $OUT/bench_prof.log | cut -c1-600; cat $OUT/kernel_stats.txt
