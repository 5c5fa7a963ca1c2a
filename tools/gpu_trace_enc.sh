#!/bin/bash
# rocprofv3 kernel trace of one encoder pass (tools/enc_bench.py), per-launch durations in launch order.
# usage (through gpurun): bash tools/gpu_trace_enc.sh <tag> [enc_bench args...]
set -o pipefail
TAG=${1:-x}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0   # rocprofv3 x hipGraph replay: see profiles/README.md
rocprofv3 --kernel-trace --output-format csv -d $OUT/enc_trace -- python3 $R/tools/enc_bench.py --reps 1 "$@" > $OUT/enc_trace.log 2>&1 || { tail -20 $OUT/enc_trace.log; exit 1; }
python3 $R/tools/trace_pass.py "$(ls $OUT/enc_trace/*/*kernel_trace.csv | head -1)" ${LIMIT:-52} > $OUT/enc_pass.txt
rm -rf $OUT/enc_trace
cat $OUT/enc_pass.txt
