#!/bin/bash
# Which knob lets rocprofv3 --kernel-trace survive the replay of the decode-step hipGraph?  One line per case.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/graph_prof; mkdir -p $OUT
ARGS="$R/bench.py --clips 8 --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 --graph-timed"
run() {  # tag, env assignments...
  tag=$1; shift
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python3 $ARGS > $OUT/triage_$tag.log 2>&1 ); rc=$?
  echo "case $tag [$*]: rocprofv3 rc=$rc" | tee -a $OUT/triage_summary.txt
}
run asis WH_TRIAGE=1
run nopktcap DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run nograph WH_NO_GRAPH=1
ls $OUT/t_nopktcap/*/ 2>/dev/null | head -5
rm -rf $OUT/t_*
