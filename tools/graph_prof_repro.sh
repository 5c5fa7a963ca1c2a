#!/bin/bash
# Runs tools/graph_prof_repro under rocprofv3 --kernel-trace in several shapes; one line per case with the exit code.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/graph_prof; mkdir -p $OUT
for c in "4 3 0 0" "49 3 0 0" "49 16 320 0" "49 16 320 1" "200 4 2048 0"; do
  tag=$(echo $c | tr ' ' '_')
  $R/tools/graph_prof_repro $c > $OUT/plain_$tag.log 2>&1; a=$?
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- $R/tools/graph_prof_repro $c > $OUT/prof_$tag.log 2>&1; b=$?
  echo "case [$c]: plain rc=$a, under rocprofv3 rc=$b" | tee -a $OUT/summary.txt
done
rm -rf $OUT/t_*
