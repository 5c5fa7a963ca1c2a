#!/bin/bash
# round 3: (a) accuracy figures of the bf16 path with the LayerNorm fold (printed by the tests), (b) cross-attention with fewer workgroups per CU
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "teacher_forced or logit_bound" 2>&1 | grep -i "logit\|encoder\|passed\|failed" > $O/accuracy_prints.txt; echo "accuracy tests rc $?"; cat $O/accuracy_prints.txt | cut -c1-250
for pad in 0 40000 70000 140000; do
  WH_CROSS_LDS_PAD=$pad timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_pad$pad.json 2> $O/bench_pad$pad.err; echo "bench WH_CROSS_LDS_PAD=$pad rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_pad$pad.json'));print(j['value'],j['ms_per_step'],'xattn us',j['roofline']['avg_launch_us'],'frac',j['roofline']['frac'],j['kernel_group_ms_per_step'])"
done
