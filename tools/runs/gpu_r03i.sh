#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
for pad in 0 6000 18000 38000; do
  WH_SELF_LDS_PAD=$pad timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/self$pad.json 2> $O/self$pad.err; echo "WH_SELF_LDS_PAD=$pad rc $?"
  python3 -c "import json;j=json.load(open('$O/self$pad.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step']['decode_s'],j['kernel_group_ms_per_step'])"
done
