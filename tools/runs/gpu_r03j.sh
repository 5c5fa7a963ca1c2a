#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
for st in 0 8 16 24; do
  WH_GEMM8_STAGGER_US=$st timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/stag$st.json 2> $O/stag$st.err; echo "WH_GEMM8_STAGGER_US=$st rc $?"
  python3 -c "import json;j=json.load(open('$O/stag$st.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),round(j['stage_ms_per_step']['encode_s'],1),j['kernel_group_ms_per_step'])"
done
