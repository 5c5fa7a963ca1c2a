#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
b() { name=$1; envs=$2; shift 2; env $envs timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; python3 -c "import json;j=json.load(open('$O/$name.json'));r=j['roofline'];print('  ',round(j['value']),round(j['ms_per_step'],1),r['kernel'],round(r['avg_launch_us'],1),'us frac',round(r['frac'],3))"; }
b u4_cap2 "WH_CROSS_UNROLL=4"
b u2_cap2 "WH_CROSS_UNROLL=2"
b u2_cap0 "WH_CROSS_UNROLL=2 WH_CROSS_WGS_PER_CU=0"
b u4_cap2_nt0 "WH_CROSS_UNROLL=4 WH_CROSS_NT=0"
b u4_cap2_b512 "WH_CROSS_UNROLL=4" --clips 512
b u2_cap2_b512 "WH_CROSS_UNROLL=2" --clips 512
b u4_b256 "WH_CROSS_UNROLL=4" --clips 256
b u2_b256 "WH_CROSS_UNROLL=2" --clips 256
