#!/bin/bash
# round 3: occupancy cap of the cross-attention stream (WH_CROSS_WGS_PER_CU) across precisions / presets / key unroll
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
b() { name=$1; envs=$2; shift 2; env $envs timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; python3 -c "import json;j=json.load(open('$O/$name.json'));r=j['roofline'];print('  ',round(j['value']),round(j['ms_per_step'],1),r['kernel'],round(r['avg_launch_us'],1),'us frac',round(r['frac'],3))"; }
b base_bf16_cap0 WH_CROSS_WGS_PER_CU=0
b base_bf16_cap2 WH_CROSS_WGS_PER_CU=2
b base_bf16_cap2_u8 "WH_CROSS_WGS_PER_CU=2 WH_CROSS_UNROLL=8"
b base_bf16_cap1_u8 "WH_CROSS_WGS_PER_CU=1 WH_CROSS_UNROLL=8"
b base_fp8_cap0 WH_CROSS_WGS_PER_CU=0 --precision fp8
b base_fp8_cap2 WH_CROSS_WGS_PER_CU=2 --precision fp8
b lv3_bf16_b256_cap0 WH_CROSS_WGS_PER_CU=0 --preset large-v3 --clips 256 --steps 2
b lv3_bf16_b256_cap2 WH_CROSS_WGS_PER_CU=2 --preset large-v3 --clips 256 --steps 2
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "largest_batch_rows or batched_contexts or fp8_base_256 or large_v3_bf16_teacher" > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
