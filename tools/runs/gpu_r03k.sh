#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
WH_MX_TILE_MIN_TILES=1 timeout -k 10 120 ./tools/mx_gemm_check > $O/mx_gemm_check_wide.txt 2>&1; echo "mx_gemm_check (256 x 256 tiles forced) rc $?"; cat $O/mx_gemm_check_wide.txt
timeout -k 10 120 ./tools/mx_gemm_check > $O/mx_gemm_check.txt 2>&1; echo "mx_gemm_check rc $?"; tail -2 $O/mx_gemm_check.txt
for t in 128 256; do
  WH_MX_TILE=$t timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --precision fp8 > $O/fp8_tile$t.json 2> $O/fp8_tile$t.err; echo "fp8 WH_MX_TILE=$t rc $?"
  python3 -c "import json;j=json.load(open('$O/fp8_tile$t.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fp8 or mx" > $O/pytest_fp8.log 2>&1; echo "fp8 tests rc $?"; tail -5 $O/pytest_fp8.log
