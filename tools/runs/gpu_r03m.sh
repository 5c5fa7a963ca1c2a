#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -6 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc $?"
python3 -c "import json;j=json.load(open('$O/bench_bf16.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
