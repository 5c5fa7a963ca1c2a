#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc $?"
python3 -c "import json;j=json.load(open('$O/bench_bf16.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --precision fp8 > $O/bench_fp8.json 2> $O/bench_fp8.err; echo "fp8 rc $?"
python3 -c "import json;j=json.load(open('$O/bench_fp8.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "teacher_forced or golden or fp8 or batch_equals or permutation" > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
