#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "bit_identical or largest_batch_rows or batched_contexts or argmax_edge or teacher_forced_agreement" > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc $?"
python3 -c "import json;j=json.load(open('$O/bench$i.json'));print('  ',round(j['value']),round(j['ms_per_step'],1),j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"; done
