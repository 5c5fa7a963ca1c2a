#!/bin/bash
# round 3: LM head on 256 x 256 tiles — parity (bit-identity with k_lm_head) and A/B timing; MFMA-utilisation counters of the build
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "bit_identical or largest_batch_rows or teacher_forced_agreement or batched_contexts or argmax_edge or device_entry" > $O/pytest_lm.log 2>&1; echo "lm tile tests rc $?"; tail -6 $O/pytest_lm.log
for mr in 0 256; do
  WH_LM_TILE_MIN_ROWS=$mr timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_lm$mr.json 2> $O/bench_lm$mr.err; echo "bench WH_LM_TILE_MIN_ROWS=$mr rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_lm$mr.json'));print(j['value'],j['ms_per_step'],j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
done
WH_LM_TILE_MIN_ROWS=256 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --clips 256 > $O/bench_lm256_b256.json 2>/dev/null; WH_LM_TILE_MIN_ROWS=0 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --clips 256 > $O/bench_lm0_b256.json 2>/dev/null
python3 -c "import json;a=json.load(open('$O/bench_lm256_b256.json'));b=json.load(open('$O/bench_lm0_b256.json'));print('256 clips: tile',a['ms_per_step'],'k_lm_head',b['ms_per_step'])"
