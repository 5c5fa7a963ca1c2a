#!/bin/bash
# round 3: encoder LayerNorm fold (bf16) — whole GPU suite, then A/B timing against WH_NO_ENC_FOLD=1
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -8 $O/pytest_gpu.log
grep -h "max logit err\|max |logit\|encoder max abs" $O/pytest_gpu.log | head -20
for nf in 1 0; do
  if [ $nf = 1 ]; then export WH_NO_ENC_FOLD=1; else unset WH_NO_ENC_FOLD; fi
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_nofold$nf.json 2> $O/bench_nofold$nf.err; echo "bench WH_NO_ENC_FOLD=$nf rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_nofold$nf.json'));print(j['value'],j['ms_per_step'],j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
done
