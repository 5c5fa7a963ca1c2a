#!/bin/bash
# round 3: verification of the end-of-round build — whole GPU suite, smoke, default bench line, kernel trace + counters (bf16)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-200
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-200 $O/bench_default.json
bash profiles/collect.sh bf16 r03 1024 base > gpurun_out/collect_r03_base_bf16.log 2>&1; echo "collect base bf16 rc $?"; tail -30 gpurun_out/collect_r03_base_bf16.log | head -18 | cut -c1-170
bash profiles/collect_mfma.sh bf16 r03 1024 base > gpurun_out/collect_mfma_r03_base_bf16.log 2>&1; echo "mfma rc $?"
