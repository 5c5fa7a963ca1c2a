#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -4 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-200 $O/bench_default.json
