#!/bin/bash
# round 3, first GPU call: CU-mask mapping / partition probe, the GPU test suite, plain and pipelined bench lines
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 240 ./tools/cu_mask_probe > $O/cu_mask_probe.txt 2>&1; echo "probe rc $?"; tail -30 $O/cu_mask_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "pipelined or device_entry or large_v3 or cli_end_to_end" > $O/pytest_new.log 2>&1; echo "new tests rc $?"; tail -15 $O/pytest_new.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 > $O/bench_plain.json 2> $O/bench_plain.err; echo "plain rc $?"; cut -c1-400 $O/bench_plain.json
for e in 32 64 96; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch1 --pipeline 1 --enc-cus $e > $O/bench_pipe_e$e.json 2> $O/bench_pipe_e$e.err; echo "pipe $e rc $?"; cut -c1-300 $O/bench_pipe_e$e.json
done
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch1 --pipeline 1 --enc-cus 0 > $O/bench_pipe_e0.json 2> $O/bench_pipe_e0.err; echo "pipe nomask rc $?"; cut -c1-300 $O/bench_pipe_e0.json
