#!/bin/bash
set -eo pipefail
O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 600 python bench.py --precision fp8 --no-cpu-baseline > $O/bench_n1_fp8.json 2> $O/bench_n1_fp8.err
timeout -k 10 600 python bench.py --preset large-v3 --clips 256 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > $O/bench_lv3_256.json 2> $O/bench_lv3_256.err
timeout -k 10 600 python bench.py --preset large-v3 --clips 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_lv3_32.json 2> $O/bench_lv3_32.err
