#!/bin/bash
# big-batch round: parity at 1024 clips, bench lines, large-v3 and fp8 at larger batches
set -eo pipefail
O=gpurun_out/r02h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "batched_contexts" > $O/t1024.log 2>&1
timeout -k 10 300 python tools/big_batch_check.py 1024 bf16 > $O/big1024_bf16.log 2>&1
timeout -k 10 300 python tools/big_batch_check.py 1024 fp8 > $O/big1024_fp8.log 2>&1
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 600 python bench.py --precision fp8 --no-cpu-baseline > $O/bench_n1_fp8.json 2> $O/bench_n1_fp8.err
timeout -k 10 600 python bench.py --preset large-v3 --clips 128 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > $O/bench_lv3_128.json 2> $O/bench_lv3_128.err
timeout -k 10 600 python bench.py --preset large-v3 --clips 256 --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 > $O/bench_lv3_256.json 2> $O/bench_lv3_256.err
