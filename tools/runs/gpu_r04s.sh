#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04s
timeout -k 10 600 python tools/x3_fork_probe.py 512 > gpurun_out/r04s/x3_fork_probe.txt 2>&1 || { tail -5 gpurun_out/r04s/x3_fork_probe.txt; exit 1; }
cat gpurun_out/r04s/x3_fork_probe.txt
timeout -k 10 300 python -m pytest tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -3
