#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04o
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "folded_layernorm" > gpurun_out/r04o/pytest.log 2>&1
grep -E "offset enc|passed|failed|assert" gpurun_out/r04o/pytest.log | head -20
