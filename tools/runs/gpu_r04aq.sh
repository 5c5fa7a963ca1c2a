#!/bin/bash
# round 4: the writers of the e4m3 / fp16 + e4m3 encoder states clamp to e4m3's range: the tests that run them
set -o pipefail
mkdir -p gpurun_out/r04aq
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "encoder_state or (f16x3 and (256 or 2048)) or fp8_base_256" > gpurun_out/r04aq/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r04aq/pytest.log
exit $rc
