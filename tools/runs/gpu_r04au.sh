#!/bin/bash
# round 4: after factoring the shared fp8 helpers into wh_es_fp8.h — the two kernel checks and the tests that run the kernels in the pipeline
set -o pipefail
mkdir -p gpurun_out/r04au
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "encoder_state or split_fp16 or (f16x3 and 2048) or fp8_base_256" > gpurun_out/r04au/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r04au/pytest.log
exit $rc
