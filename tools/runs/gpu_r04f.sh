#!/bin/bash
# round 4: rocprofv3 kernel trace + HBM counters of the split-fp16 mode at the benchmark shape (2048 clips per step)
set -o pipefail
bash profiles/collect.sh f16x3 r04 2048 base > gpurun_out/r04f_collect.log 2>&1 || { tail -30 gpurun_out/r04f_collect.log; exit 1; }
tail -60 gpurun_out/r04f_collect.log
