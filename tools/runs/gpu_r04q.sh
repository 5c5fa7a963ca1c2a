#!/bin/bash
# round 4, records 1/2: rocprofv3 kernel trace + HBM counters + MFMA utilisation of the bf16 and f16x3 steps at 2048 clips
set -o pipefail
for p in bf16 f16x3; do
bash profiles/collect.sh $p r04 2048 base > gpurun_out/r04q_collect_$p.log 2>&1 || { tail -30 gpurun_out/r04q_collect_$p.log; exit 1; }
tail -32 gpurun_out/r04q_collect_$p.log | head -24
bash profiles/collect_mfma.sh $p r04 2048 base > gpurun_out/r04q_mfma_$p.log 2>&1 || { tail -30 gpurun_out/r04q_mfma_$p.log; exit 1; }
tail -14 gpurun_out/r04q_mfma_$p.log
done
