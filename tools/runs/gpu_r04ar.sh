#!/bin/bash
# round 4: MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) of the f16x3 and fp8 steps at 2048 clips on the final build
set -o pipefail
mkdir -p gpurun_out/r04ar
for p in f16x3 fp8; do
bash profiles/collect_mfma.sh $p r04 2048 base > gpurun_out/r04ar/mfma_$p.log 2>&1 || { tail -30 gpurun_out/r04ar/mfma_$p.log; exit 1; }
tail -22 gpurun_out/r04ar/mfma_$p.log | head -12
done
