#!/bin/bash
# round 4: the 3-byte encoder-state cross-attention of the split-fp16 mode (k_dec_cross_attn_es3) alone: host restatement check + launch time, one and two loader waves
set -o pipefail
mkdir -p gpurun_out/r04ak
rc=0
for nl in 2 1; do
echo "--- WH_ES3_LOADERS=$nl"
WH_ES3_LOADERS=$nl timeout -k 10 300 ./tools/es3_check > gpurun_out/r04ak/es3_check_nl$nl.txt 2>&1 || rc=1
cat gpurun_out/r04ak/es3_check_nl$nl.txt | cut -c1-250
done
exit $rc
