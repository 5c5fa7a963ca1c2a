#!/bin/bash
# round 4: the e4m3 encoder-state cross-attention (k_dec_cross_attn_es8) alone: host restatement check + launch time, two loader waves / one
set -o pipefail
mkdir -p gpurun_out/r04ah
rc=0
for nl in 2 1; do
echo "--- WH_ES8_LOADERS=$nl"
WH_ES8_LOADERS=$nl timeout -k 10 240 ./tools/es8_check > gpurun_out/r04ah/es8_check_nl$nl.txt 2>&1 || rc=1
cat gpurun_out/r04ah/es8_check_nl$nl.txt | cut -c1-250
done
exit $rc
