#!/bin/bash
# round 4: the encoder feed-forward block as one kernel (k_enc_mlp): host-reference check + launch time
set -o pipefail
mkdir -p gpurun_out/r04v
timeout -k 10 240 ./tools/mlp_check > gpurun_out/r04v/mlp_check.txt 2>&1; rc=$?
cat gpurun_out/r04v/mlp_check.txt | cut -c1-300
exit $rc
