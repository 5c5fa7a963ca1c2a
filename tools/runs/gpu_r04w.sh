#!/bin/bash
# round 4: k_enc_mlp in the pipeline, A/B against the two k_gemm8 launches (WH_ENC_MLP=0), then the bf16 parity tests
set -o pipefail
mkdir -p gpurun_out/r04w
for v in 0 1; do
WH_ENC_MLP=$v timeout -k 10 500 python bench.py --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04w/bench_mlp$v.json 2> gpurun_out/r04w/bench_mlp$v.err || { tail -20 gpurun_out/r04w/bench_mlp$v.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04w/bench_mlp$v.json'))
print('mlp=$v', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
P
done
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf16 or batched or logit_bound or fold" > gpurun_out/r04w/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04w/pytest.log
exit $rc
