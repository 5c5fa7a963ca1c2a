#!/bin/bash
# round 4: the N > 1 path on the one-GPU box — two ranks launched the way the driver launches them (torch.distributed.run, RCCL), both on device 0
set -o pipefail
mkdir -p gpurun_out/r04ad
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --clips 512 --test-single-device --no-cpu-baseline --no-batch1 > gpurun_out/r04ad/bench_n2.json 2> gpurun_out/r04ad/bench_n2.err; rc=$?
tail -3 gpurun_out/r04ad/bench_n2.err | cut -c1-300
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04ad/bench_n2.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','n_gpus','ms_per_step','scaling')}, d['config'])
P
exit $rc
