#!/bin/bash
# round 4: k_dec_self_attn at four waves per SIMD (128 registers; one prefetched key per lane with f32 rows): parity, then the bench groups
set -o pipefail
mkdir -p gpurun_out/r04af
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "f32 or f16x3 or batched or deterministic or full_batch" > gpurun_out/r04af/pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r04af/pytest.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04af/pytest.log; exit $rc; }
for p in bf16 f16x3; do
timeout -k 10 500 python bench.py --precision $p --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04af/bench_$p.json 2> gpurun_out/r04af/bench_$p.err || { tail -20 gpurun_out/r04af/bench_$p.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04af/bench_$p.json'))
print('$p', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3), d['workspace_placement']['workspaces_timed'])
P
done
timeout -k 10 300 python bench.py --clips 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04af/bench_b64.json 2> gpurun_out/r04af/bench_b64.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04af/bench_b64.json').read().strip().splitlines()[-1])
print('64 clips', round(d['value']), round(d['ms_per_step'],2), 'batch1', d['batch1'])
P
