#!/bin/bash
# round 4: records of the fp8 mode on e4m3 encoder states — full GPU suite, fp8 bench line, kernel trace + HBM counters, accuracy report (64 clips)
set -o pipefail
mkdir -p gpurun_out/r04aj
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04aj/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04aj/pytest.log
[ $rc -eq 0 ] || { tail -60 gpurun_out/r04aj/pytest.log; exit $rc; }
timeout -k 10 500 python bench.py --precision fp8 --no-cpu-baseline --no-batch1 > gpurun_out/r04aj/bench_fp8.json 2> gpurun_out/r04aj/bench_fp8.err || { tail -20 gpurun_out/r04aj/bench_fp8.err; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04aj/bench_fp8.json').read().strip().splitlines()[-1])
print('fp8', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3))
P
bash profiles/collect.sh fp8 r04 2048 base > gpurun_out/r04aj/collect_fp8.log 2>&1 || { tail -30 gpurun_out/r04aj/collect_fp8.log; exit 1; }
tail -34 gpurun_out/r04aj/collect_fp8.log | head -28
timeout -k 10 600 python tools/fp8_accuracy_report.py --clips 64 --forced-clips 4 --out gpurun_out/r04aj/accuracy_64clips.json > /dev/null 2> gpurun_out/r04aj/accuracy.err || { tail -20 gpurun_out/r04aj/accuracy.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/r04aj/accuracy_64clips.json'))
for k,v in d['teacher_forced_vs_f32'].items(): print(k, round(v['max_abs_logit_err'],4), round(v['mean_abs_logit_err'],4), v['top1_agreement'])
for k,v in d['free_running_vs_f32'].items(): print(k, v)
P
