#!/bin/bash
# round 4 checkpoint: the whole GPU suite, the driver's bench command, the CLI again
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O/cli
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; rc=$?
tail -3 $O/pytest_all.log; [ $rc -eq 0 ] || { tail -60 $O/pytest_all.log; exit $rc; }
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04k/bench_default.json'))
print(round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step'], round(d['roofline']['frac'],3))
for k in ('host_resident','in_tolerance','scaling_strong'):
    print(k, json.dumps(d.get(k))[:400])
PY
for cfg in "4096 2048 1 14" "16384 2048 1 14"; do set -- $cfg
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips $1 --max-batch $2 --streams-per-gpu $3 --load-threads $4 --warmup 1 --out-csv $O/cli/p.csv --out-json $O/cli/p.json --out-summary-json $O/cli/summary_c$1_b$2_s$3_l$4.json > $O/cli/stdout_c$1_b$2_s$3_l$4.txt 2>&1; echo "cli clips $1 max-batch $2 streams $3 loaders $4 rc $?"
python3 -c "
import json;j=json.load(open('$O/cli/summary_c$1_b$2_s$3_l$4.json'));g=j.get('gpu',{});print({k:g[k] for k in g if 'rtf' in k.lower() or 'wall' in k.lower() or 'load' in k.lower()})"
done
rm -f $O/cli/p.csv $O/cli/p.json
