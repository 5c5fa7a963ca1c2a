#!/bin/bash
# round 4: the CLI over 16,384 host-resident synthetic clips in the three fast modes on the final build (wall-clock, load included)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04av; mkdir -p $O
for p in bf16 f16x3 fp8; do
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --precision $p --synthetic-clips 16384 --max-batch 2048 --streams-per-gpu 1 --load-threads 14 --warmup 1 --out-csv $O/p.csv --out-json $O/p.json --out-summary-json $O/summary_${p}_c16384_b2048.json > $O/stdout_$p.txt 2>&1; echo "cli $p rc $?"
python3 -c "
import json;j=json.load(open('$O/summary_${p}_c16384_b2048.json'));g=j.get('gpu',{});print('$p',{k:g[k] for k in g if 'rtf' in k.lower() or 'wall' in k.lower()})"
done
rm -f $O/p.csv $O/p.json
