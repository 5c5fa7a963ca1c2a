#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout -k 10 600 python tools/fp8_accuracy_report.py --clips 16 --forced-clips 4 > $O/fp8_accuracy_mx.json 2> $O/fp8_accuracy.err; echo "fp8 accuracy rc $?"; cut -c1-600 $O/fp8_accuracy_mx.json
mkdir -p $O/cli; timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips 2048 --max-batch 1024 --warmup 1 --out-csv $O/cli/p.csv --out-json $O/cli/p.json --out-summary-json $O/cli/summary_b1024.json > $O/cli/stdout_b1024.txt 2>&1; echo "cli b1024 rc $?"; grep -i "rtf\|throughput" $O/cli/stdout_b1024.txt | head -5
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips 2048 --max-batch 1024 --streams-per-gpu 2 --warmup 1 --out-csv $O/cli/p2.csv --out-json $O/cli/p2.json --out-summary-json $O/cli/summary_b1024_s2.json > $O/cli/stdout_b1024_s2.txt 2>&1; echo "cli b1024 x2 streams rc $?"
python3 -c "
import json
for n in ('summary_b1024','summary_b1024_s2'):
    j=json.load(open('$O/cli/'+n+'.json')); print(n, j.get('gpu'))"
rm -f $O/cli/p.csv $O/cli/p.json $O/cli/p2.csv $O/cli/p2.json
