#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O/cli; cd $R
for mb in 256 1024; do
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips 2048 --max-batch $mb --warmup 1 --out-csv $O/cli/p.csv --out-json $O/cli/p.json --out-summary-json $O/cli/summary_b$mb.json > $O/cli/stdout_b$mb.txt 2>&1; echo "cli b$mb rc $?"
python3 -c "
import json
j=json.load(open('$O/cli/summary_b$mb.json')); print(j.get('gpu'))"
done
rm -f $O/cli/p.csv $O/cli/p.json
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "cli" > $O/pytest_cli.log 2>&1; echo "cli tests rc $?"; tail -3 $O/pytest_cli.log
