#!/bin/bash
# round 3, encoder-state cross-attention: the CLI's throughput path over host-resident clips (run through gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O/cli
for cfg in "4096 2048 1" "4096 1024 1" "4096 2048 2"; do set -- $cfg
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips $1 --max-batch $2 --streams-per-gpu $3 --warmup 1 --out-csv $O/cli/p.csv --out-json $O/cli/p.json --out-summary-json $O/cli/summary_c$1_b$2_s$3.json > $O/cli/stdout_c$1_b$2_s$3.txt 2>&1; echo "cli clips $1 max-batch $2 streams $3 rc $?"
python3 -c "
import json;j=json.load(open('$O/cli/summary_c$1_b$2_s$3.json'));g=j.get('gpu',{});print({k:g[k] for k in g if 'rtf' in k.lower() or 'wall' in k.lower() or 'clips' in k.lower()})"
done
rm -f $O/cli/p.csv $O/cli/p.json
