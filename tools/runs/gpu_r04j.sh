#!/bin/bash
# round 4: the CLI over host-resident clips with the next batch's copy overlapped (wh_transcribe_batch_next), the accuracy record, CLI tests
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O/cli
timeout -k 10 600 python -m pytest tests/test_cli_gpu.py tests/test_hip_parity.py -m gpu -x -q -k "cli or pipelined or device_entry" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log; [ $rc -eq 0 ] || { tail -40 $O/pytest.log; exit $rc; }
for cfg in "4096 2048 1 0" "16384 2048 1 0" "16384 2048 1 14" "8192 1024 1 14"; do set -- $cfg
lt=""; [ "$4" != "0" ] && lt="--load-threads $4"
timeout -k 10 600 ./whisper-rust-ort_amd/whisper_bench --onnx-dir synthetic:base:1234 --synthetic-clips $1 --max-batch $2 --streams-per-gpu $3 $lt --warmup 1 --out-csv $O/cli/p.csv --out-json $O/cli/p.json --out-summary-json $O/cli/summary_c$1_b$2_s$3_l$4.json > $O/cli/stdout_c$1_b$2_s$3_l$4.txt 2>&1; echo "cli clips $1 max-batch $2 streams $3 loaders $4 rc $?"
python3 -c "
import json;j=json.load(open('$O/cli/summary_c$1_b$2_s$3_l$4.json'));g=j.get('gpu',{});print({k:g[k] for k in g if 'rtf' in k.lower() or 'wall' in k.lower() or 'load' in k.lower()})"
done
rm -f $O/cli/p.csv $O/cli/p.json
timeout -k 10 900 python tools/fp8_accuracy_report.py --clips 64 --forced-clips 4 --out $O/accuracy_64clips.json > /dev/null 2> $O/accuracy.err || { tail -5 $O/accuracy.err; exit 1; }
python3 - <<PY
import json
j=json.load(open('$O/accuracy_64clips.json'))
for k,v in j['free_running_vs_f32'].items(): print('free', k, v)
for k,v in j['teacher_forced_vs_f32'].items(): print('forced', k, {a:(round(b,5) if isinstance(b,float) else b) for a,b in v.items()})
PY
