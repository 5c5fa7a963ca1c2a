#!/bin/bash
# round 3, second GPU call: whole GPU suite, default bench line (with the CPU leg), CU probe (bandwidth / isolation part),
# and the triage of the rocprofv3 --pmc FETCH_SIZE crash at whisper-large-v3 size
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc $?"; tail -5 $O/pytest_gpu.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-300 $O/bench_default.json; tail -3 $O/bench_default.err
timeout -k 10 200 ./tools/cu_mask_probe > $O/cu_mask_probe.txt 2>&1; echo "probe rc $?"; grep -A 30 "non-temporal" $O/cu_mask_probe.txt
cd /tmp; export TMPDIR=/tmp; export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --graph-timed --precision bf16 --clips 32 --preset large-v3"
# (a) the crashing pass restricted to the dominant kernel: 32 x 16 profiled dispatches between two host syncs instead of ~4,300
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_dec_cross_attn_cg" --output-format csv -d $O/lv3_fetch_regex -- python3 $ARGS > $O/lv3_fetch_regex.log 2>&1; echo "lv3 FETCH regex rc $?"; tail -3 $O/lv3_fetch_regex.log | cut -c1-200
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_dec_cross_attn_cg" --output-format csv -d $O/lv3_write_regex -- python3 $ARGS > $O/lv3_write_regex.log 2>&1; echo "lv3 WRITE regex rc $?"
# (b) every kernel profiled, but the host waits after every decoder position (~270 dispatches in flight at most)
WH_SYNC_EVERY_POS=1 timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/lv3_fetch_sync -- python3 $ARGS > $O/lv3_fetch_sync.log 2>&1; echo "lv3 FETCH sync-every-position rc $?"; tail -3 $O/lv3_fetch_sync.log | cut -c1-200
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
F=$(ls $O/lv3_fetch_regex/*/*counter_collection.csv 2>/dev/null | head -1); W=$(ls $O/lv3_write_regex/*/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then WH_COLLECT_STAMP="$(cat $R/profiles/.stamp)" python3 $R/profiles/pmc_summarize.py "$F" "$W" $O/lv3_pmc_hbm_bytes.csv $O/pmc_traffic.json large-v3_bf16_b32; cat $O/lv3_pmc_hbm_bytes.csv; fi
F2=$(ls $O/lv3_fetch_sync/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$F2" ] && wc -l "$F2"
rm -rf $O/lv3_fetch_regex $O/lv3_write_regex $O/lv3_fetch_sync
