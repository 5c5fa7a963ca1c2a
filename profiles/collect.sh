#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers on the GPU box (run through gpurun):
#   bash profiles/collect.sh <precision: bf16|fp8> <tag, e.g. r02> [clips per batch, default 256 = bench.py default] [preset, default base]
# Writes under gpurun_out/prof_<tag>_<preset>_<precision>_b<clips>/ ; copy the condensed summaries to profiles/.
# Kernel trace and PMC counters are separate runs (a --pmc run must not carry trace domains).  The decode steps replay their
# hipGraph as in the timed benchmark; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 keeps rocprofv3 alive (see profiles/README.md).
set -eo pipefail
PREC=${1:-bf16}; TAG=${2:-r02}; CLIPS=${3:-256}; PRESET=${4:-base}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_${PRESET}_${PREC}_b${CLIPS}
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# whisper-large-v3: ~270 kernels per decoder position and the host only synchronises every 16 positions; with ~4,300 profiled
# dispatches in flight the --pmc FETCH_SIZE / WRITE_SIZE passes (TCC-derived: per-channel counters of 8 XCDs per dispatch) die with a
# SIGSEGV inside the tool's record handling, while the same passes survive with the host waiting after every position (<= 270 in
# flight) or restricted to one kernel (profiles/README.md, round-3 triage).  The switch changes no kernel, only when the host waits.
[ "$PRESET" = "large-v3" ] && export WH_SYNC_EVERY_POS=1
export WH_COLLECT_STAMP="${WH_COLLECT_STAMP:-$(cat $R/profiles/.stamp 2>/dev/null || echo unstamped)}"
ARGS="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --graph-timed --precision $PREC --clips $CLIPS --preset $PRESET"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
python3 $R/profiles/summarize_kernel_stats.py "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" 4 > "$OUT/kernel_stats.txt"
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
cp $R/profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
python3 $R/profiles/pmc_summarize.py "$(ls $OUT/fetch/*/*counter_collection.csv | head -1)" "$(ls $OUT/write/*/*counter_collection.csv | head -1)" \
        "$OUT/pmc_hbm_bytes.csv" "$OUT/pmc_traffic.json" "${PRESET}_${PREC}_b${CLIPS}"
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
tail -1 "$OUT/trace.log" | cut -c1-300; cat "$OUT/kernel_stats.txt"; head -12 "$OUT/pmc_hbm_bytes.csv"
