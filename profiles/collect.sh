#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers on the GPU box (run through gpurun):
#   bash profiles/collect.sh <precision: bf16|fp8> <tag, e.g. r01> [clips per batch, default 256 = bench.py default]
# Writes under gpurun_out/prof_<tag>_<precision>/ and the condensed summaries next to it; copy those to profiles/.
# Kernel trace and PMC counters are separate runs (a --pmc run must not carry trace domains), and the decode
# steps are launched eagerly (WH_NO_GRAPH=1): rocprofv3 on this image crashes when a hipGraph is replayed.
set -eo pipefail
PREC=${1:-bf16}; TAG=${2:-r01}; CLIPS=${3:-256}
OUT=gpurun_out/prof_${TAG}_${PREC}_b${CLIPS}
mkdir -p "$OUT"
export WH_NO_GRAPH=1
export TMPDIR=/tmp
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1 --precision $PREC --clips $CLIPS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
python3 profiles/summarize_kernel_stats.py "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" 4 > "$OUT/kernel_stats.txt"
cp "$(ls $OUT/trace/*/*kernel_stats.csv | head -1)" "$OUT/kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $ARGS > "$OUT/write.log" 2>&1
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
python3 profiles/pmc_summarize.py "$(ls $OUT/fetch/*/*counter_collection.csv | head -1)" "$(ls $OUT/write/*/*counter_collection.csv | head -1)" \
        "$OUT/pmc_hbm_bytes.csv" "$OUT/pmc_traffic.json" "base_${PREC}_b${CLIPS}"
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
cat "$OUT/kernel_stats.txt"; head -8 "$OUT/pmc_hbm_bytes.csv"
