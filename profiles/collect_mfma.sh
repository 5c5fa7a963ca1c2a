#!/bin/bash
# rocprof-reported MFMA utilisation of one bench.py configuration (run through gpurun):
#   bash profiles/collect_mfma.sh <precision> <tag> <clips> <preset>
# One counter pass (no trace domains); summary by profiles/mfma_util.py.
set -eo pipefail
PREC=${1:-bf16}; TAG=${2:-r02}; CLIPS=${3:-1024}; PRESET=${4:-base}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mfma_${TAG}_${PRESET}_${PREC}_b${CLIPS}; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp; export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
export WH_COLLECT_STAMP="${WH_COLLECT_STAMP:-$(cat $R/profiles/.stamp 2>/dev/null || echo unstamped)}"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check --graph-timed --precision $PREC --clips $CLIPS --preset $PRESET > "$OUT/pmc.log" 2>&1
cp $R/profiles/mfma_util.json "$OUT/mfma_util.json"
python3 $R/profiles/mfma_util.py "$(ls $OUT/pmc/*/*counter_collection.csv | head -1)" "$OUT/mfma_util.csv" 0 "$OUT/mfma_util.json" "${PRESET}_${PREC}_b${CLIPS}" | tee "$OUT/mfma_util.txt"
rm -rf "$OUT/pmc"   # (the per-dispatch table is tens of MB; the summary is what is kept)
