#!/bin/bash
set -eo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mfma_calib; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
$R/tools/mfma_util_calib > "$OUT/plain.txt" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc" -- $R/tools/mfma_util_calib > "$OUT/pmc.log" 2>&1
cp "$(ls $OUT/pmc/*/*counter_collection.csv | head -1)" "$OUT/counters.csv"
rm -rf "$OUT/pmc"
cat "$OUT/plain.txt"; head -3 "$OUT/counters.csv"; python3 - "$OUT/counters.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
for r in rows: print(r.get("Kernel_Name","")[:30], r.get("Counter_Name"), r.get("Counter_Value"), r.get("Start_Timestamp"), r.get("End_Timestamp"))
PY
