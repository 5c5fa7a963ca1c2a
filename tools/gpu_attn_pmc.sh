#!/bin/bash
# SQ counters of the encoder attention kernel (tools/enc_bench.py, 256 clips)
set -eo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/attn_pmc; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- python3 $R/tools/enc_bench.py --clips 256 --reps 1 > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_LEVEL_LDS --output-format csv -d "$OUT/p2" -- python3 $R/tools/enc_bench.py --clips 256 --reps 1 > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("p1","p2"):
    f=glob.glob(f"{out}/{p}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if "enc_attn" not in r["Kernel_Name"] and "gemm8" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
with open(f"{out}/summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        wc=v["SQ_WAVE_CYCLES"] or 1
        line=(f"{k:60s} wave_cyc {wc:.3e} | wait_any {v['SQ_WAIT_ANY']/wc:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.2f} (lds {v['SQ_WAIT_INST_LDS']/wc:.2f}) active {v['SQ_ACTIVE_INST_ANY']/wc:.2f} "
              f"[valu {v['SQ_ACTIVE_INST_VALU']/wc:.2f} lds {v['SQ_ACTIVE_INST_LDS']/wc:.2f} vmem {v['SQ_ACTIVE_INST_VMEM']/wc:.2f}] mfma_busy/wave_cyc {v['SQ_VALU_MFMA_BUSY_CYCLES']/wc:.2f} "
              f"insts valu {v['SQ_INSTS_VALU']:.3e} mfma {v['SQ_INSTS_MFMA']:.3e} lds {v['SQ_INSTS_LDS']:.3e} lds_conflict/lds_active {v['SQ_LDS_BANK_CONFLICT']/(v['SQ_LDS_IDX_ACTIVE'] or 1):.3f} lds_level/wave_cyc {v['SQ_INST_LEVEL_LDS']/wc:.2f}")
        fo.write(line+"\n"); print(line)
PY
rm -rf "$OUT/p1" "$OUT/p2"
