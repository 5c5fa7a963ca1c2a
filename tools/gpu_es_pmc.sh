#!/bin/bash
# where do k_dec_cross_attn_es*'s wave cycles go?  SQ counter passes over tools/es_bench (run through gpurun)
set -eo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/es_pmc; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- $R/tools/es_bench 1024 2 > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_LEVEL_LDS --output-format csv -d "$OUT/p2" -- $R/tools/es_bench 1024 2 > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for p in ("p1","p2"):
    f=glob.glob(f"{out}/{p}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_ACTIVE_INST_VMEM"): n[(k,p)]+=1
with open(f"{out}/summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        wc=v["SQ_WAVE_CYCLES"] or 1
        nl=n[(k,"p2")] or 1
        line=(f"{k:60s} launches {n[(k,'p1')]} wave_cyc {wc:.3e} | wait_any {v['SQ_WAIT_ANY']/wc:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.2f} (lds {v['SQ_WAIT_INST_LDS']/wc:.2f}) active {v['SQ_ACTIVE_INST_ANY']/wc:.2f} "
              f"[valu {v['SQ_ACTIVE_INST_VALU']/wc:.2f} lds {v['SQ_ACTIVE_INST_LDS']/wc:.2f}] | per launch: mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/nl:.3e} lds_idx_active {v['SQ_LDS_IDX_ACTIVE']/nl:.3e} lds_bank_conflict {v['SQ_LDS_BANK_CONFLICT']/nl:.3e} "
              f"(conflict/active {v['SQ_LDS_BANK_CONFLICT']/(v['SQ_LDS_IDX_ACTIVE'] or 1):.3f}) insts valu {v['SQ_INSTS_VALU']/nl:.3e} lds {v['SQ_INSTS_LDS']/nl:.3e} mfma {v['SQ_INSTS_MFMA']/nl:.3e} lds_level {v['SQ_INST_LEVEL_LDS']/nl:.3e}")
        fo.write(line+"\n"); print(line)
PY
rm -rf "$OUT/p1" "$OUT/p2"
