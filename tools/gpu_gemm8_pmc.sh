#!/bin/bash
# where do k_gemm8's wave cycles go?  two SQ counter passes over tools/gemm8_ablate (full kernel and ablated variants)
set -eo pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gemm8_pmc; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- $R/tools/gemm8_ablate > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/p2" -- $R/tools/gemm8_ablate > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,sys,glob,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for p in ("p1","p2"):
    f=glob.glob(f"{out}/{p}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if int(r["Grid_Size"]) < 100000: continue     # skip the small correctness launches
        k=r["Kernel_Name"].replace("void (anonymous namespace)::","")+" grid"+r["Grid_Size"]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_ACTIVE_INST_VMEM"): n[(k,r["Counter_Name"])]+=1
with open(f"{out}/summary.txt","w") as fo:
    for k,v in sorted(agg.items()):
        wc=v["SQ_WAVE_CYCLES"] or 1
        line=(f"{k[:70]:70s} wave_cyc {wc:.3e} | wait_any {v['SQ_WAIT_ANY']/wc:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.2f} (lds {v['SQ_WAIT_INST_LDS']/wc:.2f}) active {v['SQ_ACTIVE_INST_ANY']/wc:.2f} "
              f"[valu {v['SQ_ACTIVE_INST_VALU']/wc:.2f} lds {v['SQ_ACTIVE_INST_LDS']/wc:.2f} vmem {v['SQ_ACTIVE_INST_VMEM']/wc:.2f}] mfma_busy/wave_cyc {v['SQ_VALU_MFMA_BUSY_CYCLES']/wc:.2f} "
              f"lds_conflict/lds_active {v['SQ_LDS_BANK_CONFLICT']/(v['SQ_LDS_IDX_ACTIVE'] or 1):.3f} ta_addr_full {v['SQ_VMEM_TA_ADDR_FIFO_FULL']/wc:.3f} ta_cmd_full {v['SQ_VMEM_TA_CMD_FIFO_FULL']/wc:.3f} wr_data_full {v['SQ_VMEM_WR_TA_DATA_FIFO_FULL']/wc:.3f} vmem_level/wave_cyc {v['SQ_INST_LEVEL_VMEM']/wc:.2f}")
        fo.write(line+"\n"); print(line)
PY
rm -rf "$OUT/p1" "$OUT/p2"
