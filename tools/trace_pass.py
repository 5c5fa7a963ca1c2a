#!/usr/bin/env python3
"""Print the kernel dispatches of the LAST encoder pass of a rocprofv3 kernel trace, in launch order, with durations."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of the mel kernel starts the last pass
last = max(i for i, n in enumerate(names) if "k_mel_stft" in n)
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tot = 0
for r in rows[last:last + limit]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{d:9.1f} us  grid {r.get('Grid_Size','?'):>10} wg {r.get('Workgroup_Size','?'):>5} lds {r.get('LDS_Block_Size','?'):>6} vgpr {r.get('VGPR_Count','?'):>4}  {n}")
print(f"total {tot/1e3:.2f} ms")
