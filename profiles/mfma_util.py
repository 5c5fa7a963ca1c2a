#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.

    python3 mfma_util.py <counter_collection.csv> <out.csv> [skip-dispatches-per-kernel [<mfma_util.json> <config-key>]]

The clock column (and with it the utilisation) is only meaningful for launches of >= 1 ms: GRBM_GUI_ACTIVE of a
microsecond-scale launch includes the dispatch ramp on both sides of the kernel's own timestamps, so the derived clock
reads 4-13 "GHz" and the utilisation is understated.  Such rows are printed with `clock n/a` and carry
"short_launch": true in the JSON.

Calibration on gfx950 (tools/mfma_util_calib.hip, profiles/r02_mfma_util_calibration.txt): SQ_VALU_MFMA_BUSY_CYCLES comes summed
over every SIMD of the chip and counts 16 cycles per v_mfma_f32_16x16x32_bf16 (the instruction's issue time); GRBM_GUI_ACTIVE comes
summed over the 8 XCDs.  So with 256 CUs x 4 SIMDs

    util = sum(BUSY) / (sum(GUI_ACTIVE) / 8 * 1024)        # fraction of SIMD-cycles with the matrix pipe busy

and sum(GUI_ACTIVE) / 8 / sum(duration) is the shader clock the kernel actually ran at.  The calibration loops — nothing but
independent MFMAs, four to eight waves per SIMD — read 0.91-0.98 at a clock of 1.9-2.0 GHz, i.e. 1.9-2.0 PFLOP/s: under
matrix load the part does not hold the 2.4 GHz behind the nominal 2.5 PFLOP/s.  `util` is against the cycles that happened,
the bench line's flop/s fractions against the nominal peak.
"""
import csv
import collections
import re
import sys


def short_name(raw):
    """`k_gemm8<float, 256, 0>` from either the demangled or the mangled spelling rocprofv3 reports."""
    name = raw
    if name.startswith("_Z"):       # no demangler in the image knows DF16b: the few argument kinds these kernels use, by hand
        m = re.search(r"\d+(k_[A-Za-z0-9_]+?)I(.*?)EEv", name)
        if m:
            args, rest, i = [], m.group(2) + "E", 0
            while i < len(rest):
                for pat, fmt in ((r"DF16b", "bf16"), (r"Li(\d+)E", "{0}"), (r"Lb([01])E", "{0}"), (r"NS_\d+([A-Za-z0-9_]+?)E", "{0}"), (r"f", "float"), (r"h", "u8")):
                    mm = re.match(pat, rest[i:])
                    if mm:
                        args.append(fmt.format(*mm.groups()))
                        i += mm.end()
                        break
                else:
                    i += 1
            return m.group(1) + "<" + ", ".join(args) + ">"
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.search(r"(k_[A-Za-z0-9_]+|__amd_[A-Za-z0-9_]+)", name)
    if not m:
        return name[:60]
    out, rest = m.group(1), name[m.end():]
    if rest.startswith("<"):          # balanced template argument list
        depth = 0
        for i, ch in enumerate(rest):
            depth += ch == "<"
            depth -= ch == ">"
            if depth == 0:
                out += rest[: i + 1]
                break
    return out.replace("__hip_bfloat16", "bf16").replace("__bf16", "bf16")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    per = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch -> counter -> value
    dur = collections.defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(src)):
        raw = r["Kernel_Name"]
        if raw not in names:
            names[raw] = short_name(raw)
        k = names[raw]
        d = int(r["Dispatch_Id"])
        per[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
        dur[k][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    rows = []
    for k, disp in per.items():
        ids = sorted(disp)[skip:] or sorted(disp)
        busy = sum(disp[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for i in ids)
        gui = sum(disp[i].get("GRBM_GUI_ACTIVE", 0.0) for i in ids)
        t = sum(dur[k][i] for i in ids)
        if gui <= 0:
            continue
        short = t / max(1, len(ids)) < 1e-3     # average launch below 1 ms: clock (and utilisation) not meaningful
        rows.append((k, len(ids), t * 1e3, busy, gui, busy / (gui / 8.0 * 1024.0), gui / 8.0 / t / 1e9 if t > 0 else 0.0, short))
    rows.sort(key=lambda r: -r[2])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "total_ms_under_pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "mfma_util", "shader_clock_GHz"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.0f}", f"{r[4]:.0f}", f"{r[5]:.4f}", "n/a (launch < 1 ms)" if r[7] else f"{r[6]:.3f}"])
    for r in rows[:20]:
        clock = "clock n/a (launch < 1 ms: utilisation understated)" if r[7] else f"clock {r[6]:.2f} GHz"
        print(f"{r[0][:64]:64s} {r[1]:6d} launches {r[2]:10.2f} ms  MFMA util {r[5]:.3f}  {clock}")
    if len(sys.argv) > 5:
        import json
        import os
        try:
            uj = json.load(open(sys.argv[4]))
        except Exception:
            uj = {}
        uj[sys.argv[5]] = {r[0]: {"launches": r[1], "total_ms_under_pmc": round(r[2], 3), "mfma_util": round(r[5], 4),
                                  "shader_clock_GHz": None if r[7] else round(r[6], 3), "short_launch": r[7]} for r in rows}
        uj.setdefault("_collected_at", {})[sys.argv[5]] = os.environ.get("WH_COLLECT_STAMP", "unstamped")
        json.dump(uj, open(sys.argv[4], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
