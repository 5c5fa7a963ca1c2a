#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as
MI355X_MICROARCH.md §HBM prescribes) into per-kernel HBM bytes per launch.

  python3 profiles/pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
          <out_summary.csv> [<pmc_traffic.json> <config-key>]

Units / gfx950 corrections applied (MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7):
  * FETCH_SIZE and WRITE_SIZE are reported in KiB → × 1024;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read
    (16 B/lane) → × 2 on the read side (every kernel listed here reads with 16-B lane accesses);
  * WRITE_SIZE is exact for 16-B-per-lane streaming stores; narrower stores are uncalibrated.
"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d*", "", name)
    m = re.search(r"k_[a-z0-9_]+", name)
    return m.group(0) if m else name[:40]


def main():
    f = agg(sys.argv[1], "FETCH_SIZE")
    w = agg(sys.argv[2], "WRITE_SIZE")
    rows = []
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, fv in f.items():
        wv = w.get(k, [0.0])
        s = short(k)
        per[s][0] += len(fv)
        per[s][1] += sum(fv)
        per[s][2] += sum(wv) * (len(fv) / max(1, len(wv)))
    for s, (n, fs, ws) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        fetch_b = fs / n * 1024 * 2
        write_b = ws / n * 1024
        rows.append((s, n, fs / n, ws / n, fetch_b, write_b, fetch_b + write_b))
    with open(sys.argv[3], "w") as o:
        o.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,read_bytes_per_launch(x2 gfx950),write_bytes_per_launch,hbm_bytes_per_launch\n")
        for r in rows:
            o.write(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]:.0f},{r[5]:.0f},{r[6]:.0f}\n")
    if len(sys.argv) > 5:
        try:
            tj = json.load(open(sys.argv[4]))
        except Exception:
            tj = {}
        for r in rows:
            tj.setdefault(r[0], {})[sys.argv[5]] = r[6]
        import os
        tj.setdefault("_collected_at", {})[sys.argv[5]] = os.environ.get("WH_COLLECT_STAMP", "unstamped")   # build the counters describe
        json.dump(tj, open(sys.argv[4], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
