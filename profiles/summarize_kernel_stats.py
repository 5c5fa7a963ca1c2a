#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table (per batch)."""
import csv, re, sys
path, batches = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':78s} {'calls/b':>8s} {'ms/batch':>9s} {'avg_us':>8s} {'pct':>6s}")
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1", "", r["Name"])[:78]
    print(f"{n:78s} {float(r['Calls'])/batches:8.0f} {float(r['TotalDurationNs'])/1e6/batches:9.3f} {float(r['AverageNs'])/1e3:8.2f} {100*float(r['TotalDurationNs'])/tot:6.2f}")
print(f"total kernel ms per batch: {tot/1e6/batches:.2f}")
