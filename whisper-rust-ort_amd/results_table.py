#!/usr/bin/env python3
"""Cross-run results table from `inference_summary.json` files.

Consumes the summary JSON the CLI emits (`whisper_bench`, same schema as reference src/main.rs:1235-1257, with the
additive `gpu{}` / `rtfx_end_to_end{}` keys) — and, unchanged, the summaries of the reference's own variants — and
writes the markdown + CSV table of the reference's aggregation step (`compare_container_benchmarks.py:118-226`:
columns Implementation | Precision | Beam size | Time | RAM Usage; CSV fields implementation, precision, beam_size,
time_s, ram_mb), so a RESULTS.md maintained with `update_results_md.py:33-143` can take rows from either side.

    python results_table.py --results-dir results/benchmarks/mi355x            # every */inference_summary.json below it
    python results_table.py --summary "MI355X bf16=run1/inference_summary.json" --summary "CPU=run2/inference_summary.json"

Time = the summary's end-to-end p95 (falling back through p90, median, mean, max, min as the reference does, :100-115),
else the wall time of `<log-dir>/<name>.time.txt` (`/usr/bin/time -v` output); RAM from the same log.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import re
from typing import List, Optional, Tuple

STAT_ORDER = ("p95", "p90", "median", "mean", "max", "min")


def _num(v) -> Optional[float]:
    if isinstance(v, bool):
        return None
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return None
    return None


def stat_of(summary: dict, key: str) -> Optional[float]:
    block = summary.get(key)
    if not isinstance(block, dict):
        return None
    for k in STAT_ORDER:
        v = _num(block.get(k))
        if v is not None:
            return v
    return None


def time_log(path: str) -> Tuple[Optional[float], Optional[int]]:
    """(elapsed seconds, max RSS in KB) from a `/usr/bin/time -v` log; (None, None) when absent."""
    if not path or not os.path.isfile(path):
        return None, None
    elapsed = rss = None
    for line in open(path, encoding="utf-8", errors="ignore"):
        m = re.search(r"Elapsed \(wall clock\) time.*?:\s*([0-9:.]+)\s*$", line)
        if m:
            parts = [float(x) for x in m.group(1).split(":")]
            elapsed = sum(p * 60.0 ** i for i, p in enumerate(reversed(parts)))
        m = re.search(r"Maximum resident set size.*?:\s*(\d+)", line)
        if m:
            rss = int(m.group(1))
    return elapsed, rss


def human_time(s: Optional[float]) -> str:
    if s is None:
        return "n/a"
    t = int(round(s))
    h, t = divmod(t, 3600)
    m, sec = divmod(t, 60)
    return f"{h}h{m:02d}m{sec:02d}s" if h else (f"{m}m{sec:02d}s" if m else f"{sec}s")


def precision_of(summary: dict, default: str = "fp32") -> str:
    gpu = summary.get("gpu")
    if isinstance(gpu, dict) and isinstance(gpu.get("precision"), str):
        return gpu["precision"]
    ct = (summary.get("config_used") or {}).get("compute_type")
    if isinstance(ct, str):
        low = ct.strip().lower()
        return {"float32": "fp32", "fp32": "fp32", "qint8": "int8", "int8": "int8"}.get(low, ct)
    return default


def beam_of(summary: dict) -> int:
    cfg = summary.get("config_used") or {}
    for k in ("num_beams", "beam_size"):
        v = cfg.get(k)
        if isinstance(v, int):
            return v
        if isinstance(v, str) and v.isdigit():
            return int(v)
    return 1   # the path is greedy (src/main.rs:753-829)


def label_of(summary: dict, name: str) -> str:
    gpu = summary.get("gpu")
    if isinstance(gpu, dict):
        return f"{gpu.get('backend', 'libwhisper_hip')} [{name}]"
    return name


def row_for(label: Optional[str], name: str, path: str, log_dir: Optional[str]) -> dict:
    summary = json.load(open(path, encoding="utf-8")) if os.path.isfile(path) else {}
    elapsed, rss = time_log(os.path.join(log_dir, name + ".time.txt") if log_dir else "")
    t = stat_of(summary, "latency_end_to_end_s")
    if t is None:
        t = elapsed
    rtfx = stat_of(summary, "rtfx_end_to_end")
    if rtfx is None:
        r = stat_of(summary, "rtf_end_to_end")      # reference rtf = latency / duration (src/main.rs:1191)
        rtfx = 1.0 / r if r else None
    return {"implementation": label or label_of(summary, name), "precision": precision_of(summary), "beam_size": beam_of(summary),
            "time_s": None if t is None else round(t, 3), "time": human_time(t),
            "ram_mb": None if rss is None else int(round(rss / 1024.0)), "ram": "n/a" if rss is None else f"{int(round(rss / 1024.0))}MB",
            "n_files": summary.get("n_files"), "rtfx": None if rtfx is None else round(rtfx, 1)}


def discover(results_dir: str) -> List[Tuple[Optional[str], str, str]]:
    out = []
    for name in sorted(os.listdir(results_dir)):
        p = os.path.join(results_dir, name, "inference_summary.json")
        if os.path.isfile(p):
            out.append((None, name, p))
    p = os.path.join(results_dir, "inference_summary.json")
    if os.path.isfile(p):
        out.append((None, os.path.basename(os.path.abspath(results_dir)), p))
    return out


def write_tables(rows: List[dict], out_md: str, out_csv: str, extra: bool) -> None:
    for p in (out_md, out_csv):
        os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
    with open(out_md, "w", encoding="utf-8") as f:
        head = ["Implementation", "Precision", "Beam size", "Time", "RAM Usage"] + (["Files", "x real time"] if extra else [])
        f.write("| " + " | ".join(head) + " |\n| " + " | ".join("---" for _ in head) + " |\n")
        for r in rows:
            cells = [r["implementation"], r["precision"], str(r["beam_size"]), r["time"], r["ram"]]
            if extra:
                cells += ["n/a" if r["n_files"] is None else str(r["n_files"]), "n/a" if r["rtfx"] is None else str(r["rtfx"])]
            f.write("| " + " | ".join(cells) + " |\n")
    with open(out_csv, "w", newline="", encoding="utf-8") as f:
        fields = ["implementation", "precision", "beam_size", "time_s", "ram_mb"] + (["n_files", "rtfx"] if extra else [])
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        for r in rows:
            w.writerow({k: r[k] for k in fields})


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--results-dir", default="")
    ap.add_argument("--summary", action="append", default=[], metavar="LABEL=PATH")
    ap.add_argument("--log-dir", default="")
    ap.add_argument("--out-md", default="results/benchmarks/summary_table.md")
    ap.add_argument("--out-csv", default="results/benchmarks/summary_table.csv")
    ap.add_argument("--extra-columns", action="store_true", help="append Files and x-real-time columns (not in the reference's table)")
    a = ap.parse_args(argv)
    jobs: List[Tuple[Optional[str], str, str]] = []
    if a.results_dir:
        jobs += discover(a.results_dir)
    for s in a.summary:
        label, _, path = s.rpartition("=")
        name = os.path.basename(os.path.dirname(os.path.abspath(path))) or "run"
        jobs.append((label or None, name, path))
    if not jobs:
        ap.error("nothing to tabulate: give --results-dir or --summary")
    rows = [row_for(lbl, name, path, a.log_dir or None) for lbl, name, path in jobs]
    write_tables(rows, a.out_md, a.out_csv, a.extra_columns)
    print("Wrote summary table:", a.out_md)
    print("Wrote summary csv:", a.out_csv)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
