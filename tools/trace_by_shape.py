#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): launches, mean / median duration.
usage: trace_by_shape.py <dir-or-csv> [substring-filter]"""
import csv
import glob
import os
import statistics
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        print("no rows", files)
        return
    keys = rows[0].keys()
    name_k = "Kernel_Name" if "Kernel_Name" in keys else [k for k in keys if "ame" in k][0]
    gx = [k for k in keys if k.lower() in ("grid_size_x", "grid_size")]
    wx = [k for k in keys if k.lower() in ("workgroup_size_x", "workgroup_size")]
    agg = {}
    for r in rows:
        n = r[name_k]
        if filt and filt not in n:
            continue
        key = (n[:70], r[gx[0]] if gx else "", r[wx[0]] if wx else "")
        agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':70s} {'grid':>9s} {'wg':>5s} {'n':>6s} {'mean_us':>9s} {'med_us':>9s} {'total_ms':>9s} {'%':>5s}")
    for (n, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:70s} {g:>9s} {w:>5s} {len(v):6d} {statistics.mean(v):9.1f} {statistics.median(v):9.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot:5.1f}")
    print("total ms", tot / 1e3)


if __name__ == "__main__":
    main()
