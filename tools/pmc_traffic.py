#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`) into
bytes per launch per kernel.  FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B... rocprofv3 reports them in kilobytes;
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM").   pmc_traffic.py <fetch.csv> <write.csv> [out.json]"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = "gemm_nt_kernel" if "gemm_nt_kernel" in name else "gemm_tn_kernel" if "gemm_tn_kernel" in name else "adamw_rows_kernel" if "adamw_rows" in name else \
            "adamw_kernel" if "adamw_kernel" in name else None
        if key:
            tot[key] += float(r["Counter_Value"]) * 1024.0
            cnt[key] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
       "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)", "kernel": "gemm_nt_kernel"}
for k in f:
    out[k] = {"launches_sampled": f[k][1], "fetch_bytes_per_launch": round(2 * f[k][0]), "write_bytes_per_launch": round(w.get(k, (0, 0))[0]),
              "hbm_bytes_per_launch": round(2 * f[k][0] + w.get(k, (0, 0))[0])}
out["hbm_bytes_per_launch"] = out["gemm_nt_kernel"]["hbm_bytes_per_launch"]
json.dump(out, open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout, indent=1)
print()
