#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of `bench.py --steps 3 --warmup 1 --no-cpu-baseline`) into
bytes per launch per kernel family.  rocprofv3 reports both in kilobytes; FETCH_SIZE is doubled (gfx950 tallies 128-B requests
at 64 B, MI355X_MICROARCH.md "HBM").  The output carries the sha256 of the kernel sources it was collected on: bench.py uses
the number only for a build of the same sources.   pmc_traffic.py <fetch.csv|dir> <write.csv|dir> [out.json]"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name):
    if "gemm_nt8_kernel" in name or "gemm_nt_kernel" in name:
        return "gemm_nt"                      # the NT GEMM family (round-1 tiles + deep-pipelined gemm8)
    if "gemm_tn_kernel" in name:
        return "gemm_tn"
    if "adamw_rows" in name:
        return "adamw_rows_kernel"
    if "adamw_kernel" in name:
        return "adamw_kernel"
    return None


def rows_of(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f) as fh:
            yield from csv.DictReader(fh)


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in rows_of(path):
        if r["Counter_Name"] != counter:
            continue
        key = family(r["Kernel_Name"])
        if key:
            tot[key] += float(r["Counter_Value"]) * 1024.0
            cnt[key] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def csrc_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "socialmedia-textimage-classification-auxlosses_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(csrc, fn), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
       "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)", "kernel": "gemm_nt (gemm_nt_kernel + gemm_nt8_kernel)",
       "csrc_sha256_16": csrc_hash(), "dtype": "bf16"}
for k in f:
    out[k] = {"launches_sampled": f[k][1], "fetch_bytes_per_launch": round(2 * f[k][0]), "write_bytes_per_launch": round(w.get(k, (0, 0))[0]),
              "hbm_bytes_per_launch": round(2 * f[k][0] + w.get(k, (0, 0))[0])}
out["hbm_bytes_per_launch"] = out["gemm_nt"]["hbm_bytes_per_launch"] if "gemm_nt" in out else None
json.dump(out, open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout, indent=1)
print()
