#!/usr/bin/env python3
"""MFMA utilisation per kernel family from a rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE run.
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs; GRBM_GUI_ACTIVE comes out summed over the 8 XCDs:
utilisation = MFMA_BUSY / (1024 x GUI_ACTIVE / 8).      usage: pmc_mfma.py <dir>"""
import csv, glob, os, re, sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
trace = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    trace += list(csv.DictReader(open(f)))
dur = defaultdict(list)
for r in trace:
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def fam(n):
    m = re.search(r"gemm_nt8k32_kernel", n)
    if m: return "gemm_nt8k32 256x256 BK=32"
    m = re.search(r"gemm_nt8_kernelI\w+?Li(\d+)ELi(\d)", n)
    if m: return f"gemm_nt8 256x{m.group(1)} (epilogue class {m.group(2)})"
    m = re.search(r"gemm_nt_kernelI\w+?Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi\d+ELi(\d+)", n)
    if m: return f"gemm_nt {m.group(1)}x{m.group(2)}" + (" role-specialised" if m.group(3) != "0" else "")
    if "gemm_tn_kernel" in n: return "gemm_tn (grouped dW)"
    for k in ("attn_fwd", "attn_bwd", "fusion_attn_fwd", "fusion_attn_bwd", "small_gemm"):
        if k in n: return k
    return None


busy, act, cnt, us = defaultdict(float), defaultdict(float), defaultdict(int), defaultdict(list)
for r in rows:
    f = fam(r["Kernel_Name"])
    if not f:
        continue
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[f] += float(r["Counter_Value"]); cnt[f] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[f] += float(r["Counter_Value"])
for n, v in dur.items():
    f = fam(n)
    if f:
        us[f] += v
print(f"{'kernel':52s} {'launches':>8s} {'avg us':>8s} {'MFMA utilisation':>17s}")
for f in sorted(busy, key=lambda k: -sum(us[k])):
    print(f"{f:52s} {cnt[f]:8d} {sum(us[f]) / max(len(us[f]), 1):8.1f} {busy[f] / (1024 * act[f] / 8):17.3f}")
