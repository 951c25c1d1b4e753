#!/usr/bin/env python3
"""Where a workgroup of the deep-pipelined GEMM (csrc/gemm8.hip) spends its time: in-kernel wall-clock stamps (100 MHz) at
kernel entry, end of the prologue, and per tile after the K loop / after the epilogue (GEMM_DEBUG_TS hook).
usage: gemm8_ts.py M N K tile_code [resid] [cycles]      (tile_code: 13/15 = 256x256 one-shot / persistent, 17/18 = 256x192, 14/16 = 256x128)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
M, N, K, tile = (int(x) for x in sys.argv[1:5])
resid = "resid" in sys.argv[5:]
cyc_flag = 4 if "cycles" in sys.argv[5:] else 0
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
R = torch.randn(M, N, device=dev).to(torch.bfloat16) if resid else None
bias = torch.randn(N, device=dev)
ts = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
flush = torch.empty(128 * 1024 * 1024, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    ts.zero_()
    flush.fill_(1.0)
    lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, p(ts), 0, None, 0, 0.0, 0, 0, p(R), N, 0, (tile << 4) | 2 | cyc_flag, st)
    torch.cuda.synchronize()
raw = ts.cpu().numpy().reshape(-1, 64)
used = raw[:, 0] > 0
cyc = raw[used][:, 48:57].astype(np.float64)
t = raw[used][:, :48].astype(np.float64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0            # 100 MHz -> microseconds
us[t == 0] = np.nan
n = int(np.isfinite(us[0]).sum())
print(f"M={M} N={N} K={K} tile={tile} resid={resid}: {t.shape[0]} workgroups, {n} stamps each (entry, prologue, [k-loop, epilogue] per tile, drain)")
names = ["entry", "prologue done"]
for i in range((n - 3) // 2):
    names += [f"tile{i} k-loop done", f"tile{i} epilogue issued"]
names += ["stores drained"]
for i, nm in enumerate(names[:n]):
    col = us[:, i]
    print(f"  {nm:24s} median {np.nanmedian(col):8.2f} us   min {np.nanmin(col):8.2f}   max {np.nanmax(col):8.2f}")

if cyc[:, 8].max() > 0:
    ph = cyc[:, 8].mean()
    names = ["DMA wait (vmcnt)", "barrier after L", "MFMA interval", "barrier after C"]
    for wv in (0, 1):
        tot = cyc[:, 4 * wv:4 * wv + 4].mean(axis=0)
        print(f"  wave {4 * wv}: shader-clock cycles per phase ({int(ph)} phases; each clock read costs ~40-70 and waits for the wave's LDS reads): "
              + ", ".join(f"{n} {c / ph:.0f}" for n, c in zip(names, tot)) + f"  | sum {tot.sum() / ph:.0f} (16 MFMA = 256)")
