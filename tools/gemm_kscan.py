#!/usr/bin/env python3
"""NT GEMM time vs K for each tile variant at fixed M x N: the slope is the steady-state cost of a 64-deep k-step, the
intercept the per-launch fixed cost (prologue, epilogue, launch ramp).  gemm_kscan.py M N"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
M, N = int(sys.argv[1]), int(sys.argv[2])


def time_it(fn, iters=15):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


Ks = (256, 768, 1536, 3072, 6144)
print(f"M={M} N={N}   us at K = {Ks};  slope us/k-step (3072..6144), intercept")
for tile, name in ((1, "128x128 2st"), (4, "128x128 ring4"), (8, "WS128x128"), (2, "256x128 2st"), (5, "256x128 ring3"), (9, "WS256x128"), (3, "256x256 2st"), (10, "128x96"), (11, "160x128")):
    if (tile == 3 and N % 256) or (tile == 10 and N % 96):
        continue
    ts = []
    for K in Ks:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st())
        ts.append(time_it(fn))
    slope = (ts[4] - ts[3]) / ((Ks[4] - Ks[3]) / 64)
    print(f"{name:14s} " + " ".join(f"{t:7.1f}" for t in ts) + f"   slope {slope:.3f}  intercept {ts[3] - slope * Ks[3] / 64:6.1f}   ({2.0 * M * N * 64 / slope / 1e6:.0f} TF/s marginal)", flush=True)
