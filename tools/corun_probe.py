#!/usr/bin/env python3
"""Do an HBM-bound kernel (fused AdamW over 96 M parameters) and an MFMA GEMM share the chip gracefully?  Times each alone and
both started together on two streams."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
n = 96_000_000
P_, G_, M_, V_ = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
V_.abs_()
M, N, K = 8192, 3072, 768
A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sp = lambda s: C.c_void_p(s.cuda_stream)
REP = 10                                      # GEMM launches per AdamW launch (~0.5 ms each side)


def adamw(s):
    lib.mmhip_adamw(p(P_), p(G_), p(M_), p(V_), n, 1e-5, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 0, sp(s))


def gemms(s):
    for _ in range(REP):
        lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 1 << 4, sp(s))


def timed(fn):
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[3] * 1e3


adamw(s1); gemms(s2); torch.cuda.synchronize()
ta = timed(lambda: adamw(s1))
tg = timed(lambda: gemms(s2))
tb = timed(lambda: (adamw(s1), gemms(s2)))
print(f"AdamW alone {ta:.3f} ms, {REP} GEMMs alone {tg:.3f} ms, together {tb:.3f} ms  (sum {ta + tg:.3f}, max {max(ta, tg):.3f})")
