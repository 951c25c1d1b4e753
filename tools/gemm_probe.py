#!/usr/bin/env python3
"""timing experiments: (1) NT fixed overhead: sweep K at fixed M,N; (2) TN grouped layer launch, variants, and the
'no K advance' debug mode (all operand tiles L2-resident)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd
from smtc_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def time_it(fn, iters=20):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3
print("NT K sweep (128x128 tile): us")
for M, N in ((8192, 768), (8192, 3072), (8192, 2304), (12608, 768)):
    row = []
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 1 << 4, st())
        row.append(f"K={K}:{time_it(fn):6.1f}")
    print(f"M={M} N={N}  " + "  ".join(row), flush=True)
print("empty-kernel launch pair baseline")
x = torch.zeros(1024, device=dev)
print("  torch tiny op us", time_it(lambda: x.add_(1.0)))
print("TN single GEMM, accumulate modes (0 store, 2 = no K advance debug)")
for name, M, Nn, Nc in [("dW fc1", 8192, 3072, 768), ("dW ao", 8192, 768, 768)]:
    A = (torch.randn(M, Nn, device=dev) * 0.1).to(torch.bfloat16); B = (torch.randn(M, Nc, device=dev) * 0.5).to(torch.bfloat16)
    Cm = torch.empty(Nn, Nc, device=dev)
    for var in (1, 3):
        for acc in (0, 2):
            fn = lambda: lib.mmhip_op_gemm_tn(0, p(A), Nn, p(B), Nc, p(Cm), Nc, M, Nn, Nc, acc, var << 4, None, st())
            print(f"  {name} variant {var} acc {acc}: {time_it(fn):7.1f} us", flush=True)
