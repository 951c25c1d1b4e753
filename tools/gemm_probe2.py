#!/usr/bin/env python3
"""does padding the leading dimensions (breaking 256-B-multiple row strides) change GEMM time?"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd
from smtc_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def time_it(fn, iters=30):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3
def mat(rows, cols, pad, scale):
    buf = (torch.randn(rows, cols + pad, device=dev) * scale).to(torch.bfloat16)
    return buf, cols + pad
print("TN dW, pad elements on both operands' leading dims (variant 3 = 256x128 ring, variant 1 = 128x128)")
for name, M, Nn, Nc in [("dW fc1", 8192, 3072, 768), ("dW fc2", 8192, 768, 3072), ("dW qkv", 8192, 2304, 768)]:
    for pad in (0, 8, 64, 72, 136):
        A, lda = mat(M, Nn, pad, 0.1); B, ldb = mat(M, Nc, pad, 0.5)
        Cm = torch.empty(Nn, Nc, device=dev)
        r = []
        for var in (1, 3):
            fn = lambda: lib.mmhip_op_gemm_tn(0, p(A), lda, p(B), ldb, p(Cm), Nc, M, Nn, Nc, 0, var << 4, None, st())
            r.append(f"v{var} {time_it(fn):6.1f}us")
        print(f"  {name} pad {pad:3d}: " + "  ".join(r), flush=True)
print("NT, pad on A and B leading dims (128x128)")
for name, M, N, K in [("fc2", 8192, 768, 3072), ("fc1", 8192, 3072, 768), ("ao", 8192, 768, 768), ("dx_qkv", 8192, 768, 2304)]:
    for pad in (0, 8, 64, 72):
        A, lda = mat(M, K, pad, 0.5); B, ldb = mat(N, K, pad, 0.05)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), lda, p(B), ldb, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 1 << 4, st())
        print(f"  {name} pad {pad:3d}: {time_it(fn):6.1f}us", flush=True)
