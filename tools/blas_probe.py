#!/usr/bin/env python3
"""What the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the model's GEMM shapes, next to this repo's NT kernel."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def time_it(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


for name, M, N, K in (("txt qkv", 8192, 2304, 768), ("txt ao", 8192, 768, 768), ("txt fc1", 8192, 3072, 768), ("txt fc2", 8192, 768, 3072), ("dx_qkv", 8192, 768, 2304),
                      ("vit qkv", 12608, 2304, 768), ("vit ao", 12608, 768, 768), ("vit fc1", 12608, 3072, 768), ("vit fc2", 12608, 768, 3072), ("sq 4096", 4096, 4096, 4096)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_lib = time_it(lambda: torch.matmul(A, B.t(), out=Cm))
    t_mine = time_it(lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 0, st()))
    fl = 2.0 * M * N * K
    print(f"{name:8s} {M:6d}x{N:5d}x{K:5d}  library {t_lib:7.1f} us {fl / t_lib / 1e6:6.0f} TF | mmhip {t_mine:7.1f} us {fl / t_mine / 1e6:6.0f} TF", flush=True)
