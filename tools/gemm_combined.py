#!/usr/bin/env python3
"""What a cross-tower grouped NT GEMM could give: the text (M=8192) and ViT (M=12608) GEMMs of one layer as ONE launch of
M=20800 rows (timing stand-in: same weights for both halves), per tile variant, against the two separate launches."""
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
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


for name, N, K in (("qkv", 2304, 768), ("ao", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    row = []
    for M in (8192, 12608, 20800):
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev)
        cells = []
        for tile in (1, 2, 3, 9, 5):
            if tile == 3 and N % 256:
                cells.append("      -"); continue
            fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st())
            cells.append(f"{time_it(fn):7.1f}")
        row.append(f"M={M}: " + " ".join(cells))
    print(f"{name:4s} (tiles 128^2, 256x128, 256^2, WS256x128, ring256x128) us | " + " | ".join(row), flush=True)
