#!/usr/bin/env python3
"""run one GEMM variant a few times (for rocprofv3 --pmc):  one_gemm.py nt|tn M N K variant [iters]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib

kind, M, N, K, var = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
if kind == "nt":
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(iters):
        lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, var << 4, st)
else:   # tn: M = reduction, N = Nn, K = Nc
    A = (torch.randn(M, N, device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    Cm = torch.empty(N, K, device=dev)
    for _ in range(iters):
        lib.mmhip_op_gemm_tn(0, p(A), N, p(B), K, p(Cm), K, M, N, K, 0, var << 4, None, st)
torch.cuda.synchronize()
