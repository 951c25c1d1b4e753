#!/usr/bin/env python3
"""launch one NT GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one.py M N K tile [reps] [cold]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
M, N, K, tile = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
cold = len(sys.argv) > 6
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
flush = torch.empty(128 * 1024 * 1024, device=dev) if cold else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    if cold:
        flush.fill_(1.0)
    lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st)
torch.cuda.synchronize()
