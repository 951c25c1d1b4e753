#!/usr/bin/env python3
"""GPU micro-benchmark of the MFMA GEMM kernels on the model's shapes, all tile variants, random bf16 data.
Interleaved rounds in one process (variants x rounds), median of per-launch times from HIP events."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def time_it(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3   # us


def main():
    shapes = [("txt qkv", 8192, 2304, 768), ("txt ao", 8192, 768, 768), ("txt fc1", 8192, 3072, 768), ("txt fc2", 8192, 768, 3072),
              ("txt dx_qkv", 8192, 768, 2304), ("vit qkv", 12608, 2304, 768), ("vit ao", 12608, 768, 768), ("vit fc1", 12608, 3072, 768),
              ("vit fc2", 12608, 768, 3072), ("itm qkv", 16384, 2304, 768), ("itm ao", 16384, 768, 768), ("itm fc2", 16384, 768, 3072),
              ("square 4096", 4096, 4096, 4096)]
    if "--early" in sys.argv:      # config 5 (LXMERT): 32 posts x 128 tokens = 4096 language rows, 32 x 36 = 1152 vision rows
        shapes = [("lang qkv", 4096, 2304, 768), ("lang ao", 4096, 768, 768), ("lang fc1", 4096, 3072, 768), ("lang fc2", 4096, 768, 3072),
                  ("lang kv", 4096, 1536, 768), ("lang dqkv", 4096, 768, 2304), ("visn q", 1152, 768, 768), ("visn qkv", 1152, 2304, 768),
                  ("visn fc1", 1152, 3072, 768), ("visn fc2", 1152, 768, 3072), ("visn feat", 1152, 768, 2048)]
        print(f"{'shape':14s} {'M':>6} {'N':>5} {'K':>5} | " + " | ".join(f"{n:>18s}" for n in ("128x128 2-stage", "128x128 ring3", "128x128 ring4", "WS 256x128", "auto")))
        for name, M, N, K in shapes:
            A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            bias = torch.randn(N, device=dev)
            cells = []
            for tile in (1, 21, 20, 9, 0):
                fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st())
                us = time_it(fn)
                cells.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
            print(f"{name:14s} {M:6d} {N:5d} {K:5d} | " + " | ".join(cells), flush=True)
        return
    print(f"{'shape':14s} {'M':>6} {'N':>5} {'K':>5} | " + " | ".join(f"{n:>18s}" for n in ("128x128", "128x96", "128x192", "WS 256x128", "auto")))
    for name, M, N, K in shapes:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev)
        cells = []
        for tile in (1, 10, 6, 9, 0):
            if (tile == 3 and N % 256) or (tile in (6, 7) and N % 192) or (tile == 10 and N % 96):
                cells.append(f"{'-':>18s}")
                continue
            fn = lambda: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st())
            us = time_it(fn)
            cells.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
        print(f"{name:14s} {M:6d} {N:5d} {K:5d} | " + " | ".join(cells), flush=True)
    print("\nTN (dW = dY^T X), fp32 out")
    for name, M, Nn, Nc in [("dW fc1", 8192, 3072, 768), ("dW fc2", 8192, 768, 3072), ("dW qkv", 8192, 2304, 768), ("dW ao", 8192, 768, 768),
                            ("dW fc1 itm", 16384, 3072, 768)]:
        A = (torch.randn(M, Nn, device=dev) * 0.1).to(torch.bfloat16)
        B = (torch.randn(M, Nc, device=dev) * 0.5).to(torch.bfloat16)
        Cm = torch.empty(Nn, Nc, device=dev)
        cells = []
        for var in (1, 2, 3):
            fn = lambda: lib.mmhip_op_gemm_tn(0, p(A), Nn, p(B), Nc, p(Cm), Nc, M, Nn, Nc, 0, var << 4, None, st())
            us = time_it(fn)
            cells.append(f"{us:7.1f}us {2.0 * M * Nn * Nc / us / 1e6:6.0f}TF")
        print(f"{name:14s} M={M} {Nn}x{Nc}: " + " | ".join(cells) + "   (128x128 2-stage | 128x128 ring4 | 256x128 ring3)", flush=True)


if __name__ == "__main__":
    main()
