#!/usr/bin/env python3
"""GPU micro-benchmark of the deep-pipelined NT GEMM (csrc/gemm8.hip) against the round-1 tiles on the model's shapes.
Interleaved rounds in one process (variants x rounds), median per-launch time from HIP events, random bf16 operands."""
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
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())

VARIANTS = [("r1 auto", 0), ("128x128", 1), ("role 256x128", 9), ("role 256x96", 12), ("dp 256x256", 13), ("dp 256x256 P", 15), ("dp 256x128 P", 16), ("dp 256x192 P", 18)]
if os.environ.get("VARIANTS"):
    VARIANTS = [v for v in VARIANTS if str(v[1]) in os.environ["VARIANTS"].split(",")]


def main():
    rounds = int(os.environ.get("ROUNDS", "15"))
    epi = os.environ.get("EPI", "bias")
    shapes = [("txt qkv", 8192, 2304, 768), ("txt ao", 8192, 768, 768), ("txt fc1", 8192, 3072, 768), ("txt fc2", 8192, 768, 3072),
              ("txt dx_qkv", 8192, 768, 2304), ("vit qkv", 12608, 2304, 768), ("vit ao", 12608, 768, 768), ("vit fc1", 12608, 3072, 768),
              ("vit fc2", 12608, 768, 3072), ("itm qkv", 16384, 2304, 768), ("itm fc1", 16384, 3072, 768), ("itm fc2", 16384, 768, 3072),
              ("fc1 K3072", 8192, 3072, 3072), ("vfc1 K3072", 12608, 3072, 3072), ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]
    if os.environ.get("SHAPES"):
        shapes = [s for s in shapes if s[0] in os.environ["SHAPES"].split(",")]
    cold = os.environ.get("COLD", "0") == "1"       # evict L2 / Infinity Cache before every timed launch (512 MiB streamed write)
    flush = torch.empty(128 * 1024 * 1024, device=dev) if cold else None
    print(f"epilogue={epi} rounds={rounds} cold={cold}")
    print(f"{'shape':12s} {'M':>6} {'N':>5} {'K':>5} | " + " | ".join(f"{n:>16s}" for n, _ in VARIANTS))
    for name, M, N, K in shapes:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        Cm = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        resid = torch.randn(M, N, device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        fns = []
        for _, tile in VARIANTS:
            if (tile in (13, 15) and N % 256) or (tile in (17, 18) and N % 192) or (tile == 12 and N % 96):
                fns.append(None)
                continue
            if epi == "gelu":
                fns.append(lambda tile=tile: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 1, p(aux), N, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st()))
            elif epi == "gelunoaux":
                fns.append(lambda tile=tile: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 1, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st()))
            elif epi == "gelugrad":
                fns.append(lambda tile=tile: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, None, 0, None, 0, p(aux), N, 0.0, 0, 0, None, 0, 0, tile << 4, st()))
            elif epi == "resid":
                fns.append(lambda tile=tile: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.1, 5, 7, p(resid), N, 0, tile << 4, st()))
            else:
                fns.append(lambda tile=tile: lib.mmhip_op_gemm_nt(0, p(A), K, p(B), K, p(Cm), N, M, N, K, p(bias), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, tile << 4, st()))
        times = [[] for _ in fns]
        for f in fns:
            if f:
                f()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for i, f in enumerate(fns):
                if f is None:
                    continue
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if cold:
                    flush.fill_(1.0)
                a.record()
                f()
                b.record()
                times[i].append((a, b))
        torch.cuda.synchronize()
        cells = []
        for i, f in enumerate(fns):
            if f is None:
                cells.append(f"{'-':>16s}")
                continue
            ts = sorted(a.elapsed_time(b) for a, b in times[i])
            us = ts[len(ts) // 2] * 1e3
            cells.append(f"{us:7.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF")
        print(f"{name:12s} {M:6d} {N:5d} {K:5d} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
