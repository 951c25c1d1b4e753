#!/usr/bin/env python3
"""attention forward / backward kernel times at the model's shapes, dropout on / off (HIP events, median of 20)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def time_it(fn, iters=20):
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3


for name, posts, S, heads in (("text", 64, 128, 12), ("itm", 128, 128, 12), ("vit", 64, 197, 12)):
    H = heads * 64
    qkv = torch.randn(posts * S, 3 * H, device=dev).to(torch.bfloat16)
    ctx = torch.zeros(posts * S, H, dtype=torch.bfloat16, device=dev)
    dctx = torch.randn(posts * S, H, device=dev).to(torch.bfloat16)
    dqkv = torch.zeros(posts * S, 3 * H, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(posts, heads, S, device=dev)
    mb = torch.zeros(posts, S, device=dev)
    row = [name]
    for pd in (0.0, 0.1):
        f = lambda: lib.mmhip_op_attn_fwd(0, p(qkv), p(mb), p(ctx), p(lse), posts, S, heads, pd, 7, 3, st())
        row.append(f"fwd p={pd}: {time_it(f):6.1f}us")
        if S <= 128:
            b = lambda: lib.mmhip_op_attn_bwd(0, p(qkv), p(mb), p(ctx), p(dctx), p(lse), p(dqkv), posts, S, heads, pd, 7, 3, st())
            row.append(f"bwd p={pd}: {time_it(b):6.1f}us")
    fl = 4.0 * S * S * 64 * heads * posts
    row.append(f"(fwd {fl / 1e9:.1f} GF, bwd {2.5 * fl / 1e9:.1f} GF)")
    print("  ".join(row), flush=True)
