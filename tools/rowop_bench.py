#!/usr/bin/env python3
"""GPU micro-benchmark of the non-GEMM kernels of the step -- LayerNorm forward / backward and self-attention forward / backward -- each alone on
the chip, in the three element forms the engines use: bf16 (0), fp32 (2) and plane pairs (3: the parity mode's tensors).  Median per-launch time
from HIP events over interleaved rounds, the algorithmic bytes of each launch and the HBM fraction they imply (8 TB/s spec).
    python tools/rowop_bench.py [--rounds 15] > profiles/rNN_rowop_bench.txt"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401,E402
from smtc_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda:0")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
NAME = {0: "bf16", 2: "fp32", 3: "pair"}


def timed(fns, rounds):
    ev = [[(torch.cuda.Event(True), torch.cuda.Event(True)) for _ in fns] for _ in range(rounds)]
    for f in fns:
        f(); f()
    torch.cuda.synchronize()
    for r in range(rounds):
        for i, f in enumerate(fns):
            ev[r][i][0].record()
            rc = f()
            ev[r][i][1].record()
            assert rc == 0, rc
    torch.cuda.synchronize()
    out = []
    for i in range(len(fns)):
        ts = sorted(ev[r][i][0].elapsed_time(ev[r][i][1]) * 1e3 for r in range(rounds))
        out.append(ts[len(ts) // 2])
    return out


def elem(dt):
    return torch.bfloat16 if dt == 0 else torch.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=15)
    a = ap.parse_args()
    H = 768
    print(f"# rounds={a.rounds}; us = median per launch, kernel alone on the chip; MB = algorithmic bytes (read + written); frac = of 8 TB/s")
    print("## LayerNorm forward (rows x 768)")
    for rows in (12608, 8192):
        fns, tags, mbs = [], [], []
        g, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
        keep = []
        for dt in (0, 2, 3):
            x = torch.randn(rows, H, device=dev).to(elem(dt))
            y = torch.empty(rows, H, device=dev, dtype=torch.float32 if dt else torch.bfloat16)
            mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
            keep += [x, y, mean, rstd]
            fns.append(lambda dt=dt, x=x, y=y, mean=mean, rstd=rstd: lib.mmhip_op_layernorm_fwd(dt, p(x), p(y), p(g), p(b), p(mean), p(rstd), rows, H, 1e-5, st()))
            tags.append(NAME[dt])
            mbs.append(rows * H * (x.element_size() + y.element_size()) / 1e6)
        for tag, us, mb in zip(tags, timed(fns, a.rounds), mbs):
            print(f"ln_fwd  rows {rows:6d} {tag:5s} {us:8.1f} us  {mb:7.1f} MB  frac {mb * 1e6 / (us * 1e-6) / 8e12:5.2f}")
    print("## LayerNorm backward (rows x 768; dy, x in; dx out; dgamma / dbeta accumulated)")
    for rows in (8192,):
        fns, tags, mbs = [], [], []
        keep = []
        for dt in (0, 2):
            x = torch.randn(rows, H, device=dev).to(elem(dt))
            dy = torch.randn(rows, H, device=dev).to(elem(dt))
            dx = torch.empty_like(x)
            mean, rstd = torch.zeros(rows, device=dev), torch.ones(rows, device=dev)
            dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
            g = torch.randn(H, device=dev)
            keep += [x, dy, dx, mean, rstd, dg, db, g]
            fns.append(lambda dt=dt, x=x, dy=dy, dx=dx, mean=mean, rstd=rstd, dg=dg, db=db, g=g:
                       lib.mmhip_op_layernorm_bwd(dt, p(dy), p(x), p(g), p(mean), p(rstd), p(dx), None, p(dg), p(db), rows, H, st()))
            tags.append(NAME[dt])
            mbs.append(rows * H * 3 * x.element_size() / 1e6)
        for tag, us, mb in zip(tags, timed(fns, a.rounds), mbs):
            print(f"ln_bwd  rows {rows:6d} {tag:5s} {us:8.1f} us  {mb:7.1f} MB  frac {mb * 1e6 / (us * 1e-6) / 8e12:5.2f}")
    print("## self-attention (posts x S x 12 heads x 64), forward and backward")
    for posts, S, pdrop in ((64, 197, 0.0), (64, 128, 0.1), (32, 577, 0.0)):
        heads = 12
        rows = posts * S
        fns, tags = [], []
        keep = []
        for dt in (0, 2, 3):
            w = 2 if dt == 3 else 1
            et = torch.bfloat16 if dt in (0, 3) else torch.float32
            qkv = (torch.randn(rows, 3 * H * w, device=dev) * 0.5).to(et)
            ctx = torch.empty(rows, H * w, device=dev, dtype=et)
            lse = torch.empty(posts * heads * S, device=dev)
            keep += [qkv, ctx, lse]
            fns.append(lambda dt=dt, qkv=qkv, ctx=ctx, lse=lse: lib.mmhip_op_attn_fwd(dt, p(qkv), None, p(ctx), p(lse), posts, S, heads, pdrop, 7, 3, st()))
            tags.append("fwd " + NAME[dt])
            if S <= 128:
                dctx = (torch.randn(rows, H * w, device=dev) * 0.1).to(et)
                dqkv = torch.empty_like(qkv)
                keep += [dctx, dqkv]
                fns.append(lambda dt=dt, qkv=qkv, ctx=ctx, lse=lse, dctx=dctx, dqkv=dqkv:
                           lib.mmhip_op_attn_bwd(dt, p(qkv), None, p(ctx), p(dctx), p(lse), p(dqkv), posts, S, heads, pdrop, 7, 3, st()))
                tags.append("bwd " + NAME[dt])
        for tag, us in zip(tags, timed(fns, a.rounds)):
            print(f"attn posts {posts} S {S} p_drop {pdrop} {tag:9s} {us:8.1f} us")


if __name__ == "__main__":
    main()
