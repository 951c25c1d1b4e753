#!/usr/bin/env python3
"""Can the early-fusion (config 5) training step be captured into a hipGraph, and what does replay gain over eager enqueue?
Forward + loss + backward + weight-gradient flush of MMEarly_Model on a resident batch (bs = 32, T = 128, 36 x 2048 ROI features);
AdamW stays outside the captured region here.  Prints eager / replay ms per step."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd  # noqa: F401
from smtc_amd.mm_early import MMEarly_Model

B, T, NB, C = 32, 128, 36, 3
cfg = types.SimpleNamespace(batch_size=B, num_labels=C, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=T, dropout=0.05)
tr = MMEarly_Model(cfg, "lxmert", dtype="bf16", seed=0)
m = tr.model
m.train()
g = torch.Generator().manual_seed(1234)
ids = torch.randint(1, 30522, (B, T), generator=g).cuda()
mask = torch.ones(B, T, dtype=torch.int64).cuda()
tt = torch.zeros_like(ids)
feats = (torch.rand(B, NB, 2048, generator=g) * 2).cuda()
boxes = torch.rand(B, NB, 4, generator=g).cuda()
onehot = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).cuda()


def body():
    m._wsig = None
    m.oc.cache.clear()
    out, et, ev, otim = m(ids, mask, tt, feats, boxes, tim_inputs=None)
    loss = tr.loss(out, onehot, None, et, ev, otim, None)
    loss.backward()
    m.finish_backward()
    return loss.detach()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m._flat_grad.zero_()
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
eager = timed(lambda: (m._flat_grad.zero_(), body()))
graph = torch.cuda.CUDAGraph()
m._flat_grad.zero_()
try:
    with torch.cuda.graph(graph):
        loss = body()
except Exception as e:          # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:500])
    sys.exit(1)
torch.cuda.synchronize()
g0 = m._flat_grad.clone()
m._flat_grad.zero_()
graph.replay()
torch.cuda.synchronize()
print("replay reproduces the captured gradient:", torch.equal(g0, m._flat_grad), "loss", float(loss))
replay = timed(lambda: (m._flat_grad.zero_(), graph.replay()))
print(f"fwd+bwd eager {eager:.2f} ms, hipGraph replay {replay:.2f} ms")
