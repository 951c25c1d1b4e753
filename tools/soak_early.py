"""Soak of the early-fusion (LXMERT) path: a few hundred train steps with ITC + ITM and dropout; loss, device memory over time."""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_early import MMEarly_Model
cfg = types.SimpleNamespace(batch_size=32, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMEarly_Model(cfg, "lxmert", seed=0)
g = torch.Generator().manual_seed(1)
N = 32 * 8
ids = torch.randint(1, 30522, (N, 128), generator=g); lens = torch.randint(8, 129, (N,), generator=g)
mask = (torch.arange(128)[None] < lens[:, None]).long(); ids = ids * mask
feats = torch.rand(N, 36, 2048, generator=g) * 2; boxes = torch.rand(N, 36, 4, generator=g)
lab = torch.randint(0, 3, (N,), generator=g); onehot = torch.nn.functional.one_hot(lab, 3)
np.random.seed(30)
t0 = time.time()
for step in range(1, 241):
    i = (step % 8) * 32
    sl = slice(i, i + 32)
    loss = tr.train_step(ids[sl], mask[sl], torch.zeros_like(ids[sl]), feats[sl], boxes[sl], onehot[sl], None, 2e-5, 0.00025, step)
    if step % 40 == 0:
        l = float(loss)
        print(f"step {step:4d} loss {l:.4f} dev {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB {time.time() - t0:.0f}s", flush=True)
        assert np.isfinite(l)
print("done")
