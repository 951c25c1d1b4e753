"""host enqueue time vs GPU time of one early-fusion (LXMERT) train step"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_early import MMEarly_Model
aux = len(sys.argv) > 1
cfg = types.SimpleNamespace(batch_size=32, num_labels=3, use_clip_loss=aux, beta_itc=0.1, use_tim_loss=aux, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMEarly_Model(cfg, "lxmert", seed=0)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1, 30522, (32, 128), generator=g).cuda(); mask = torch.ones(32, 128, dtype=torch.int64).cuda(); tt = torch.zeros_like(ids)
feats = (torch.rand(32, 36, 2048, generator=g) * 2).cuda(); boxes = torch.rand(32, 36, 4, generator=g).cuda()
onehot = torch.nn.functional.one_hot(torch.randint(0, 3, (32,), generator=g), 3).cuda()
np.random.seed(30)
for s in range(1, 4):
    tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, s)
torch.cuda.synchronize()
for s in range(4, 8):
    t0 = time.perf_counter()
    tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.2f ms, then GPU drain %.2f ms (total %.2f)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), flush=True)
