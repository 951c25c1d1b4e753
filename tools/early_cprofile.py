import os, sys, types, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_early import MMEarly_Model
cfg = types.SimpleNamespace(batch_size=32, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMEarly_Model(cfg, "lxmert", seed=0)
g = torch.Generator().manual_seed(1)
ids = torch.randint(1, 30522, (32, 128), generator=g).cuda(); mask = torch.ones(32, 128, dtype=torch.int64).cuda(); tt = torch.zeros_like(ids)
feats = (torch.rand(32, 36, 2048, generator=g) * 2).cuda(); boxes = torch.rand(32, 36, 4, generator=g).cuda()
onehot = torch.nn.functional.one_hot(torch.randint(0, 3, (32,), generator=g), 3).cuda()
for s in range(1, 4):
    tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, s)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for s in range(4, 8):
    tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-5, 0.00025, s)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
