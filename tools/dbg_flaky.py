"""count how often the 3-step run lands on the alternate step-3 loss; argv: n_trainers sync(0/1)"""
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
torch.cuda.set_device(0)
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=64, dropout=0.05)
arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130)
ids, mask, px, oh = synthetic_batch(300, 3, 4, 64, 77, pad=True)
n = int(sys.argv[1]); sync = int(sys.argv[2])
seen = {}
for i in range(n):
    tr = MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5)
    np.random.seed(30)
    ls = []
    for step in (1, 2, 3):
        loss, _ = tr.train_step(ids.cuda(), mask.cuda(), px, oh, None, 1e-3, 0.00025, step)
        if sync:
            torch.cuda.synchronize()
        ls.append(loss)
    key = tuple("%.7f" % float(l[0]) for l in ls)
    seen[key] = seen.get(key, 0) + 1
print(os.environ.get("TAG"), "sync", sync, seen, flush=True)
