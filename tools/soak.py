"""Soak: a few hundred fused training steps with ITC + ITM through the DataLoader / prefetcher path; prints loss, device memory and host
RSS over time (looking for NaNs, leaks, drift)."""
import os, sys, types, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import SyntheticPosts
cfg = types.SimpleNamespace(batch_size=64, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", seed=0)
a = tr.model.arch
ds = SyntheticPosts(64 * 40, a["vocab"], 3, 128, 11, a["txt_kind"], a["pad_id"], a["image"])
dl = torch.utils.data.DataLoader(ds, batch_size=64, shuffle=True, drop_last=True)
np.random.seed(30)
step, t0 = 0, time.time()
for epoch in range(8):
    for batch in tr._device_batches(dl):
        ids, mask, px = tr._unpack(batch)
        step += 1
        loss, nc = tr.train_step(ids, mask, px, batch["labels"], None, 1e-5, 0.00025, step)
        if step % 40 == 0:
            l = loss.tolist()
            print(f"step {step:4d} loss {l[0]:.4f} (cls {l[1]:.4f} itc {l[2]:.4f} itm {l[3]:.4f}) dev {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved "
                  f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:.2f} GiB  {time.time() - t0:.0f}s", flush=True)
            assert all(np.isfinite(l)), l
print("done", step, "steps")
