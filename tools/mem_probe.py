"""Device-memory growth probe: reserved / allocated bytes over steps for (A) a static device batch, (B) CPU batches through DevicePrefetcher."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
from smtc_amd.image_processing import DevicePrefetcher
aux = "--aux" in sys.argv
cfg = types.SimpleNamespace(batch_size=64, num_labels=3, use_clip_loss=aux, beta_itc=0.1, use_tim_loss=aux, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", seed=0)
a = tr.model.arch
np.random.seed(30)
gib = lambda x: x / 2**30
ids, mask, px, oh = synthetic_batch(a["vocab"], 3, 64, 128, 1, a["txt_kind"], a["pad_id"], False, a["image"], "cpu")
step = 0
print("A: static device batch")
d = [t.to(tr.device) for t in (ids, mask, px, oh)]
for i in range(200):
    step += 1
    tr.train_step(d[0], d[1], d[2], d[3], None, 1e-5, 0.00025, step)
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize(); print(f"  step {i + 1}: allocated {gib(torch.cuda.memory_allocated()):.3f} reserved {gib(torch.cuda.memory_reserved()):.3f} GiB", flush=True)
print("B: CPU batches through DevicePrefetcher")
batches = [{"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "pixel_values": px.unsqueeze(1), "labels": oh, "data_id": torch.arange(64)} for _ in range(50)]
for rep in range(4):
    for b in DevicePrefetcher(batches, tr.device, None):
        i_, m_, p_ = tr._unpack(b)
        step += 1
        tr.train_step(i_, m_, p_, b["labels"], None, 1e-5, 0.00025, step)
    torch.cuda.synchronize(); print(f"  {50 * (rep + 1)} batches: allocated {gib(torch.cuda.memory_allocated()):.3f} reserved {gib(torch.cuda.memory_reserved()):.3f} GiB", flush=True)
