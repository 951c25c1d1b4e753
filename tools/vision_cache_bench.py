"""Step time of the steady state of multi-epoch training with the image-tower output cache on (every post seen before):
NOT the benchmark of record (bench.py computes the tower every step) -- it shows what epochs after the first cost."""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
cfg = types.SimpleNamespace(batch_size=64, num_labels=2, use_clip_loss=False, beta_itc=None, use_tim_loss=False, beta_itm=None, max_length=128, dropout=0.05)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", seed=0)
a = tr.model.arch
ids, mask, px, oh = synthetic_batch(a["vocab"], 2, 64, 128, 1234, a["txt_kind"], a["pad_id"], False, a["image"], tr.device)
tr.model.enable_vision_cache(64)
keys = list(range(64))
for mode, k in (("tower computed every step", None), ("tower outputs from the HBM cache", keys)):
    for s in range(1, 6):
        tr.train_step(ids, mask, px, oh, None, 1e-5, 0.00025, s, vision_keys=k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(6, 26):
        tr.train_step(ids, mask, px, oh, None, 1e-5, 0.00025, s, vision_keys=k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{mode}: {ms:.3f} ms/step = {64 / ms * 1e3:.0f} posts/s")
