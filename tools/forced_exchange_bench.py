"""Step time with the data-parallel exchange forced through RCCL at world size 1 (MMHIP_FORCE_EXCHANGE=1): the cost of the
collective call pattern itself (launches, copies, the row-sparse word-table exchange) on a one-GPU box."""
import os, sys, types, time
os.environ["MMHIP_FORCE_EXCHANGE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29777", rank=0, world_size=1)
cfg = types.SimpleNamespace(batch_size=64, num_labels=2, use_clip_loss=False, beta_itc=None, use_tim_loss=False, beta_itm=None, max_length=128, dropout=0.05)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", seed=0)
a = tr.model.arch
ids, mask, px, oh = synthetic_batch(a["vocab"], 2, 64, 128, 1234, a["txt_kind"], a["pad_id"], False, a["image"], tr.device)
for s in range(1, 6):
    tr.train_step(ids, mask, px, oh, None, 1e-5, 0.00025, s)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(6, 26):
    tr.train_step(ids, mask, px, oh, None, 1e-5, 0.00025, s)
torch.cuda.synchronize()
print("forced-exchange (RCCL, world 1) ms/step", round((time.perf_counter() - t0) / 20 * 1e3, 3))
torch.distributed.destroy_process_group()
