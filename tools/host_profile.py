"""Host-side cost of enqueuing one training step (no sync inside the loop): cProfile over N steps."""
import cProfile, pstats, sys, os, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch

aux = "--aux" in sys.argv
cfg = types.SimpleNamespace(batch_size=64, num_labels=3 if aux else 2, use_clip_loss=aux, beta_itc=0.1, use_tim_loss=aux, beta_itm=0.1, max_length=128, dropout=0.05)
tr = MMLate_Model(cfg, "bernice", "vit", "attention", seed=0)
a = tr.model.arch
dev = tr.device
ids, mask, pixels, onehot = synthetic_batch(a["vocab"], cfg.num_labels, 64, 128, 1234, a["txt_kind"], a["pad_id"], False, a["image"], dev)
np.random.seed(30)
for s in range(1, 4):
    tr.train_step(ids, mask, pixels, onehot, None, 1e-5, 0.00025, s)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for s in range(4, 14):
    tr.train_step(ids, mask, pixels, onehot, None, 1e-5, 0.00025, s)
pr.disable()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step", host / 10 * 1e3, "total ms/step", (time.perf_counter() - t0) / 10 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
