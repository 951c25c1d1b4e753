"""two steps, then: forward / extra weight refresh / forward -- does the extra refresh change anything?"""
import os, sys, types, hashlib, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smtc_amd
from smtc_amd.mm_late import MMLate_Model
from smtc_amd.synthetic import synthetic_batch
torch.cuda.set_device(0)
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=64, dropout=0.05)
arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130)
ids, mask, px, oh = synthetic_batch(300, 3, 4, 64, 77, pad=True)
ids, mask = ids.cuda(), mask.cuda()
h = lambda t: hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:8]
for trial in range(24):
    tr = MMLate_Model(cfg, "bernice", "vit", "attention", arch=arch, seed=5)
    m = tr.model
    np.random.seed(30)
    for step in (1, 2):
        tr.train_step(ids, mask, px, oh, None, 1e-3, 0.00025, step)
    torch.cuda.synchronize()
    tim = tr.prepare_itm_inputs(ids, mask)
    m.train(True)
    a = m._engine_forward(ids, mask, px, tim[0], tim[1], seed=1234567)
    torch.cuda.synchronize()
    ha = [h(t) for t in a]
    m._refresh_weights(3)
    b = m._engine_forward(ids, mask, px, tim[0], tim[1], seed=1234567)
    torch.cuda.synchronize()
    hb = [h(t) for t in b]
    em, ev = tr._moments()
    print("trial", trial, "train", h(m._flat_train), "m", h(em), "v", h(ev), "rows", h(m._word_row_state), "fwd", ha, "after refresh", hb if hb != ha else "same", flush=True)
