import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import smtc_amd
from oracle import mm_oracle as O
from smtc_amd.mm_late import MM_Model
torch.set_num_threads(16)
B, T = int(sys.argv[1]), int(sys.argv[2])
cfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=400, max_pos=130, num_labels=4, p_hidden=0.0, p_attn=0.0, p_head=0.0)
P = O.make_params(cfg, 9)
ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 100 + B, True)
print("lens", mask.sum(1).tolist())
Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
r = O.mm_forward(Pg, ids, mask, pixels, cfg, None)
ref = O.mix_loss(r[0], onehot, None, r[1], None, None, True, False); ref.backward()
for dt in ("f16", "bf16"):
    arch = dict(layers_txt=2, layers_img=1, vocab=400, max_pos=130, type_vocab=1, p_hidden=0.0, p_attn=0.0)
    m = MM_Model(4, "bernice", "vit", 0.0, "attention", arch=arch, dtype=dt, max_posts=B, max_text_len=T)
    m.load_state_dict(P, strict=False); m.train()
    out_cls, lpt, _, _, feats = m(ids, mask, pixels)
    loss = O.mix_loss(out_cls, onehot.cuda(), None, lpt, None, None, True, False); loss.backward()
    errs = []
    for k, p in m.named_parameters():
        if Pg[k].grad is None or p.grad is None: continue
        g = Pg[k].grad
        errs.append(((p.grad.cpu() - g).norm().item() / max(g.norm().item(), 1e-30), g.norm().item(), k))
    errs.sort(reverse=True)
    print(dt, "loss", loss.item(), ref.item())
    for e in errs[:8]: print("   %.4f  |g|=%.3e  %s" % e)
