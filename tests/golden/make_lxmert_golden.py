#!/usr/bin/env python3
"""Golden vectors for the early-fusion LXMERT path (BASELINE config 5), produced by the REFERENCE's own
`models/mm_early.py:Lxmert` module (this container only) around HF `LxmertModel`, with the deterministic weights of
`oracle.lxmert_oracle.make_params` loaded into it.

`mm_early.py:10-12` imports `lxmert_scripts.*`, a package the reference repository does not contain (it serves the offline
Faster-RCNN feature extraction, `obj_features.py`, not the model): empty stand-in modules carry those three names so that the file
imports; nothing from them is called.  Everything else is the shim of make_golden.py.

Stored: inputs (ids, mask, ROI features, boxes, ITM inputs, labels), the module's four outputs, `get_logits_per_text`, the three
loss mixes of mm_early.py:366-379 and, for the ITC + ITM mix, the gradients of a few parameters (dropout 0; matrices: first rows + norm).
Run:  python tests/golden/make_lxmert_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import install_shim  # noqa: E402
from oracle import lxmert_oracle as L  # noqa: E402
from oracle import mm_oracle as O  # noqa: E402

WATCH = ["linear_fusion.weight", "linear.bias", "linear_tim.weight", "logit_scale",
         "model.encoder.x_layers.0.visual_attention.att.query.weight", "model.encoder.x_layers.1.visn_output.dense.weight",
         "model.encoder.r_layers.0.attention.self.value.weight", "model.encoder.layer.0.intermediate.dense.bias",
         "model.encoder.visn_fc.box_fc.weight", "model.encoder.visn_fc.visn_layer_norm.weight", "model.embeddings.position_embeddings.weight"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shim()
    for name, attrs in (("lxmert_scripts", {}), ("lxmert_scripts.modeling_frcnn", {"GeneralizedRCNN": object}),
                        ("lxmert_scripts.utils", {"Config": object}), ("lxmert_scripts.processing_image", {"Preprocess": object})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    import mm_early as ref
    from transformers import LxmertConfig, LxmertModel
    c = L.LxmertConfig(l_layers=2, r_layers=1, x_layers=2, vocab=1000, max_pos=64, num_labels=3)
    hf = LxmertConfig(vocab_size=c.vocab, hidden_size=c.hidden, num_attention_heads=c.heads, intermediate_size=c.inter, l_layers=c.l_layers,
                      x_layers=c.x_layers, r_layers=c.r_layers, visual_feat_dim=c.feat_dim, visual_pos_dim=c.pos_dim,
                      max_position_embeddings=c.max_pos, type_vocab_size=c.type_vocab, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    with tempfile.TemporaryDirectory() as tmp:
        LxmertModel(hf).save_pretrained(tmp)
        model = ref.Lxmert(tmp, c.num_labels, dropout=0.0)
    P = L.make_params(c, 0)
    sd = model.state_dict()
    missing = [k for k in sd if k not in P and not k.endswith("position_ids")]
    extra = [k for k in P if k not in sd]
    assert not missing and not extra, (missing[:5], extra[:5])
    model.load_state_dict(P, strict=False)
    B, T = 4, 20
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, B, T, 7)
    np.random.seed(30)
    t_ids, t_mask, lbl = O.prepare_itm_inputs(ids, mask)
    w = torch.tensor([0.7, 1.6, 0.9])
    # (ROI features and boxes are regenerated from the seed by the tests: oracle.lxmert_oracle.synthetic_batch)
    out = dict(cfg=np.array(repr(L.asdict(c))), B=B, T=T, seed_w=0, seed_x=7, ids=ids.numpy(), mask=mask.numpy(),
               feats_sum=np.float64(feats.double().sum().item()), onehot=onehot.numpy(), tim_ids=t_ids.numpy(), tim_mask=t_mask.numpy(), lbl_tim=lbl.numpy(), class_w=w.numpy())
    model.eval()
    with torch.no_grad():
        o, et, ev, otim = model(ids, mask, tt, feats, boxes, tim_inputs=(t_ids, t_mask, torch.zeros_like(t_ids)))
        out.update(out_cls=o.numpy(), emb_t=et.numpy(), emb_v=ev.numpy(), out_tim=otim.numpy(), logits_per_text=model.get_logits_per_text(et, ev).numpy())
    model.train()                                  # dropout probabilities are 0: train mode == eval mode here, gradients flow
    loss_fn, tim_fn = torch.nn.CrossEntropyLoss(weight=w), torch.nn.CrossEntropyLoss()
    for tag, itc, itm in (("cls", False, False), ("itc", True, False), ("itcitm", True, True)):
        model.zero_grad()
        o, et, ev, otim = model(ids, mask, tt, feats, boxes, tim_inputs=(t_ids, t_mask, torch.zeros_like(t_ids)) if itm else None)
        label = onehot.type_as(o)
        if itc and itm:                            # mm_early.py:366-379
            loss = (1 - 0.2) * loss_fn(o, label) + 0.1 * ref.clip_loss(model.get_logits_per_text(et, ev)) + 0.1 * tim_fn(otim, lbl)
        elif itc:
            loss = (1 - 0.1) * loss_fn(o, label) + 0.1 * ref.clip_loss(model.get_logits_per_text(et, ev))
        else:
            loss = loss_fn(o, label)
        out["loss." + tag] = np.float64(loss.item())
        if tag == "itcitm":
            loss.backward()
            named = dict(model.named_parameters())
            for k in WATCH:
                g = named[k].grad
                # matrices: the first 4 rows and the Frobenius norm (a full 768 x 3072 gradient is 9 MB)
                out["grad." + k] = (g[:4] if g.dim() == 2 else g).numpy()
                out["gradnorm." + k] = np.float64(g.double().norm().item())
            out["no_grad"] = np.array([k for k, p in named.items() if p.grad is None])
        print(tag, "loss", loss.item())
    np.savez_compressed(os.path.join(HERE, "lxmert_small.npz"), **out)
    print("out_cls", out["out_cls"].flatten()[:4])


if __name__ == "__main__":
    main()
