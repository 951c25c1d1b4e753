#!/usr/bin/env python3
"""Parity mode (bf16x3): what the BACKWARD's product count (mmhip_set_backward_products: 3 | 2 | 1 bf16 MFMA products per reduction slice)
does to the gradients.  The forward always takes three products, so logits and loss are unchanged; this prints, per setting and model depth,
the relative L2 error of every parameter gradient against the fp32 CPU oracle (worst / median / per depth) and the cosine of the whole
flat gradient.  GPU tool: python tools/x3_bwd_policy.py [--layers 2,12] > profiles/rNN_x3_bwd_policy.txt"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import smtc_amd  # noqa: F401,E402
from smtc_amd import _lib  # noqa: E402
from smtc_amd.mm_late import MM_Model  # noqa: E402
from oracle import mm_oracle as O  # noqa: E402


def run(layers, B, T, itm=True):
    cfg = O.OracleConfig(layers_txt=layers, layers_img=2, vocab=1000, max_pos=130, num_labels=3, p_hidden=0.0, p_attn=0.0, p_head=0.0)
    arch = dict(layers_txt=layers, layers_img=2, vocab=1000, max_pos=130, p_hidden=0.0, p_attn=0.0)
    P = O.make_params(cfg, 11)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 5, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    r_cls, r_lpt, r_tim, _, _ = O.mm_forward(Pg, ids, mask, pixels, cfg, (tim_ids, tim_mask))
    ref = O.mix_loss(r_cls, onehot, None, r_lpt, r_tim, lbl, True, True)
    ref.backward()
    lib = _lib.lib()
    for npd in (3, 2, 1):
        model = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, max_posts=B, max_text_len=T, device="cuda:0", dtype="bf16x3")
        model.load_state_dict(P, strict=False)
        model.train()
        _lib.check(lib.mmhip_set_backward_products(model._handle, npd))
        dev = model.device_
        model._flat_grad.zero_()
        model._engine_forward(ids, mask, pixels, tim_ids, tim_mask, seed=1)
        lo = torch.empty(4, device=dev)
        oh, lt = onehot.to(dev).contiguous(), lbl.to(dev)
        _lib.check(lib.mmhip_loss(model._handle, _lib.ptr(oh), None, _lib.ptr(lt), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(lib.mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        errs, by_layer, dot, na, nb = {}, {}, 0.0, 0.0, 0.0
        for i in model._train_params:
            k = i["name"]
            if Pg[k].grad is None or k.endswith("key.bias") or k == "fc_K.bias":
                continue
            g = model._flat_grad[i["offset"]: i["offset"] + i["numel"]].view(i["shape"]).double().cpu()
            r = Pg[k].grad.double()
            e = (g - r).norm().item() / max(r.norm().item(), 1e-30)
            errs[k] = e
            dot += float((g * r).sum()); na += float((g * g).sum()); nb += float((r * r).sum())
            if ".layer." in k:
                li = int(k.split(".layer.")[1].split(".")[0])
                by_layer.setdefault(li, []).append(e)
        v = sorted(errs.values())
        worst = max(errs, key=errs.get)
        print(f"layers {layers:2d} B {B} T {T} | backward products {npd} | loss rel err {abs(lo[0].item() - ref.item()) / abs(ref.item()):.1e} | "
              f"gradient rel-L2: worst {v[-1]:.2e} ({worst}) median {v[len(v) // 2]:.2e} | 1 - cos(flat gradient) {1 - dot / (na * nb) ** 0.5:.2e}")
        print("    per text layer (max over the layer's tensors), layer 0 first: " +
              " ".join("%.1e" % max(by_layer[li]) for li in sorted(by_layer)))
        print("    embeddings: " + " ".join(f"{k.split('embeddings.')[1]} {e:.1e}" for k, e in errs.items() if "embeddings." in k))
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="2,12")
    a = ap.parse_args()
    print("# parity mode, backward product count vs gradient error (fp32 CPU oracle as the reference; forward = three products in every row)")
    for L in (int(x) for x in a.layers.split(",")):
        run(L, 4, 64)
