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


def drift(layers, B, T, steps, lr):
    """the same short training run on the GPU (each setting) and in the fp32 oracle (its own AdamW, reference models/mm_late.py:420-422, 489-491): how far
    loss and logits are apart after `steps` optimizer steps.  Adam's first steps are sign-like (m / sqrt(v) = g / |g|), so ANY gradient noise moves a
    weight whose gradient is near zero by up to 2 lr per step: the drift below is that, not a forward error."""
    import types
    from smtc_amd.mm_late import MMLate_Model
    cfg = O.OracleConfig(layers_txt=layers, layers_img=2, vocab=1000, max_pos=130, num_labels=3, p_hidden=0.0, p_attn=0.0, p_head=0.0)
    arch = dict(layers_txt=layers, layers_img=2, vocab=1000, max_pos=130, p_hidden=0.0, p_attn=0.0)
    P = O.make_params(cfg, 11)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 5, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    wd = 2.5e-4
    Pt = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in Pt.items()}
    ref_losses = []
    for step in range(1, steps + 1):
        for q in Pt.values():
            q.grad = None
        o = O.mm_forward(Pt, ids, mask, pixels, cfg, (tim_ids, tim_mask))
        rl = O.mix_loss(o[0], onehot, None, o[1], o[2], lbl, True, True)
        rl.backward()
        ref_losses.append(rl.item())
        with torch.no_grad():
            for k, q in Pt.items():
                if q.grad is not None:
                    O.adamw_step(q, q.grad, mom[k][0], mom[k][1], step, lr, wd)
    with torch.no_grad():
        ro = O.mm_forward({k: v.detach() for k, v in Pt.items()}, ids, mask, pixels, cfg, (tim_ids, tim_mask))
    cfgd = types.SimpleNamespace(batch_size=B, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=T, dropout=0.0)
    for tag, kw in (("bf16x3, backward products 3", dict(dtype="bf16x3", backward_products=3)), ("bf16x3, backward products 2", dict(dtype="bf16x3", backward_products=2)),
                    ("bf16x3, backward products 1", dict(dtype="bf16x3", backward_products=1)), ("f16", dict(dtype="f16")), ("bf16", dict(dtype="bf16"))):
        tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, **kw)
        tr.model.load_state_dict(P, strict=False)
        tr.model._refresh_weights(3)
        dev = tr.device
        worst_loss = 0.0
        for step in range(1, steps + 1):
            loss, _ = tr.train_step(ids.to(dev), mask.to(dev), pixels, onehot, None, lr, wd, step, tim=(tim_ids.to(dev), tim_mask.to(dev), lbl.to(dev)))
            worst_loss = max(worst_loss, abs(loss[0].item() - ref_losses[step - 1]) / abs(ref_losses[step - 1]))
        tr.model.eval()
        with torch.no_grad():
            go = tr.model(ids, mask, pixels, tim_inputs=(tim_ids, tim_mask))
        rel = lambda a, b: (a.double().cpu() - b.double()).abs().max().item() / b.double().abs().max().item()
        print(f"layers {layers:2d} lr {lr:g} steps {steps:2d} | {tag:28s} | loss trajectory, worst rel err {worst_loss:.1e} | after the last step: out_cls {rel(go[0], ro[0]):.1e} "
              f"logits_per_text {rel(go[1], ro[1]):.1e} out_tim {rel(go[2], ro[2]):.1e} mm_features {rel(go[4], ro[4]):.1e}")
        del tr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="2,12")
    a = ap.parse_args()
    print("# parity mode, backward product count vs gradient error (fp32 CPU oracle as the reference; forward = three products in every row)")
    for L in (int(x) for x in a.layers.split(",")):
        run(L, 4, 64)
    print("\n# drift of a short training run against the fp32 oracle trained by its own AdamW (same batch every step, dropout off, ITC + ITM)")
    for lr, steps in ((1e-5, 4), (1e-5, 16), (1e-4, 4)):
        drift(4, 4, 64, steps, lr)
