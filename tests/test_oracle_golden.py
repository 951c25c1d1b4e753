"""CPU: the oracle (oracle/mm_oracle.py) against vectors produced by the reference's own code
(tests/golden/make_golden.py).  Tolerances are fp32 round-off of two implementations of one algorithm."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import mm_oracle as O



def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name), allow_pickle=False)
    cfg = O.OracleConfig(**ast.literal_eval(str(z["cfg"]))) if "cfg" in z.files else None
    return z, cfg


def t(z, k):
    return torch.from_numpy(z[k])


def pixels_of(cfg, z):
    return O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), bool(z["pad"]) if "pad" in z.files else True)[2]


FWD_GOLDENS = ["fwd_small_xlmr", "fwd_small_bert", "fwd_small_concat", "fwd_full_xlmr",
               # round 4 (make_golden.py --extra): other seeds, batch sizes, lengths, padding on / off, 2 / 6 / 12 layers
               "fwd_x_full_xlmr_a", "fwd_x_full_xlmr_b", "fwd_x_full_xlmr_c", "fwd_x_full_bert_a", "fwd_x_full_bert_b", "fwd_x_full_concat",
               "fwd_x_mid_xlmr", "fwd_x_small_xlmr", "fwd_x_small_bert"]


@pytest.mark.parametrize("name", FWD_GOLDENS)
def test_forward_matches_reference(golden_dir, name):
    z, cfg = load(golden_dir, name + ".npz")
    P = O.make_params(cfg, int(z["seed_w"]))
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), bool(z["pad"]))
    assert torch.equal(ids, t(z, "ids")) and torch.equal(mask, t(z, "mask")) and torch.equal(onehot, t(z, "onehot"))
    col = {}
    with torch.no_grad():
        out_cls, lpt, out_tim, _, feats = O.mm_forward(P, ids, mask, pixels, cfg, (t(z, "tim_ids"), t(z, "tim_mask")), collect=col)
    for got, key in ((out_cls, "out_cls"), (lpt, "logits_per_text"), (out_tim, "out_tim"), (feats, "mm_features")):
        ref = t(z, key)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (key, err)
    if "vit_cls_per_layer" in z.files:
        v = torch.stack([h[:, 0] for h in col["vit_layers"]])
        x = torch.stack([h[:, 0] for h in col["txt_layers"]])
        assert (v - t(z, "vit_cls_per_layer")).abs().max() < 2e-4
        assert (x - t(z, "txt_cls_per_layer")).abs().max() < 2e-5


@pytest.mark.parametrize("name", ["clip_small_224", "clip_small_336"])
def test_clip_vision_tower_matches_hf(golden_dir, name):
    """BASELINE config 4's image tower: the oracle's CLIPVisionModel restatement against vectors produced by HuggingFace's own
    VisionTextDualEncoderModel + CLIPVisionModel (tests/golden/make_clip_golden.py); 257 tokens at 224, 577 at 336"""
    z, cfg = load(golden_dir, name + ".npz")
    P = O.make_params(cfg, int(z["seed_w"]))
    ids, mask, pixels, _ = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), True)
    assert torch.equal(ids, t(z, "ids")) and torch.equal(mask, t(z, "mask"))
    with torch.no_grad():
        x_v, v_pool = O.clip_vision_forward(P, pixels, cfg)
        _, t_pool = O.text_forward(P, ids, mask, cfg)
        lpt = O.itc_logits(P, t_pool, v_pool)
    assert x_v.shape[1] == (cfg.image // 14) ** 2 + 1
    for got, key in ((x_v[:, 0], "v_cls"), (x_v[:, 7], "v_tok7"), (x_v[:, -1], "v_last"), (v_pool, "v_pool"), (t_pool, "t_pool"), (lpt, "logits_per_text")):
        ref = t(z, key)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (key, err)


def test_train_losses_and_grads_match_reference(golden_dir):
    z, cfg = load(golden_dir, "train_small_xlmr.npz")
    P = {k: v.requires_grad_(O.trainable(k)) for k, v in O.make_params(cfg, int(z["seed_w"])).items()}
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), True)
    w = t(z, "class_weight")
    watch = [str(s) for s in z["watch"]]
    for mix, (itc, itm) in {"plain": (False, False), "itc": (True, False), "itm": (False, True), "itcitm": (True, True)}.items():
        for p in P.values():
            p.grad = None
        tim = (t(z, "tim_ids"), t(z, "tim_mask")) if itm else None
        out_cls, lpt, out_tim, _, _ = O.mm_forward(P, ids, mask, pixels, cfg, tim)
        loss = O.mix_loss(out_cls, onehot, w, lpt, out_tim, t(z, "lbl_tim"), itc, itm)
        loss.backward()
        assert abs(loss.item() - float(z[f"{mix}.loss"])) < 2e-6 * abs(float(z[f"{mix}.loss"])) + 1e-6
        none_ref = {str(s) for s in z[f"{mix}.grad_none"]}
        none_got = {k for k, p in P.items() if p.requires_grad and p.grad is None}
        # tensors that exist in the reference module but are not on this path never get gradients there either
        assert none_got == none_ref, (mix, none_got ^ none_ref)
        for k in watch:
            key = f"{mix}.gnorm.{k}"
            if key not in z.files:
                assert P[k].grad is None, (mix, k)
                continue
            g = P[k].grad
            if k.endswith("key.bias"):      # mathematically zero (softmax shift invariance): round-off only
                assert g.norm().item() < 1e-7 and float(z[key]) < 1e-7
                continue
            assert abs(g.norm().item() - float(z[key])) <= 2e-4 * float(z[key]) + 1e-9, (mix, k)
            ref = t(z, f"{mix}.gslice.{k}")
            if k.endswith("word_embeddings.weight"):
                got = g[t(z, f"{mix}.gslice_rows.{k}")][:, :48]
                assert g[cfg.pad_id].norm().item() == float(z[f"{mix}.gpad_row_norm"]) == 0.0
            elif g.dim() == 2:
                got = g[:8, :48]
            else:
                got = g.flatten()[:64]
            assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-9, (mix, k)


def test_itm_sampling_matches_reference(golden_dir):
    z, _ = load(golden_dir, "itm_sampling.npz")
    for B in (1, 2, 8, 64):
        np.random.seed(30)
        a, b, c = O.prepare_itm_inputs(t(z, f"B{B}.ids"), t(z, f"B{B}.mask"))
        assert torch.equal(a, t(z, f"B{B}.tim_ids")) and torch.equal(b, t(z, f"B{B}.tim_mask")) and torch.equal(c, t(z, f"B{B}.lbl"))
    assert z["B8.lbl"].tolist() == [1, 1, 1, 1, 0, 1, 0, 1]          # SURVEY.md §8c known answer (5)


def test_adamw_matches_torch(golden_dir):
    z, _ = load(golden_dir, "adamw.npz")
    p = t(z, "p0").clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(3):
        O.adamw_step(p, t(z, "grads")[i], m, v, i + 1, float(z["lr"]), float(z["wd"]))
        assert (p - t(z, "traj")[i]).abs().max().item() < 1e-8


def test_losses_match_reference(golden_dir):
    z, _ = load(golden_dir, "losses.npz")
    assert abs(O.cls_loss(t(z, "out"), t(z, "onehot"), t(z, "w")).item() - float(z["l_cls"])) < 1e-6
    assert abs(O.clip_loss(t(z, "sim")).item() - float(z["l_clip"])) < 1e-6


def test_hash_dropout_rate():
    keep = O.hash_keep_mask(1 << 20, 12345, O.stream_attn(3), 0x1234_5678_9ABC, 0.1)
    assert abs(keep.mean() - 0.9) < 2e-3
    k2 = O.hash_keep_mask(1 << 20, 12345, O.stream_attn(4), 0x1234_5678_9ABC, 0.1)
    assert abs((keep & k2).mean() - 0.81) < 3e-3


def eval_loader(cfg, z):
    """the loader of tests/golden/make_eval_golden.py (items shaped like MM_Dataset's)"""
    out, first = [], 5000
    for B, seed in z["batches"].tolist():
        ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, int(z["T"]), seed, True)
        out.append({"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "pixel_values": pixels.unsqueeze(1),
                    "labels": onehot, "data_id": torch.arange(first, first + B)})
        first += B
    return out


@pytest.mark.parametrize("tag,itc,itm", [("plain", False, False), ("itcitm", True, True)])
def test_eval_loop_matches_reference(golden_dir, tag, itc, itm):
    """reference MMLate_Model.eval (models/mm_late.py:534-638) restated over the oracle: eval-mode forward, per-batch loss mix
    with re-sampled ITM negatives, argmax predictions / labels, mean of the per-batch losses"""
    z, cfg = load(golden_dir, "eval_small_xlmr.npz")
    P = O.make_params(cfg, int(z["seed_w"]))
    w = t(z, "class_w")
    np.random.seed(30)
    losses, preds, labels, ids_all, logits = [], [], [], [], []
    with torch.no_grad():
        for b in eval_loader(cfg, z):
            ids, mask, px = torch.squeeze(b["input_ids"]), torch.squeeze(b["attention_mask"]), torch.squeeze(b["pixel_values"])
            tim = O.prepare_itm_inputs(ids, mask) if itm else None
            out_cls, lpt, out_tim, _, _ = O.mm_forward(P, ids, mask, px, cfg, tim[:2] if tim else None)
            losses.append(O.mix_loss(out_cls, b["labels"], w, lpt, out_tim, tim[2] if tim else None, itc, itm).item())
            preds.append(out_cls.argmax(1)); labels.append(b["labels"].argmax(1)); ids_all.append(b["data_id"]); logits.append(out_cls)
    assert abs(np.mean(losses) - float(z[tag + ".loss"])) < 2e-5 * abs(float(z[tag + ".loss"]))
    assert torch.equal(torch.cat(preds), t(z, tag + ".predictions")) and torch.equal(torch.cat(labels), t(z, tag + ".labels"))
    assert torch.equal(torch.cat(ids_all), t(z, tag + ".data_id"))
    assert (torch.cat(logits) - t(z, "out_cls")).abs().max() < 2e-5 * t(z, "out_cls").abs().max()


def test_lxmert_oracle_matches_reference(golden_dir):
    """groundwork for BASELINE config 5 (early-fusion LXMERT; the HIP path for it does not exist yet): oracle/lxmert_oracle.py
    against the reference's own mm_early.Lxmert module (tests/golden/make_lxmert_golden.py) -- four outputs, ITC logits, three loss
    mixes, gradients of the ITC + ITM mix (one cross-attention module used in both directions, detached text embedding)"""
    import ast as _ast
    from oracle import lxmert_oracle as L
    z = np.load(os.path.join(golden_dir, "lxmert_small.npz"), allow_pickle=False)
    c = L.LxmertConfig(**_ast.literal_eval(str(z["cfg"])))
    P = L.make_params(c, int(z["seed_w"]))
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, int(z["B"]), int(z["T"]), int(z["seed_x"]))
    assert torch.equal(ids, t(z, "ids")) and torch.equal(mask, t(z, "mask")) and abs(feats.double().sum().item() - float(z["feats_sum"])) < 1e-6
    tim = (t(z, "tim_ids"), t(z, "tim_mask"), torch.zeros_like(ids))
    rel = lambda a, b: (a - b).abs().max().item() / b.abs().max().item()
    with torch.no_grad():
        out, et, ev, otim = L.early_forward(P, ids, mask, tt, feats, boxes, c, tim)
        for got, key in ((out, "out_cls"), (et, "emb_t"), (ev, "emb_v"), (otim, "out_tim"), (L.logits_per_text(P, et, ev), "logits_per_text")):
            assert rel(got, t(z, key)) < 2e-5, (key, rel(got, t(z, key)))
    w, lbl = t(z, "class_w"), t(z, "lbl_tim")
    for tag, itc, itm in (("cls", False, False), ("itc", True, False), ("itcitm", True, True)):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        out, et, ev, otim = L.early_forward(Pg, ids, mask, tt, feats, boxes, c, tim if itm else None)
        loss = L.mix_loss(Pg, out, onehot, w, et, ev, otim, lbl, itc, itm)
        assert abs(loss.item() - float(z["loss." + tag])) < 2e-5 * abs(float(z["loss." + tag])), tag
        if tag == "itcitm":
            loss.backward()
            for k in [f[5:] for f in z.files if f.startswith("grad.")]:
                g, ref = Pg[k].grad, t(z, "grad." + k)
                got = g[:4] if g.dim() == 2 else g
                assert rel(got, ref) < 2e-4, (k, rel(got, ref))
                assert abs(g.double().norm().item() - float(z["gradnorm." + k])) < 2e-4 * float(z["gradnorm." + k]), k
            # the pooler is not on the path (mm_early.py:132 takes the CLS row itself): no gradient, in the reference either
            assert sorted(str(k) for k in z["no_grad"]) == sorted(k for k, v in Pg.items() if v.grad is None) == ["model.pooler.dense.bias", "model.pooler.dense.weight"]
            # padding_idx=0 on all three embedding tables: position 0 and token type 0 receive no gradient
            assert not Pg["model.embeddings.position_embeddings.weight"].grad[0].any() and not Pg["model.embeddings.token_type_embeddings.weight"].grad[0].any()


def test_rounding_emulation_policies(golden_dir):
    """oracle `rounding`: policy None is the pinned fp32 oracle bit for bit; the 16-bit policies deviate from the reference golden by what
    twelve layers of operand / storage rounding cost (the magnitudes the HIP modes measure on MI355X); hi+lo bf16 pairs on
    both operands (the bf16x3 parity mode's products) stay at fp32 level."""
    z, cfg = load(golden_dir, "fwd_small_xlmr.npz")
    P = O.make_params(cfg, int(z["seed_w"]))
    ids, mask, pixels, _ = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), bool(z["pad"]))
    tim = (t(z, "tim_ids"), t(z, "tim_mask"))

    def run(**kw):
        with torch.no_grad(), O.rounding(**kw):
            o = O.mm_forward(P, ids, mask, pixels, cfg, tim)
        return {"out_cls": o[0], "logits_per_text": o[1], "out_tim": o[2], "mm_features": o[4]}

    with torch.no_grad():
        base = O.mm_forward(P, ids, mask, pixels, cfg, tim)
    none = run(round_operands=None)
    assert all(torch.equal(none[k], b) for k, b in zip(("out_cls", "logits_per_text", "out_tim"), base[:3]))
    err = lambda o: max((o[k] - t(z, k)).abs().max().item() / t(z, k).abs().max().item() for k in o)
    e_bf16, e_f16 = err(run(round_operands="bf16")), err(run(round_operands="f16"))
    e_x3 = err(run(round_operands=None, op_a="bf16x2", op_w="bf16x2"))
    assert 1e-3 < e_bf16 < 6e-2 and 1e-4 < e_f16 < 8e-3 and e_f16 < e_bf16, (e_bf16, e_f16)
    assert e_x3 < 1e-4, e_x3


def test_rounding_emulation_gradients_are_finite_and_close(golden_dir):
    """backward through the rounding policy (straight-through rounding of stored activations, gradient rounding, gelu' of the stored
    pre-activation): gradients stay within the 16-bit band of the fp32 reference gradients"""
    z, cfg = load(golden_dir, "train_small_xlmr.npz")
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, int(z["B"]), int(z["T"]), int(z["seed_x"]), True)
    w = t(z, "class_weight")
    for dtype, band in (("bf16", 0.6), ("f16", 0.05)):
        P = {k: v.requires_grad_(O.trainable(k)) for k, v in O.make_params(cfg, int(z["seed_w"])).items()}
        with O.rounding(dtype):
            out_cls, lpt, out_tim, _, _ = O.mm_forward(P, ids, mask, pixels, cfg, None)
            O.mix_loss(out_cls, onehot, w, lpt, out_tim, None, False, False).backward()
        for k in (str(s) for s in z["watch"]):
            key = f"plain.gnorm.{k}"
            if key not in z.files or k.endswith("key.bias") or P[k].grad is None:
                continue
            g = P[k].grad
            assert torch.isfinite(g).all(), k
            assert abs(g.norm().item() - float(z[key])) / float(z[key]) < band, (dtype, k)
