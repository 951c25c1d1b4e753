"""-m gpu: the whole HIP path (MM_Model drop-in -> C ABI -> engine) against
  (1) golden vectors produced by the reference's own MM_Model (tests/golden/*.npz),
  (2) the CPU oracle on the same seeded inputs, including train mode with the kernels' own dropout masks replayed,
  (3) size-independent properties at the full BASELINE size (B=64, T=128, 12+12 layers).

Error metric: max|got - ref| / max|ref| per output tensor.  Tolerances: operands and stored activations
are rounded to the 16-bit type; twelve post-LN layers amplify that to ~1e-2 (bf16, 8 mantissa bits) / ~1e-3 (f16, 11 bits)
on random-init logits (SURVEY.md 7.3 measured 7e-3 / 9e-4 for operand rounding alone); the loss is an average and sits
well inside 1e-3 for both."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mm_oracle as O
from smtc_amd.synthetic import synthetic_batch

if torch.cuda.is_available():
    import smtc_amd  # noqa: F401
    from smtc_amd.mm_late import MM_Model, MMLate_Model
    from gpu_util import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# Tolerances.  "bf16x3" is the strict-parity mode (fp32 activations, every Linear as three bf16 MFMA products of hi/lo-split
# operands, fp32 attention): north_star's 1e-3 on the per-post outputs is asserted in it (measured on MI355X: <= 1.8e-5 on all
# four forward goldens, loss 4e-7, gradients <= 2.8e-5).  The 16-bit modes are bounded at about twice what they measure
# (round 2, max over the goldens):   bf16  out_cls 8.1e-3  logits_per_text 1.9e-2  out_tim 1.5e-2  mm_features 1.15e-2
#                                     f16   out_cls 1.6e-3  logits_per_text 2.5e-3  out_tim 1.4e-3  mm_features 1.4e-3
# i.e. neither 16-bit mode meets 1e-3 on the logits (12 post-LN layers amplify operand rounding); both meet it on the loss.
# Round 4: thirteen goldens instead of four (nine more seeds / shapes, six of them full depth); worst measured over all of them:
#                                     bf16  out_cls 3.65e-2  logits_per_text 3.3e-2  out_tim 2.85e-2  mm_features 1.6e-2
TOL_OUT = {"bf16": {"out_cls": 7e-2, "logits_per_text": 7e-2, "out_tim": 6e-2, "mm_features": 3.2e-2},
           "f16": {"out_cls": 1e-2, "logits_per_text": 2e-2, "out_tim": 1e-2, "mm_features": 6e-3},
           "bf16x3": {"out_cls": 1e-3, "logits_per_text": 1e-3, "out_tim": 1e-3, "mm_features": 1e-3}}
TOL_LOSS = {"bf16": 1e-3, "f16": 1e-3, "bf16x3": 1e-4}
# gradients: relative L2 error of the compared slice / tensor.  The LOGIC of the backward is pinned in bf16x3 at 1e-3 (measured
# 2.8e-5).  In the 16-bit modes a tensor's error is rounding noise that depends on how much cancellation its gradient has:
# measured worst tensor 0.254 (bf16, linear_fusion.bias) / 0.0126 (f16), medians 0.09 / 0.0025 -- bounded at twice the worst
# per tensor and, against systematic regressions, at TOL_GRAD_MEDIAN over the watched tensors.
TOL_GRAD = {"bf16": 0.5, "f16": 0.03, "bf16x3": 1e-3}
TOL_GRAD_MEDIAN = {"bf16": 0.12, "f16": 6e-3, "bf16x3": 1e-4}


def load(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    return z, O.OracleConfig(**ast.literal_eval(str(z["cfg"])))


def t(z, k):
    return torch.from_numpy(z[k])


def build(cfg, dtype, txt="bernice", B=8, T=128):
    arch = dict(layers_txt=cfg.layers_txt, layers_img=cfg.layers_img, vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                p_hidden=cfg.p_hidden, p_attn=cfg.p_attn)
    return MM_Model(cfg.num_labels, txt, "vit", cfg.p_head, cfg.fusion, arch=arch, dtype=dtype, max_posts=B, max_text_len=T)


def load_oracle_params(model, P):
    sd = model.state_dict()
    assert set(P) | {"dual_encoder.text_model.embeddings.position_ids"} == set(sd), set(P) ^ set(sd)
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert unexpected == [] and missing == ["dual_encoder.text_model.embeddings.position_ids"]


# round 4: nine more reference-generated forward cases (tests/golden/make_golden.py --extra): other seeds, batch sizes, lengths, padding
# on / off, BERT and XLM-R, concat, 2 / 6 / 12 layers -- a numerics policy is judged on thirteen goldens
FWD_GOLDENS = ["fwd_small_xlmr", "fwd_small_bert", "fwd_small_concat", "fwd_full_xlmr",
               "fwd_x_full_xlmr_a", "fwd_x_full_xlmr_b", "fwd_x_full_xlmr_c", "fwd_x_full_bert_a", "fwd_x_full_bert_b", "fwd_x_full_concat",
               "fwd_x_mid_xlmr", "fwd_x_small_xlmr", "fwd_x_small_bert"]


@pytest.mark.parametrize("dtype", ["bf16", "f16", "bf16x3"])
@pytest.mark.parametrize("name", FWD_GOLDENS)
def test_forward_matches_reference_golden(name, dtype):
    z, cfg = load(name + ".npz")
    txt = "bert" if cfg.txt_kind == "bert" else "bernice"
    B, T = int(z["B"]), int(z["T"])
    model = build(cfg, dtype, txt, B, T)
    load_oracle_params(model, O.make_params(cfg, int(z["seed_w"])))
    model.eval()
    pixels = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), bool(z["pad"]))[2]
    with torch.no_grad():
        out_cls, lpt, out_tim, none, feats = model(t(z, "ids"), t(z, "mask"), pixels, tim_inputs=(t(z, "tim_ids"), t(z, "tim_mask")))
    assert none is None
    errs = {k: rel_err(v, t(z, k)) for k, v in (("out_cls", out_cls), ("logits_per_text", lpt), ("out_tim", out_tim), ("mm_features", feats))}
    print(name, dtype, errs)
    for k, e in errs.items():
        assert e < TOL_OUT[dtype][k], (k, e)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "bf16x3"])
@pytest.mark.parametrize("path", ["fused", "autograd"])
def test_train_losses_and_grads_match_reference_golden(dtype, path):
    """train() mode, dropout p = 0: the three loss mixes and gradients of the watched parameters"""
    z, cfg = load("train_small_xlmr.npz")
    B, T = int(z["B"]), int(z["T"])
    model = build(cfg, dtype, "bernice", B, T)
    load_oracle_params(model, O.make_params(cfg, int(z["seed_w"])))
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), True)
    w = t(z, "class_weight")
    dev = model.device_
    named = dict(model.named_parameters())
    for mix, (itc, itm) in {"plain": (False, False), "itc": (True, False), "itm": (False, True), "itcitm": (True, True)}.items():
        tim = (t(z, "tim_ids"), t(z, "tim_mask")) if itm else None
        bi, bm = (0.1 if itc else 0.0), (0.1 if itm else 0.0)
        if path == "autograd":
            for p in model.parameters():
                p.grad = None
            out_cls, lpt, out_tim, _, _ = model(ids, mask, pixels, tim_inputs=tim)
            loss = O.mix_loss(out_cls, onehot.to(dev), w.to(dev), lpt, out_tim, t(z, "lbl_tim").to(dev), itc, itm)
            loss.backward()
            grads = {k: p.grad for k, p in named.items()}
            none_got = {k for k, p in named.items() if p.requires_grad and p.grad is None}
            assert none_got == {str(s) for s in z[f"{mix}.grad_none"]}, mix
            loss_v = loss.item()
        else:
            import ctypes as C
            from smtc_amd import _lib
            model._flat_grad.zero_()
            ti, tm_ = tim if tim else (None, None)
            model._engine_forward(ids, mask, pixels, ti, tm_)
            lo = torch.empty(4, device=dev)
            oh, cw, lt = onehot.to(dev).contiguous(), w.to(dev), t(z, "lbl_tim").to(dev)
            _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), _lib.ptr(cw), _lib.ptr(lt) if itm else None, 1.0 - bi - bm, bi, bm,
                                             _lib.ptr(lo), None, _lib.stream_ptr()))
            _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
            grads = {i["name"]: model._flat_grad[i["offset"]: i["offset"] + i["numel"]].view(i["shape"]) for i in model._train_params}
            loss_v = lo[0].item()
        ref_loss = float(z[f"{mix}.loss"])
        assert abs(loss_v - ref_loss) < TOL_LOSS[dtype] * abs(ref_loss), (mix, loss_v, ref_loss)
        worst = {}
        for k in (str(s) for s in z["watch"]):
            key = f"{mix}.gnorm.{k}"
            if key not in z.files or k.endswith("key.bias"):
                continue
            g = grads[k].detach().float().cpu()
            ref = t(z, f"{mix}.gslice.{k}")
            if k.endswith("word_embeddings.weight"):
                got = g[t(z, f"{mix}.gslice_rows.{k}")][:, :48]
                assert g[cfg.pad_id].abs().max().item() == 0.0
            elif g.dim() == 2:
                got = g[:8, :48]
            else:
                got = g.flatten()[:64]
            worst[k] = max((got - ref).norm().item() / max(ref.norm().item(), 1e-20),
                           abs(g.norm().item() - float(z[key])) / float(z[key]))
        print(mix, dtype, path, "loss", loss_v, ref_loss, "worst grad", max(worst.items(), key=lambda kv: kv[1]))
        print("GRADERR", dtype, mix, path, {k: float("%.3g" % e) for k, e in worst.items()})
        for k, e in worst.items():
            assert e < TOL_GRAD[dtype], (mix, k, e)
        assert float(np.median(list(worst.values()))) < TOL_GRAD_MEDIAN[dtype], (mix, sorted(worst.values()))


# Parity mode, backward product count (include/mmhip.h mmhip_set_backward_products).  The forward always takes three bf16 MFMA
# products per reduction slice, so logits and loss keep bf16x3's tolerance whatever the backward does; with 2 / 1 products the gradients carry
# the bf16 rounding of one / both operands of the backward's matrix products AT THE MATRIX CORES (stores stay plane pairs / fp32).
# Measured (tools/x3_bwd_policy.py, profiles/r05_x3_bwd_policy.txt), relative L2 per tensor against the fp32 oracle, worst tensor:
#   2 text layers   3: 4.4e-5   2: 2.8e-3   1: 4.2e-3        12 text layers   3: 4.2e-5   2: 4.4e-3   1: 6.1e-3   (1 - cos of the flat gradient: 7e-6)
TOL_GRAD_BWD_PRODUCTS = {3: 1e-3, 2: 8e-3, 1: 1.2e-2}


@pytest.mark.parametrize("products", [3, 2, 1])
def test_backward_product_policy_keeps_outputs_and_bounds_gradients(products):
    """(a) one step: outputs and loss within bf16x3's tolerance whatever the backward does (the forward always takes three products), every gradient
    tensor within the stated bound, the flat gradient parallel to the oracle's to 1e-4.  (b) a short training run against the fp32 oracle trained by
    its own AdamW: with THREE products the trajectory itself keeps north_star's 1e-3 (loss of every step, logits after the last); with two or one
    it does not -- Adam's early steps are sign-like (m / sqrt(v) = g / |g|), gradient noise of 4e-3 flips weights whose gradient is near zero by
    2 lr per step, and after 16 steps at the reference's lr the logits are 1e-2 off, the level of the 16-bit modes after 4
    (profiles/r05_x3_bwd_policy.txt): the cheaper backward keeps the tolerance per step on given weights, not along a training run.  Printed, not
    asserted, for products < 3."""
    from smtc_amd import _lib
    cfg = O.OracleConfig(layers_txt=4, layers_img=2, vocab=800, max_pos=130, num_labels=3, p_hidden=0.0, p_attn=0.0, p_head=0.0)
    B, T = 4, 64
    arch = dict(layers_txt=cfg.layers_txt, layers_img=cfg.layers_img, vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab, p_hidden=0.0, p_attn=0.0)
    model = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, dtype="bf16x3", max_posts=B, max_text_len=T, backward_products=products)
    with pytest.raises(ValueError):
        MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, dtype="bf16", max_posts=B, max_text_len=T, backward_products=1)
    P = O.make_params(cfg, 21)
    load_oracle_params(model, P)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 9, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    dev = model.device_
    model._flat_grad.zero_()
    out = model._engine_forward(ids, mask, pixels, tim_ids, tim_mask, seed=3)
    lo = torch.empty(4, device=dev)
    oh, lt = onehot.to(dev).contiguous(), lbl.to(dev)
    _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, _lib.ptr(lt), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
    _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
    Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    r_cls, r_lpt, r_tim, _, _ = O.mm_forward(Pg, ids, mask, pixels, cfg, (tim_ids, tim_mask))
    ref = O.mix_loss(r_cls, onehot, None, r_lpt, r_tim, lbl, True, True)
    ref.backward()
    assert abs(lo[0].item() - ref.item()) < TOL_LOSS["bf16x3"] * abs(ref.item())
    assert rel_err(out[0], r_cls.detach()) < 1e-3
    errs, dot, na, nb = {}, 0.0, 0.0, 0.0
    for i in model._train_params:
        k = i["name"]
        if Pg[k].grad is None or k.endswith("key.bias") or k == "fc_K.bias":
            continue
        g = model._flat_grad[i["offset"]: i["offset"] + i["numel"]].view(i["shape"]).double().cpu()
        r = Pg[k].grad.double()
        errs[k] = (g - r).norm().item() / max(r.norm().item(), 1e-30)
        dot += float((g * r).sum()); na += float((g * g).sum()); nb += float((r * r).sum())
    worst = max(errs, key=errs.get)
    print("BWD_PRODUCTS", products, "worst", worst, "%.3g" % errs[worst], "median %.3g" % float(np.median(list(errs.values()))), "1-cos %.3g" % (1 - dot / (na * nb) ** 0.5))
    assert errs[worst] < TOL_GRAD_BWD_PRODUCTS[products], (worst, errs[worst])
    if products < 3:
        assert errs[worst] > 1e-4, "the policy did not take effect (gradients as exact as with three products)"
    assert 1 - dot / (na * nb) ** 0.5 < 1e-4
    # ---- (b) a short training run on both sides
    import types
    cfgd = types.SimpleNamespace(batch_size=B, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=T, dropout=0.0)
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, dtype="bf16x3", backward_products=products)
    load_oracle_params(tr.model, P)
    tr.model._refresh_weights(3)
    Pt = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in Pt.items()}
    lr, wd, worst_loss = 1e-4, 2.5e-4, 0.0
    for step in range(1, 5):
        loss, _ = tr.train_step(ids.to(dev), mask.to(dev), pixels, onehot, None, lr, wd, step, tim=(tim_ids.to(dev), tim_mask.to(dev), lbl.to(dev)))
        for q in Pt.values():
            q.grad = None
        o_cls, o_lpt, o_tim, _, _ = O.mm_forward(Pt, ids, mask, pixels, cfg, (tim_ids, tim_mask))
        rl = O.mix_loss(o_cls, onehot, None, o_lpt, o_tim, lbl, True, True)
        rl.backward()
        worst_loss = max(worst_loss, abs(loss[0].item() - rl.item()) / abs(rl.item()))
        with torch.no_grad():
            for k, q in Pt.items():
                if q.grad is not None:                      # torch.optim.AdamW skips `grad is None` tensors
                    O.adamw_step(q, q.grad, mom[k][0], mom[k][1], step, lr, wd)
    tr.model.eval()
    with torch.no_grad():
        g_cls, g_lpt, g_tim, _, g_feats = tr.model(ids, mask, pixels, tim_inputs=(tim_ids, tim_mask))
        o_cls, o_lpt, o_tim, _, o_feats = O.mm_forward({k: v.detach() for k, v in Pt.items()}, ids, mask, pixels, cfg, (tim_ids, tim_mask))
    after = {k: rel_err(a, b) for k, a, b in (("out_cls", g_cls, o_cls), ("logits_per_text", g_lpt, o_lpt), ("out_tim", g_tim, o_tim), ("mm_features", g_feats, o_feats))}
    print("BWD_PRODUCTS", products, "4 steps at lr 1e-4: loss trajectory worst rel err %.2g, after the last step" % worst_loss, after)
    assert np.isfinite(worst_loss) and all(np.isfinite(v) for v in after.values())
    if products == 3:
        assert worst_loss < 1e-3 and all(e < 1e-3 for e in after.values()), (worst_loss, after)


# per-tensor bounds against the ROUNDING-EMULATING oracle (oracle/mm_oracle.py `rounding`): an fp32 execution that rounds both operands of
# every tower matrix product, every stored activation and every stored gradient to the dtype where the HIP path does.  What is left
# between the two is the noise of single rounding decisions (fp32 summation order flips an operand by one 16-bit ulp), not twelve layers
# of operand rounding -- so the 16-bit backward is pinned per tensor an order of magnitude tighter than against fp32 (TOL_GRAD).
# measured on MI355X (round 3), dropout off, the emulation's fusion-head ReLU passing the units that passed on the GPU:
#   bf16  outputs <= 5.0e-3   gradients <= 7.2e-3 (plain), <= 2.1e-2 (ITC head tensors), median 6e-3
#   f16   outputs <= 6.3e-4   gradients <= 8.9e-4 (plain), <= 2.6e-3 (ITC head tensors), median 8e-4
TOL_EMU_OUT = {"bf16": 1e-2, "f16": 1.3e-3}
TOL_EMU_GRAD = {"bf16": 4e-2, "f16": 5e-3}
TOL_EMU_GRAD_MEDIAN = {"bf16": 1.5e-2, "f16": 2e-3}


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_modes_track_the_rounding_emulating_oracle(dtype):
    """forward outputs against the emulation; then the BACKWARD alone: both sides start from the same upstream gradients (those of the
    emulated loss), so that a per-tensor difference is the backward's own rounding noise and not the loss gradient p - y amplifying a
    half-percent difference of the logits"""
    z, cfg = load("train_small_xlmr.npz")
    B, T = int(z["B"]), int(z["T"])
    model = build(cfg, dtype, "bernice", B, T)
    P0 = O.make_params(cfg, int(z["seed_w"]))
    load_oracle_params(model, P0)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), True)
    w = t(z, "class_weight")
    dev = model.device_
    named = dict(model.named_parameters())
    for mix, (itc, itm) in {"plain": (False, False), "itc": (True, False)}.items():
        tim = None
        for p_ in model.parameters():
            p_.grad = None
        out_cls, lpt, out_tim, _, feats = model(ids, mask, pixels, tim_inputs=tim)
        # the emulation: same parameters, same batch, the dtype's rounding policy (f16: gradients carried x 1024, as engine.hip gscale());
        # the fusion head's ReLU passes the units that passed on the GPU (oracle `_relu`)
        P = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P0.items()}
        with O.rounding(dtype):
            r_cls, r_lpt, r_tim, _, r_feats = O.mm_forward(P, ids, mask, pixels, cfg, tim, relu_mask=(feats.detach().cpu() > 0).float())
            ref = O.mix_loss(r_cls, onehot, w, r_lpt, r_tim, t(z, "lbl_tim"), itc, itm)
            heads = [r_cls] + ([r_lpt] if itc else []) + ([r_tim] if itm else [])
            ups = torch.autograd.grad(ref, heads, retain_graph=True)
            ref.backward()
        got_out = {"out_cls": out_cls, "logits_per_text": lpt, "mm_features": feats}
        ref_out = {"out_cls": r_cls, "logits_per_text": r_lpt, "mm_features": r_feats}
        if itm:
            got_out["out_tim"], ref_out["out_tim"] = out_tim, r_tim
        oerr = {k: rel_err(got_out[k].detach(), ref_out[k].detach()) for k in got_out}
        print("EMU_OUT", dtype, mix, {k: float("%.3g" % e) for k, e in oerr.items()})
        hip_heads = [out_cls] + ([lpt] if itc else []) + ([out_tim] if itm else [])
        torch.autograd.backward(hip_heads, [u.to(dev) for u in ups])
        gerr = {}
        for k in (str(s_) for s_ in z["watch"]):
            if P[k].grad is None or k.endswith("key.bias") or named[k].grad is None:
                continue
            r = P[k].grad
            if r.norm().item() == 0.0:
                continue
            gerr[k] = (named[k].grad.detach().float().cpu() - r).norm().item() / r.norm().item()
        print("EMU_GRAD", dtype, mix, {k.replace("dual_encoder.text_model.", ""): float("%.3g" % e) for k, e in gerr.items()})
        for k, e in oerr.items():
            assert e < TOL_EMU_OUT[dtype], (mix, k, e)
        for k, e in gerr.items():
            assert e < TOL_EMU_GRAD[dtype], (mix, k, e)
        assert float(np.median(list(gerr.values()))) < TOL_EMU_GRAD_MEDIAN[dtype], (mix, sorted(gerr.values()))


# the same comparison, forward only, on ALL thirteen reference goldens (round 5; VERDICT r4 asked for it so that the 16-bit kernels' logic is pinned
# tighter than the 7e-2 allowance against fp32 on the nine newer goldens).  What it shows (MI355X, round 5, worst over the outputs):
#   2 text layers (4 goldens)       bf16 4e-3 .. 1.9e-2    f16 5e-4 .. 2.8e-3      -- three to ten times tighter than against fp32
#   6 / 12 text layers (9 goldens)  bf16 1.3e-2 .. 5.1e-2  f16 1.2e-3 .. 6.8e-3    -- NOT tighter than against fp32 (3.65e-2 / 8.8e-3)
# i.e. with depth the emulation stops tracking: a product that lands one bf16 ulp elsewhere (fp32 summation order) is amplified by the following post-LN
# layers exactly like the rounding error itself, so two executions of the SAME rounding policy end as far apart as either is from fp32.  The shallow
# goldens therefore carry the tight bound; the deep ones are bounded at the fp32 allowance here, and their kernels' logic is what bf16x3 pins at 1e-3 on
# all thirteen (same kernels' structure, three products).
TOL_EMU_FWD = {"bf16": (3e-2, 8e-2), "f16": (4.5e-3, 1.2e-2)}          # (<= 2 text layers, deeper)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("name", FWD_GOLDENS)
def test_16bit_forward_tracks_the_emulating_oracle_on_every_golden(name, dtype):
    z, cfg = load(name + ".npz")
    txt = "bert" if cfg.txt_kind == "bert" else "bernice"
    B, T = int(z["B"]), int(z["T"])
    model = build(cfg, dtype, txt, B, T)
    P0 = O.make_params(cfg, int(z["seed_w"]))
    load_oracle_params(model, P0)
    model.eval()
    pixels = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), bool(z["pad"]))[2]
    with torch.no_grad():
        out_cls, lpt, out_tim, _, feats = model(t(z, "ids"), t(z, "mask"), pixels, tim_inputs=(t(z, "tim_ids"), t(z, "tim_mask")))
        with O.rounding(dtype):
            r_cls, r_lpt, r_tim, _, r_feats = O.mm_forward(P0, t(z, "ids"), t(z, "mask"), pixels, cfg, (t(z, "tim_ids"), t(z, "tim_mask")),
                                                           relu_mask=(feats.detach().cpu() > 0).float())
    errs = {k: rel_err(a, b) for k, a, b in (("out_cls", out_cls, r_cls), ("logits_per_text", lpt, r_lpt), ("mm_features", feats, r_feats))}
    print("EMU_FWD", name, dtype, {k: float("%.3g" % e) for k, e in errs.items()}, "out_tim (ReLU units not shared) %.3g" % rel_err(out_tim, r_tim))
    for k, e in errs.items():
        assert e < TOL_EMU_FWD[dtype][0 if cfg.layers_txt <= 2 else 1], (k, e)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "bf16x3"])
def test_dropout_train_step_matches_oracle_with_replayed_masks(dtype):
    """dropout ON: the oracle replays the kernels' counter-based masks (same hash), so loss and gradients must agree"""
    cfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=500, max_pos=130, num_labels=3)
    B, T, seed = 4, 64, 0xC0FFEE1234
    model = build(cfg, dtype, "bernice", B, T)
    P = O.make_params(cfg, 5)
    load_oracle_params(model, P)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 31, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    dev = model.device_
    from smtc_amd import _lib
    model._flat_grad.zero_()
    model._engine_forward(ids, mask, pixels, tim_ids, tim_mask, seed=seed)
    lo = torch.empty(4, device=dev)
    oh, lt = onehot.to(dev).contiguous(), lbl.to(dev)
    _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, _lib.ptr(lt), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
    _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
    Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    out_cls, lpt, out_tim, _, _ = O.mm_forward(Pg, ids, mask, pixels, cfg, (tim_ids, tim_mask), O.Dropout("hash", seed))
    ref = O.mix_loss(out_cls, onehot, None, lpt, out_tim, lbl, True, True)
    ref.backward()
    assert abs(lo[0].item() - ref.item()) < 2e-3 * abs(ref.item()), (lo[0].item(), ref.item())
    allerr = {}
    for i in model._train_params:
        k = i["name"]
        if Pg[k].grad is None or k.endswith("key.bias") or k == "fc_K.bias":
            continue
        g = model._flat_grad[i["offset"]: i["offset"] + i["numel"]].view(i["shape"]).float().cpu()
        e = (g - Pg[k].grad).norm().item() / max(Pg[k].grad.norm().item(), 1e-20)
        allerr[k] = float("%.3g" % e)
        assert e < TOL_GRAD[dtype], (k, e)
    print("GRADERR", dtype, "dropout_replay", {k: v for k, v in sorted(allerr.items(), key=lambda kv: -kv[1])[:12]})
    assert float(np.median(list(allerr.values()))) < TOL_GRAD_MEDIAN[dtype], sorted(allerr.values())


def build_clip(cfg, dtype, B, T):
    arch = dict(layers_txt=cfg.layers_txt, layers_img=cfg.layers_img, vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                p_hidden=cfg.p_hidden, p_attn=cfg.p_attn, hidden_img=cfg.Hv, heads_img=cfg.heads_v, inter_img=cfg.Iv, image=cfg.image)
    return MM_Model(cfg.num_labels, "bernice", "clip", cfg.p_head, cfg.fusion, arch=arch, dtype=dtype, max_posts=B, max_text_len=T)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "bf16x3"])
@pytest.mark.parametrize("name", ["clip_small_224", "clip_small_336"])
def test_config4_clip_tower_concat_fusion(name, dtype):
    """BASELINE config 4 (CLIP-ViT-L/14-shaped image tower + concat fusion): logits_per_text against the
    HuggingFace-generated golden (tests/golden/make_clip_golden.py), heads / ITM / loss / gradients against the oracle.
    257 tokens at 224, 577 at 336 (parity mode: the keys are walked in LDS-sized chunks, attention.hip attn_fwd_x3_long_kernel -- round 4; the
    case was skipped before)."""
    z, cfg = load(name + ".npz")
    B, T = int(z["B"]), int(z["T"])
    model = build_clip(cfg, dtype, B, T)
    P = O.make_params(cfg, int(z["seed_w"]))
    load_oracle_params(model, P)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, int(z["seed_x"]), True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    dev = model.device_
    model.train()                                            # dropout p = 0 in this configuration
    out_cls, lpt, out_tim, _, feats = model(ids, mask, pixels, tim_inputs=(tim_ids, tim_mask))
    assert rel_err(lpt, t(z, "logits_per_text")) < TOL_OUT[dtype]["logits_per_text"], rel_err(lpt, t(z, "logits_per_text"))
    loss = O.mix_loss(out_cls, onehot.to(dev), None, lpt, out_tim, lbl.to(dev), True, True)
    loss.backward()
    Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    r_cls, r_lpt, r_tim, _, r_feats = O.mm_forward(Pg, ids, mask, pixels, cfg, (tim_ids, tim_mask))
    ref = O.mix_loss(r_cls, onehot, None, r_lpt, r_tim, lbl, True, True)
    ref.backward()
    errs = {k: rel_err(a.detach(), b.detach()) for k, a, b in (("out_cls", out_cls, r_cls), ("logits_per_text", lpt, r_lpt), ("out_tim", out_tim, r_tim),
                                                               ("mm_features", feats, r_feats))}
    print(name, dtype, errs, "loss", loss.item(), ref.item())
    for k, e in errs.items():
        assert e < TOL_OUT[dtype][k], (k, e)
    assert abs(loss.item() - ref.item()) < TOL_LOSS[dtype] * abs(ref.item())
    named = dict(model.named_parameters())
    for k in ("linear_fusion.weight", "linear_cls.weight", "dual_encoder.visual_projection.weight", "dual_encoder.text_model.encoder.layer.0.output.dense.weight"):
        g, r = named[k].grad.float().cpu(), Pg[k].grad
        e = (g - r).norm().item() / max(r.norm().item(), 1e-20)
        # B = 3 posts: more cancellation per tensor than the 8-post golden case behind TOL_GRAD (measured 0.037 in f16)
        assert e < {"bf16": 0.6, "f16": 0.08, "bf16x3": 1e-3}[dtype], (k, e)
    assert named["fc_K.weight"].grad is None                 # concat: the fusion-attention group receives no gradient


def test_adamw_matches_golden_and_skips_inactive():
    import ctypes as C
    from smtc_amd import _lib
    z = np.load(os.path.join(GOLD, "adamw.npz"))
    dev = torch.device("cuda:0")
    p = t(z, "p0").to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i in range(3):
        g = t(z, "grads")[i].to(dev).clone()
        _lib.check(_lib.lib().mmhip_adamw(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), float(z["lr"]), 0.9, 0.999, 1e-8,
                                          float(z["wd"]), i + 1, 1.0, 1, _lib.stream_ptr()))
        assert (p.cpu() - t(z, "traj")[i]).abs().max().item() < 2e-8
        assert g.abs().max().item() == 0.0


def test_adamw_guards_against_nonfinite_gradients():
    """an inf / NaN in the gradient (f16 overflow) must not poison m / v / p: the element only decays, the counter moves, and the
    trainer halves the f16 loss scale (or raises for the other dtypes)"""
    import types
    from smtc_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.mmhip_set_nonfinite_counter(_lib.ptr(cnt)))
    n = 4099
    p = torch.randn(n, device=dev)
    p0 = p.clone()
    g = torch.randn(n, device=dev)
    g[5], g[1000], g[4098] = float("inf"), float("nan"), float("-inf")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    _lib.check(lib.mmhip_adamw(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 1, _lib.stream_ptr()))
    assert torch.isfinite(p).all() and torch.isfinite(m).all() and torch.isfinite(v).all() and int(cnt.item()) >= 2
    for i in (5, 1000, 4098):
        assert m[i].item() == 0.0 and v[i].item() == 0.0 and abs(p[i].item() - p0[i].item() * (1 - 1e-3 * 0.01)) < 1e-7
    _lib.check(lib.mmhip_set_nonfinite_counter(None))          # the process-wide registration must not outlive `cnt`
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=32, dropout=0.0)
    arch = dict(layers_txt=1, layers_img=1, vocab=300, max_pos=130)
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, dtype="f16")
    tr.model._nonfinite.fill_(3)
    assert tr.check_overflow() == 3 and tr.model._loss_scale == 512.0 and int(tr.model._nonfinite[0].item()) == 0
    tr2 = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, dtype="bf16")
    tr2.model._nonfinite.fill_(1)
    with pytest.raises(FloatingPointError):
        tr2.check_overflow()


def test_step_guard_skips_a_void_step_as_a_whole():
    """include/mmhip.h mmhip_set_step_guard: with the flag raised the AdamW entry points leave p / m / v (and the word rows'
    moments) alone and only clear the gradient; an f16 step whose gradient chain overflows raises the flag on the device by itself, the
    whole step is void -- parameters, moments and row flags bit-identical to before, gradient buffer clean -- and the host, reading the
    counter one step late and without synchronising, halves the loss scale; with a sane scale training goes on."""
    import types
    from smtc_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    # ---- operator level
    words = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    _lib.check(lib.mmhip_set_step_guard(_lib.ptr(words)))
    n = 4099
    p, g = torch.randn(n, device=dev), torch.randn(n, device=dev)
    m, v = torch.rand(n, device=dev), torch.rand(n, device=dev)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    _lib.check(lib.mmhip_adamw(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 1, _lib.stream_ptr()))
    assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and g.abs().max().item() == 0.0
    rows, width = 10, 64
    P, G = torch.randn(rows, width, device=dev), torch.randn(rows, width, device=dev)
    M, V = torch.rand(rows, width, device=dev), torch.rand(rows, width, device=dev)
    st = torch.tensor([0, 1, 2, 3, 1, 0, 2, 3, 1, 1, 0, 0], dtype=torch.uint8, device=dev)
    P0, M0, V0 = P.clone(), M.clone(), V.clone()
    _lib.check(lib.mmhip_adamw_rows(_lib.ptr(P), _lib.ptr(G), _lib.ptr(M), _lib.ptr(V), rows, width, _lib.ptr(st), 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 1,
                                    _lib.stream_ptr()))
    assert torch.equal(P, P0) and torch.equal(M, M0) and torch.equal(V, V0)
    had = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1, 1, 1], dtype=torch.bool, device=dev)
    assert G[had].abs().max().item() == 0.0 and st[:rows].tolist() == [0, 0, 2, 2, 0, 0, 2, 2, 0, 0]
    words.zero_()
    _lib.check(lib.mmhip_adamw(_lib.ptr(p), _lib.ptr(torch.ones_like(p)), _lib.ptr(m), _lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 1, _lib.stream_ptr()))
    assert not torch.equal(p, p0)                                      # flag down: the update runs
    _lib.check(lib.mmhip_set_step_guard(None))
    # ---- a whole f16 step whose backward overflows
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=32, dropout=0.0)
    arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130)
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, dtype="f16")
    mm = tr.model
    ids, mask, px, oh = synthetic_batch(mm.arch["vocab"], 3, 4, 32, 7, mm.arch["txt_kind"], mm.arch["pad_id"], True, mm.arch["image"], dev)
    tr.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 1)               # a normal step: moments exist, rows are flagged
    torch.cuda.synchronize()
    before = (mm._flat_train.clone(), tr._opt[0].clone(), tr._opt[1].clone(), mm._word_row_state.clone())
    mm._loss_scale = 2.0 ** 40                                          # every 16-bit gradient of the chain overflows
    _lib.check(lib.mmhip_set_loss_scale(mm._handle, mm._loss_scale))
    tr.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 2)
    torch.cuda.synchronize()
    assert int(mm._nonfinite[0].item()) > 0
    assert torch.equal(mm._flat_train, before[0]) and torch.equal(tr._opt[0], before[1]) and torch.equal(tr._opt[1], before[2])
    assert torch.equal(mm._word_row_state, before[3]) and mm._flat_grad.abs().max().item() == 0.0
    tr.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 3)               # its entry reads the counter of step 2: the scale halves
    assert mm._loss_scale == 2.0 ** 39
    torch.cuda.synchronize()
    tr._poll_guard()                                                    # step 3 overflowed as well (2^39): consume its sightings
    assert mm._loss_scale == 2.0 ** 38
    mm._loss_scale = 1024.0
    _lib.check(lib.mmhip_set_loss_scale(mm._handle, mm._loss_scale))
    torch.cuda.synchronize()
    mid = mm._flat_train.clone()
    tr.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 4)
    torch.cuda.synchronize()
    assert not torch.equal(mm._flat_train, mid) and torch.isfinite(mm._flat_train).all() and mm._loss_scale == 1024.0


def test_guard_is_per_handle():
    """include/mmhip.h mmhip_set_guard: two models in one process own separate {counter, flag} words -- an overflowing
    f16 model must not void (or count into) the step of a second model created after it, whichever steps last"""
    import types
    from smtc_amd import _lib
    dev = torch.device("cuda")
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=32, dropout=0.0)
    arch = dict(layers_txt=2, layers_img=1, vocab=300, max_pos=130)
    a = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3, dtype="f16")
    b = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=4, dtype="f16")       # created later: the old process-wide slot was its
    ids, mask, px, oh = synthetic_batch(a.model.arch["vocab"], 3, 4, 32, 7, a.model.arch["txt_kind"], a.model.arch["pad_id"], True, a.model.arch["image"], dev)
    a.model._loss_scale = 2.0 ** 40
    _lib.check(_lib.lib().mmhip_set_loss_scale(a.model._handle, a.model._loss_scale))
    pa, pb = a.model._flat_train.clone(), b.model._flat_train.clone()
    a.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 1)                # overflows: void on ITS handle
    b.train_step(ids, mask, px, oh, None, 1e-3, 0.01, 1)                # a clean step of the other model
    torch.cuda.synchronize()
    assert int(a.model._nonfinite[0].item()) > 0 and int(a.model._nonfinite[1].item()) == 1
    assert int(b.model._nonfinite[0].item()) == 0 and int(b.model._nonfinite[1].item()) == 0
    assert torch.equal(a.model._flat_train, pa) and not torch.equal(b.model._flat_train, pb)


@pytest.mark.parametrize("lr,wd", [(1e-3, 0.01), (1e-5, 2.5e-4)])
def test_adamw_rows_bit_identical_to_dense(lr, wd):
    """row-lazy AdamW over an embedding table (rows without gradient and moments only decay) == the dense kernel, bit for
    bit, over steps with changing touched-row sets; flags follow the protocol of include/mmhip.h.  Second case: the
    reference's default lr x weight_decay = 2.5e-9, where 1 - lr*wd is exactly 1.0f and the decay-only rows are not touched"""
    from smtc_amd import _lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    V, H = 1003, 768
    p_d = torch.randn(V, H, generator=gen).to(dev)
    p_r = p_d.clone()
    m_d, v_d, m_r, v_r = (torch.zeros(V, H, device=dev) for _ in range(4))
    state = torch.zeros((V + 3) // 4 * 4, dtype=torch.uint8, device=dev)
    g_r = torch.zeros(V, H, device=dev)
    lib = _lib.lib()
    for step in range(1, 6):
        rows = torch.randperm(V, generator=gen)[: 40 + 10 * step].to(dev)
        g = torch.zeros(V, H, device=dev)
        g[rows] = torch.randn(len(rows), H, generator=gen).to(dev)
        g_r += g
        state[rows] = state[rows] | 1
        g_d = g.clone()
        _lib.check(lib.mmhip_adamw(_lib.ptr(p_d), _lib.ptr(g_d), _lib.ptr(m_d), _lib.ptr(v_d), p_d.numel(), lr, 0.9, 0.999, 1e-8, wd, step, 0.5, 1,
                                   _lib.stream_ptr()))
        _lib.check(lib.mmhip_adamw_rows(_lib.ptr(p_r), _lib.ptr(g_r), _lib.ptr(m_r), _lib.ptr(v_r), V, H, _lib.ptr(state), lr, 0.9, 0.999, 1e-8,
                                        wd, step, 0.5, 1, _lib.stream_ptr()))
        assert torch.equal(p_d, p_r) and torch.equal(m_d, m_r) and torch.equal(v_d, v_r), step
        assert g_r.abs().max().item() == 0.0
        st = state[:V].cpu()
        assert int((st & 1).sum()) == 0 and int((st == 2).sum()) == int((m_d.abs().sum(1) > 0).sum())
    # zero_grad = 0 keeps gradient and flag
    rows = torch.arange(7, device=dev)
    g_r[rows] = 1.0
    state[rows] = state[rows] | 1
    _lib.check(lib.mmhip_adamw_rows(_lib.ptr(p_r), _lib.ptr(g_r), _lib.ptr(m_r), _lib.ptr(v_r), V, H, _lib.ptr(state), 1e-3, 0.9, 0.999, 1e-8, 0.01, 6,
                                    1.0, 0, _lib.stream_ptr()))
    assert g_r[:7].min().item() == 1.0 and int((state[:7] & 1).sum()) == 7
    assert lib.mmhip_adamw_rows(_lib.ptr(p_r), _lib.ptr(g_r), _lib.ptr(m_r), _lib.ptr(v_r), V, 770, _lib.ptr(state), 1e-3, 0.9, 0.999, 1e-8, 0.01, 1,
                                1.0, 1, _lib.stream_ptr()) < 0


def test_trainer_word_table_equals_dense_adamw():
    """MMLate_Model.train_step (row-lazy word table) == the same steps with one dense AdamW over the whole buffer"""
    import types
    from smtc_amd import _lib
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1,
                                 max_length=32, dropout=0.05)
    arch = dict(layers_txt=1, layers_img=1, vocab=3000, max_pos=130)
    ocfg = O.OracleConfig(layers_txt=1, layers_img=1, vocab=3000, max_pos=130, num_labels=3)
    a = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3)
    b = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3)

    def dense(self, lr, wd, step, dense=True, rows=True):
        m = self.model
        if not dense:
            return                                # one dense launch covers the word table too
        if self._opt is None:
            self._opt = (torch.zeros_like(m._flat_train), torch.zeros_like(m._flat_train))
        at = lambda tns, el: C.c_void_p(tns.data_ptr() + el * 4)
        for bb, ee in m.active_ranges(False, False):
            _lib.check(_lib.lib().mmhip_adamw(at(m._flat_train, bb), at(m._flat_grad, bb), at(self._opt[0], bb), at(self._opt[1], bb), ee - bb, lr,
                                              0.9, 0.999, 1e-8, wd, step, 1.0, 1, _lib.stream_ptr()))
    import ctypes as C
    b._adamw = types.MethodType(dense, b)
    calls = {"n": 0}
    orig_dense = b._adamw

    def counted(*args, **kw):
        calls["n"] += 1
        return orig_dense(*args, **kw)
    b._adamw = counted
    for step in range(1, 5):
        ids, mask, pixels, onehot = O.synthetic_batch(ocfg, 4, 32, 100 + step, True)
        for tr in (a, b):
            tr.model._calls = step               # same dropout stream in both trainers
            # trainer a: the default native step (mmhip_train_step: per-layer AdamW on the side stream, row-lazy word table);
            # trainer b: the staged step, the only one that goes through _adamw -- here ONE dense AdamW over every active range
            os.environ["MMHIP_NATIVE_STEP"] = "1" if tr is a else "0"
            try:
                tr.train_step(ids.cuda(), mask.cuda(), pixels, onehot, None, 1e-3, 0.01, step)
            finally:
                os.environ.pop("MMHIP_NATIVE_STEP", None)
    assert calls["n"] == 8, calls                # two _adamw calls (dense ranges, word rows) per staged step: the dense closure really ran
    # the backward's fp32 atomics are not run-to-run bit-stable, so touched values are compared closely, not bitwise
    assert (a.model._flat_train - b.model._flat_train).abs().max().item() < 2e-5
    assert (a._opt[0] - b._opt[0]).abs().max().item() < 1e-5 and (a._opt[1] - b._opt[1]).abs().max().item() < 1e-5
    assert a.model._flat_grad.abs().max().item() == 0.0 and b.model._flat_grad.abs().max().item() == 0.0
    st = a.model._word_row_state[:3000]
    touched = int((st == 2).sum())
    assert 0 < touched < 3000 and int((st & 1).sum()) == 0
    w = a.model._word_info
    ta = a.model._flat_train[w["offset"]: w["offset"] + w["numel"]].view(w["shape"])
    tb = b.model._flat_train[w["offset"]: w["offset"] + w["numel"]].view(w["shape"])
    assert torch.equal(ta[st == 0], tb[st == 0])            # decay-only rows: bit-identical to the dense update
    assert a._opt[0][w["offset"]: w["offset"] + w["numel"]].view(w["shape"])[st == 0].abs().max().item() == 0.0


def _fresh_trainer(seed=3, itc=True, itm=True):
    import types
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=itc, beta_itc=0.1, use_tim_loss=itm, beta_itm=0.1,
                                 max_length=32, dropout=0.05)
    arch = dict(layers_txt=2, layers_img=1, vocab=3000, max_pos=130)
    return MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=seed)


def _run_steps(tr, n, env):
    ocfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=3000, max_pos=130, num_labels=3)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        losses = []
        for step in range(1, n + 1):
            ids, mask, pixels, onehot = O.synthetic_batch(ocfg, 4, 32, 100 + step, True)
            tr.model._calls = step
            np.random.seed(30 + step)                # the ITM sampling draws from numpy's global stream (reference mm_late.py:389-414)
            loss, _ = tr.train_step(ids.cuda(), mask.cuda(), pixels, onehot, None, 1e-3, 0.01, step)
            losses.append(loss.clone())
        torch.cuda.synchronize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return torch.stack(losses).cpu()


def test_deterministic_mode_is_bitwise_reproducible():
    """MMHIP_DETERMINISTIC=1: no fp32 atomic decides the last bits of a sum (single-writer second stage of the column reductions, word /
    position rows summed in slot order): two runs of three steps -- dropout on, ITC + ITM on, the default native step with its side
    streams -- end on bit-identical parameters, moments, row flags and losses.  A stream race (a missing wait between the side-stream
    AdamW / weight refresh and the kernels that read those weights) would show up here; rounding noise can not."""
    res = []
    for _ in range(2):
        tr = _fresh_trainer()
        losses = _run_steps(tr, 3, {"MMHIP_DETERMINISTIC": "1"})
        res.append((tr.model._flat_train.clone(), tr._opt[0].clone(), tr._opt[1].clone(), tr.model._word_row_state.clone(), losses))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    assert res[0][0].isfinite().all() and float(res[0][4][-1][0]) > 0


def test_native_and_staged_steps_agree_bitwise_in_deterministic_mode():
    """one parameter update three ways from identical state: the native step with each text layer's AdamW + 16-bit weight refresh on the
    side stream behind that layer's weight-gradient GEMM (default), the native step with the optimizer after the whole backward
    (MMHIP_EARLY_ADAMW=0), and the staged Python step (MMHIP_NATIVE_STEP=0).  Same kernels, same elementwise AdamW: with the deterministic
    reductions the parameters, moments and row flags after TWO steps are bit-identical -- the second step's forward reads the 16-bit
    weight copies the first step's (side-stream) refresh wrote, so a missing wait there cannot hide."""
    res = []
    for env in ({"MMHIP_EARLY_ADAMW": "1"}, {"MMHIP_EARLY_ADAMW": "0"}, {"MMHIP_NATIVE_STEP": "0"}):
        tr = _fresh_trainer()
        losses = _run_steps(tr, 2, dict(env, MMHIP_DETERMINISTIC="1"))
        res.append((tr.model._flat_train.clone(), tr._opt[0].clone(), tr._opt[1].clone(), tr.model._word_row_state.clone(), losses))
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert torch.equal(x, y), (x.float() - y.float()).abs().max()


def test_tiny_image_tower_splits_k_beside_the_text_tower():
    """image 32 / patch 16: the image tower has B * 5 <= 128 rows, so its FC2 (K = 3072) takes the split-K path on the tower's side
    stream while the CLS-row GEMMs of the last text layer take it on the caller's stream -- each stream has its own fp32 scratch
    (they used to share one).  Outputs match the oracle and repeat bit for bit."""
    cfg = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, num_labels=3, image=32, patch=16, p_hidden=0.0, p_attn=0.0, p_head=0.0)
    arch = dict(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, image=32, patch=16, p_hidden=0.0, p_attn=0.0)
    B, T = 8, 32
    model = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, dtype="bf16", max_posts=B, max_text_len=T)
    P = O.make_params(cfg, 11)
    load_oracle_params(model, P)
    model.eval()
    ids, mask, pixels, _ = O.synthetic_batch(cfg, B, T, 5, True)
    with torch.no_grad():
        ref = O.mm_forward(P, ids, mask, pixels, cfg, None)
        outs = [model(ids, mask, pixels) for _ in range(6)]
    for k, i in (("out_cls", 0), ("logits_per_text", 1), ("mm_features", 4)):
        assert rel_err(outs[0][i], ref[i]) < TOL_OUT["bf16"][k], (k, rel_err(outs[0][i], ref[i]))
        for o in outs[1:]:
            assert torch.equal(o[i], outs[0][i]), k


def test_trainer_step_and_itm_sampling():
    """fused MMLate_Model.train_step: loss goes down on a fixed batch; parameters outside the active set stay untouched;
    prepare_itm_inputs reproduces the reference's numpy RNG stream (tests/golden/itm_sampling.npz)"""
    import types
    cfgd = types.SimpleNamespace(batch_size=8, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1,
                                 max_length=64, dropout=0.05)
    arch = dict(layers_txt=2, layers_img=1, vocab=500, max_pos=130)
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=arch, seed=3)
    z = np.load(os.path.join(GOLD, "itm_sampling.npz"))
    for B in (1, 2, 8, 64):
        np.random.seed(30)
        a, b, c = tr.prepare_itm_inputs(t(z, f"B{B}.ids"), t(z, f"B{B}.mask"))
        assert torch.equal(a.cpu(), t(z, f"B{B}.tim_ids")) and torch.equal(b.cpu(), t(z, f"B{B}.tim_mask")) and torch.equal(c.cpu(), t(z, f"B{B}.lbl"))
    ocfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=500, max_pos=130, num_labels=3)
    ids, mask, pixels, onehot = O.synthetic_batch(ocfg, 8, 64, 5, True)
    never = {k: p.detach().clone() for k, p in tr.model.named_parameters() if k.split(".")[0] in ("aspectattention", "linear_iadds", "linear_gmu_t", "linear_gmu_v")}
    frozen = tr.model._flat_frozen.clone()
    np.random.seed(30)
    losses = []
    for step in range(1, 9):
        loss, nc = tr.train_step(ids.cuda(), mask.cuda(), pixels, onehot, None, 1e-3, 0.00025, step)
        losses.append(loss[0].item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    for k, p0 in never.items():
        assert torch.equal(dict(tr.model.named_parameters())[k].detach(), p0), k
    assert torch.equal(tr.model._flat_frozen, frozen)
    assert tr.model._flat_grad.abs().max().item() == 0.0       # fused AdamW leaves the gradient buffer cleared


def test_full_size_properties():
    """B=64, T=128, 12+12 layers, Bernice-shaped vocabulary: (a) finite outputs, (b) eval determinism, (c) permuting the
    posts permutes the outputs (and transposes logits_per_text accordingly), (d) tokens under mask=0 do not matter"""
    cfg = O.OracleConfig(num_labels=2)
    model = MM_Model(2, "bernice", "vit", 0.05, "attention", max_posts=64, max_text_len=128, seed=1)
    model.eval()
    ids, mask, pixels, _ = O.synthetic_batch(cfg, 64, 128, 1234, True)
    with torch.no_grad():
        a = model(ids, mask, pixels)
        b = model(ids, mask, pixels)
        perm = torch.randperm(64, generator=torch.Generator().manual_seed(0))
        c = model(ids[perm], mask[perm], pixels[perm])
        junk = ids.clone()
        junk[mask == 0] = 777
        junk_posids_safe = cfg.txt_kind != "xlmr"      # XLM-R position ids look at ids != pad: keep pads as pads there
        d = model(junk if junk_posids_safe else ids, mask, pixels)
    for x in (a[0], a[1], a[4]):
        assert torch.isfinite(x).all()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
    assert rel_err(c[0], a[0][perm.cuda()]) < 1e-5 and rel_err(c[4], a[4][perm.cuda()]) < 1e-5
    assert rel_err(c[1], a[1][perm.cuda()][:, perm.cuda()]) < 1e-5
    assert torch.equal(d[0], a[0])


@pytest.mark.parametrize("B,T,itm", [(3, 50, True), (1, 32, True), (5, 128, False), (2, 7, False)])
def test_ragged_shapes_match_oracle(B, T, itm):
    """row counts that are not multiples of any tile (generic GEMM paths, partial attention tiles, B == 1 ITM rule)"""
    cfg = O.OracleConfig(layers_txt=2, layers_img=1, vocab=400, max_pos=130, num_labels=4, p_hidden=0.0, p_attn=0.0, p_head=0.0)
    model = build(cfg, "f16", "bernice", B, T)
    P = O.make_params(cfg, 9)
    load_oracle_params(model, P)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 100 + B, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    tim = (tim_ids, tim_mask) if itm else None
    dev = model.device_
    out_cls, lpt, out_tim, _, feats = model(ids, mask, pixels, tim_inputs=tim)
    loss = O.mix_loss(out_cls, onehot.to(dev), None, lpt, out_tim, lbl.to(dev), True, itm)
    loss.backward()
    Pg = {k: v.clone().requires_grad_(O.trainable(k)) for k, v in P.items()}
    r = O.mm_forward(Pg, ids, mask, pixels, cfg, tim)
    ref = O.mix_loss(r[0], onehot, None, r[1], r[2], lbl, True, itm)
    ref.backward()
    assert rel_err(out_cls, r[0].detach()) < TOL_OUT["f16"]["out_cls"] and rel_err(feats, r[4].detach()) < TOL_OUT["f16"]["mm_features"]
    assert abs(loss.item() - ref.item()) < 1e-3 * abs(ref.item())
    for k, p in model.named_parameters():
        if Pg[k].grad is None or k.endswith("key.bias") or k == "fc_K.bias":
            assert p.grad is None or k.endswith("key.bias") or k == "fc_K.bias", k
            continue
        e = (p.grad.cpu() - Pg[k].grad).norm().item() / max(Pg[k].grad.norm().item(), 1e-20)
        assert e < 0.05, (k, e)            # tiny batches (1-5 posts): more cancellation per tensor than the golden case behind TOL_GRAD


def test_token_ids_outside_the_table_are_clamped_not_followed():
    """a token id outside the word table (a tokenizer that does not match the checkpoint) must not send the embedding gather / the
    gradient scatter outside the table: the engine clamps ids into [0, vocab) on their way into its own copy, so forward, backward and the
    row-sparse optimizer see row vocab - 1 (the reference raises IndexError on the CPU; a GPU kernel reading past the table faults the card)"""
    from smtc_amd import _lib
    cfg = O.OracleConfig(layers_txt=1, layers_img=1, vocab=300, max_pos=130, num_labels=2)
    model = build(cfg, "bf16", "bernice", 4, 32)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, 4, 32, 3, True)
    bad = ids.clone()
    bad[0, 3], bad[1, 5], bad[2, 7] = 10 ** 12, 300, -7
    want = bad.clamp(0, 299)
    model.train()
    outs = []
    for x in (bad, want):
        model._flat_grad.zero_()
        o = model._engine_forward(x, mask, pixels, x.flip(0), mask.flip(0), seed=5)      # (same dropout masks in both runs)
        lo = torch.empty(4, device=model.device_)
        oh, lt = onehot.to(model.device_).contiguous(), torch.tensor([1, 0, 1, 0], device=model.device_)
        _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, _lib.ptr(lt), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(([t.clone() for t in o if t is not None], lo.clone(), model._flat_grad.clone()))
    (oa, la, ga), (ob, lb, gb) = outs
    assert all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(la, lb) and torch.isfinite(ga).all()
    assert rel_err(ga, gb) < 1e-6                      # (atomics: the order of the word-row sums is not fixed)
    # ... and they are COUNTED (include/mmhip.h mmhip_set_index_counter): the trainer turns the count into the reference's IndexError one step late
    # (pinned copy of the guard words, no synchronisation in the step loop) and at the end of an evaluation loop
    assert int(model._bad_index.item()) == 6           # three ids, met in `ids` and once more in the flipped ITM copy
    import types
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=2, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=32, dropout=0.0)
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=dict(layers_txt=1, layers_img=1, vocab=300, max_pos=130), seed=3)
    dev = tr.device
    tr.train_step(want.to(dev), mask.to(dev), pixels, onehot, None, 1e-5, 0.0, 1)
    tr.train_step(bad.to(dev), mask.to(dev), pixels, onehot, None, 1e-5, 0.0, 2)          # the step itself goes through on the clamped rows
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        tr.train_step(want.to(dev), mask.to(dev), pixels, onehot, None, 1e-5, 0.0, 3)     # ... the next one reports it
    tr.train_step(want.to(dev), mask.to(dev), pixels, onehot, None, 1e-5, 0.0, 4)         # reported once
    tr.check_indices()


def test_capacity_growth_param_updates_and_errors():
    from smtc_amd import _lib
    cfg = O.OracleConfig(layers_txt=1, layers_img=1, vocab=300, max_pos=130, num_labels=2)
    model = build(cfg, "bf16", "bernice", 2, 32)
    model.eval()
    ids, mask, pixels, _ = O.synthetic_batch(cfg, 6, 64, 3, True)
    with torch.no_grad():
        small = model(ids[:2, :32], mask[:2, :32], pixels[:2])[0]
        big = model(ids, mask, pixels)[0]                    # B and T beyond the creation capacity: workspace is re-made
        again = model(ids[:2, :32], mask[:2, :32], pixels[:2])[0]
        assert big.shape == (6, 2) and torch.equal(small, again)
        # in-place parameter change (what optimizer.step / load_state_dict do) must reach the 16-bit GEMM operands
        w = dict(model.named_parameters())["dual_encoder.text_model.encoder.layer.0.intermediate.dense.weight"]
        w.mul_(1.5)
        changed = model(ids[:2, :32], mask[:2, :32], pixels[:2])[0]
        assert not torch.equal(changed, small)
        w.div_(1.5)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(sd)
        assert rel_err(model(ids[:2, :32], mask[:2, :32], pixels[:2])[0], small) < 1e-6
        with pytest.raises(ValueError):
            model(ids[:2], mask[:2], pixels[:2, :, :200, :200])
        with pytest.raises(_lib.MMHipError):
            model(torch.cat([ids, ids, ids], 1)[:2, :160], torch.cat([mask, mask, mask], 1)[:2, :160], pixels[:2])   # T > 128
    with pytest.raises(NotImplementedError):
        MM_Model(2, "bernice", "vit", 0.05, "gmu")
    # train() applies dropout, eval() does not
    model.train()
    with torch.no_grad():
        a, b = model(ids[:2, :32], mask[:2, :32], pixels[:2])[0], model(ids[:2, :32], mask[:2, :32], pixels[:2])[0]
    assert not torch.equal(a, b)


def test_full_size_backward_is_the_mean_of_half_batches():
    """BASELINE size (12+12 layers, V=250002, T=128): without ITC/ITM posts are independent, so the gradient of the mean loss
    over 64 posts equals the mean of the gradients over its two halves (dropout off).  Exercises every backward kernel at
    the benchmark shapes without needing the CPU oracle at that size."""
    from smtc_amd import _lib
    cfg = O.OracleConfig(num_labels=2)
    model = MM_Model(2, "bernice", "vit", 0.0, "attention", arch=dict(p_hidden=0.0, p_attn=0.0), max_posts=64, max_text_len=128, seed=2)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, 64, 128, 99, True)
    dev = model.device_

    def grads(sl):
        model._flat_grad.zero_()
        model._engine_forward(ids[sl], mask[sl], pixels[sl])
        lo = torch.empty(4, device=dev)
        oh = onehot[sl].to(dev).contiguous()
        _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, None, 1.0, 0.0, 0.0, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
        return model._flat_grad.clone(), lo[0].item()

    g_all, l_all = grads(slice(0, 64))
    g_a, l_a = grads(slice(0, 32))
    g_b, l_b = grads(slice(32, 64))
    assert torch.isfinite(g_all).all()
    assert abs(l_all - 0.5 * (l_a + l_b)) < 1e-5 * abs(l_all)
    ref = 0.5 * (g_a + g_b)
    for b, e in model.active_ranges(False, False):
        err = (g_all[b:e] - ref[b:e]).norm().item() / max(ref[b:e].norm().item(), 1e-30)
        assert err < 2e-3, (b, e, err)            # identical per-row arithmetic; only fp32 / 16-bit summation order differs


@pytest.mark.parametrize("img_name,tokens", [("clip", 257), ("clip336", 577)])
def test_full_size_clip_l14_tower_properties(img_name, tokens):
    """BASELINE config 4 at its real size, BOTH legs ("224 -> 336 images"): CLIP-ViT-L/14 (1024 wide, 16 heads, 4096-wide quick-GELU MLP, 24 pre-LN
    layers), 257 tokens at 224 px and 577 at 336 px (18 464 image rows at bs = 32; the attention keeps 577 keys of a head in LDS); concat fusion,
    bs = 32: N and K = 1024 / 4096 GEMMs, the 640-column padded patch conv and these attention shapes run only here under the
    driver.  (a) finite, (b) eval determinism, (c) permuting the posts permutes the outputs, (d) the gradient of the mean loss over 32
    posts is the mean of the gradients over its two halves (the tower is frozen: text tower + heads)."""
    from smtc_amd import _lib
    B, T = 32, 128
    model = MM_Model(2, "bernice", img_name, 0.0, "concat", arch=dict(p_hidden=0.0, p_attn=0.0), max_posts=B, max_text_len=T, seed=4)
    a = model.arch
    assert (a["hidden_img"], a["heads_img"], a["inter_img"], a["layers_img"], a["patch"]) == (1024, 16, 4096, 24, 14)
    assert (a["image"] // a["patch"]) ** 2 + 1 == tokens
    ids, mask, pixels, onehot = synthetic_batch(a["vocab"], 2, B, T, 4321, a["txt_kind"], a["pad_id"], True, a["image"], "cpu")
    model.eval()
    with torch.no_grad():
        x = model(ids, mask, pixels)
        y = model(ids, mask, pixels)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
        z_ = model(ids[perm], mask[perm], pixels[perm])
    for v in (x[0], x[1], x[4]):
        assert torch.isfinite(v).all()
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[4], y[4])
    assert rel_err(z_[0], x[0][perm.cuda()]) < 1e-5 and rel_err(z_[4], x[4][perm.cuda()]) < 1e-5
    assert rel_err(z_[1], x[1][perm.cuda()][:, perm.cuda()]) < 1e-5
    model.train()
    dev = model.device_

    def grads(sl):
        model._flat_grad.zero_()
        model._engine_forward(ids[sl], mask[sl], pixels[sl])
        lo = torch.empty(4, device=dev)
        oh = onehot[sl].to(dev).contiguous()
        _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, None, 1.0, 0.0, 0.0, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
        return model._flat_grad.clone(), lo[0].item()

    g_all, l_all = grads(slice(0, B))
    g_a, l_a = grads(slice(0, B // 2))
    g_b, l_b = grads(slice(B // 2, B))
    assert torch.isfinite(g_all).all() and abs(l_all - 0.5 * (l_a + l_b)) < 1e-5 * abs(l_all)
    ref = 0.5 * (g_a + g_b)
    for b, e in model.active_ranges(False, False):
        err = (g_all[b:e] - ref[b:e]).norm().item() / max(ref[b:e].norm().item(), 1e-30)
        assert err < 2e-3, (b, e, err)


def test_full_size_itc_itm_properties():
    """BASELINE config 3 at its real size (B = 64, T = 128, 12 + 12 layers, ITC + ITM: the text tower runs 128 posts): permuting the posts --
    with the SAME swapped-text assignment carried along -- permutes out_cls / out_tim / mm_features and permutes logits_per_text in
    both indices (the contrastive head couples the posts, the permutation equivariance still holds); the three-term loss and every
    active gradient range are finite and the loss does not depend on the order of the posts."""
    from smtc_amd import _lib
    cfg = O.OracleConfig(num_labels=3)
    B, T = 64, 128
    model = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=dict(p_hidden=0.0, p_attn=0.0), max_posts=B, max_text_len=T, seed=6)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, 77, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl = O.prepare_itm_inputs(ids, mask)
    model.train()
    dev = model.device_

    def run(p):
        model._flat_grad.zero_()
        o = model._engine_forward(ids[p], mask[p], pixels[p], tim_ids[p], tim_mask[p])
        lo = torch.empty(4, device=dev)
        oh, lt = onehot[p].to(dev).contiguous(), lbl[p].to(dev).contiguous()
        _lib.check(_lib.lib().mmhip_loss(model._handle, _lib.ptr(oh), None, _lib.ptr(lt), 0.8, 0.1, 0.1, _lib.ptr(lo), None, _lib.stream_ptr()))
        _lib.check(_lib.lib().mmhip_backward(model._handle, None, None, None, None, _lib.stream_ptr()))
        return [t_.clone() for t_ in o], lo.clone(), model._flat_grad.clone()

    ident = torch.arange(B)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2))
    (c0, l0, t0, f0), lo0, g0 = run(ident)
    (c1, l1, t1, f1), lo1, g1 = run(perm)
    pc = perm.cuda()
    assert all(torch.isfinite(v).all() for v in (c0, l0, t0, f0, lo0, g0))
    assert rel_err(c1, c0[pc]) < 1e-5 and rel_err(t1, t0[pc]) < 1e-5 and rel_err(f1, f0[pc]) < 1e-5
    assert rel_err(l1, l0[pc][:, pc]) < 1e-5
    assert (lo1 - lo0).abs().max().item() < 2e-5 * lo0.abs().max().item()
    for b, e in model.active_ranges(True, True):
        err = (g1[b:e] - g0[b:e]).norm().item() / max(g0[b:e].norm().item(), 1e-30)
        assert err < 2e-3, (b, e, err)            # the same sums over posts in another order: fp32 / 16-bit summation order only


def test_vision_cache_is_bit_identical_and_skips_the_tower():
    """image-tower output cache (include/mmhip.h mmhip_vision_export/_import): a post seen again takes its tower outputs from
    HBM -- outputs, loss gradients and trained parameters are bit-identical to recomputing; mixed / over-capacity batches
    recompute; pixels == NULL without an import is refused"""
    import ctypes as C
    from smtc_amd import _lib
    arch = dict(layers_txt=1, layers_img=2, vocab=400, max_pos=130, p_hidden=0.0, p_attn=0.0)
    cfg = O.OracleConfig(layers_txt=1, layers_img=2, vocab=400, max_pos=130, num_labels=3)
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, 6, 32, 21, True)
    plain = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, max_posts=8, max_text_len=32, seed=4)
    cached = MM_Model(3, "bernice", "vit", 0.0, "attention", arch=arch, max_posts=8, max_text_len=32, seed=4)
    cached.enable_vision_cache(8)
    plain.eval(); cached.eval()
    keys = [101, 102, 103, 104, 105, 106]
    with torch.no_grad():
        ref = plain._engine_forward(ids, mask, pixels)
        a = cached._engine_forward(ids, mask, pixels, vision_keys=keys)              # miss: computes + exports
        assert cached._vcache["misses"] == 6 and cached._vcache["hits"] == 0
        garbage = torch.full_like(pixels, float("nan"))
        b = cached._engine_forward(ids, mask, garbage, vision_keys=keys)             # hit: pixels are not even looked at
        assert cached._vcache["hits"] == 6
        perm = [3, 0, 5, 1, 4, 2]
        c = cached._engine_forward(ids[perm], mask[perm], garbage, vision_keys=[keys[j] for j in perm])
        for x, y, z, w in zip(ref, a, b, c):
            if x is not None:
                assert torch.equal(x, y) and torch.equal(x, z)
        assert torch.equal(ref[0][perm], c[0]) and torch.equal(ref[3][perm], c[3])
        # a batch with an unknown key recomputes (and needs real pixels); over capacity: keys stay uncached
        d = cached._engine_forward(ids, mask, pixels, vision_keys=[101, 102, 103, 104, 105, 999])
        assert torch.equal(d[0], ref[0]) and len(cached._vcache["slots"]) == 7
        cached._engine_forward(ids, mask, pixels, vision_keys=[201, 202, 203, 204, 205, 206])
        assert len(cached._vcache["slots"]) == 8                                       # capacity 8: only one more fitted
    # gradients through a cached forward
    for m in (plain, cached):
        m.train()
        m._flat_grad.zero_()
    o1 = plain(ids, mask, pixels)
    (O.cls_loss(o1[0], onehot.to(o1[0].device).float(), None) + o1[1].float().pow(2).mean()).backward()
    out = cached._engine_forward(ids, mask, garbage, vision_keys=keys)
    assert torch.equal(out[0], o1[0].detach())
    lib = _lib.lib()
    assert lib.mmhip_forward(cached._handle, _lib.ptr(ids.cuda()), _lib.ptr(mask.cuda()), None, None, None, 6, 32, 0, 1, _lib.ptr(out[0]), _lib.ptr(out[1]), None,
                             _lib.ptr(out[3]), _lib.stream_ptr()) == -2                # pixels NULL without an import: MMHIP_E_STATE


def test_loads_huggingface_directories_like_from_vision_text_pretrained(tmp_path, monkeypatch):
    """MM_Model.__init__ with local HF model directories (the reference's from_vision_text_pretrained(img_dir, txt_dir),
    models/mm_late.py:59-61): architecture comes from config.json, every tower weight from the checkpoint (transformers 5
    ViT key names mapped back to the reference's 4.25.1 names), and the forward agrees with the installed transformers'
    own ViTModel / XLMRobertaModel run on those very weights"""
    import transformers
    from smtc_amd import mm_late as ML
    torch.manual_seed(7)
    vcfg = transformers.ViTConfig(num_hidden_layers=2)
    tcfg = transformers.XLMRobertaConfig(vocab_size=900, num_hidden_layers=2, max_position_embeddings=130, layer_norm_eps=1e-5, type_vocab_size=1,
                                         pad_token_id=1, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    vit, txt = transformers.ViTModel(vcfg).eval(), transformers.XLMRobertaModel(tcfg).eval()
    d_vit, d_txt = str(tmp_path / "vit"), str(tmp_path / "bernice")
    vit.save_pretrained(d_vit)
    txt.save_pretrained(d_txt)
    monkeypatch.setitem(ML.MODEL_DIR_DICT, "vit", d_vit)
    monkeypatch.setitem(ML.MODEL_DIR_DICT, "bernice", d_txt)
    model = MM_Model(3, "bernice", "vit", 0.05, "attention", max_posts=4, max_text_len=32, seed=1, dtype="f16")
    assert model.arch["layers_txt"] == 2 and model.arch["layers_img"] == 2 and model.arch["vocab"] == 900
    sd = model.state_dict()
    for k, v in txt.state_dict().items():
        if k.endswith("position_ids") or k.endswith("token_type_ids"):
            continue
        assert torch.equal(sd["dual_encoder.text_model." + k].cpu(), v), k
    n_vit = 0
    for k, v in vit.state_dict().items():
        rk = "dual_encoder.vision_model." + ML._vit_key_to_ref(k)
        assert rk in sd, rk
        assert torch.equal(sd[rk].cpu().reshape(v.shape), v), k
        n_vit += 1
    assert n_vit >= 2 * 16 + 6
    # forward: towers of the installed transformers on the same weights (fp32, CPU) vs the engine (f16 operands)
    cfg = O.OracleConfig(layers_txt=2, layers_img=2, vocab=900, max_pos=130, num_labels=3)
    ids, mask, pixels, _ = O.synthetic_batch(cfg, 4, 32, 3, True)
    model.eval()
    with torch.no_grad():
        out_cls, lpt, _, _, feats = model(ids, mask, pixels)
        tv = vit(pixel_values=pixels)
        tt = txt(input_ids=ids, attention_mask=mask)
        P = {k: v.detach().float().cpu() for k, v in sd.items()}
        txt_e = tt.pooler_output @ P["dual_encoder.text_projection.weight"].t()
        img_e = tv.pooler_output @ P["dual_encoder.visual_projection.weight"].t()
        txt_e, img_e = txt_e / txt_e.norm(dim=-1, keepdim=True), img_e / img_e.norm(dim=-1, keepdim=True)
        ref_lpt = txt_e @ img_e.t() * P["dual_encoder.logit_scale"].exp()
        ref = O.mm_forward(P, ids, mask, pixels, cfg)
    assert (lpt.cpu() - ref_lpt).abs().max().item() / ref_lpt.abs().max().item() < 2e-3
    assert (out_cls.cpu() - ref[0]).abs().max().item() / ref[0].abs().max().item() < 2e-3
    assert (feats.cpu() - ref[4]).abs().max().item() / ref[4].abs().max().item() < 2e-3


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("tag,itc,itm", [("plain", False, False), ("itcitm", True, True)])
def test_eval_loop_matches_reference_golden(dtype, tag, itc, itm):
    """MMLate_Model.eval against the dict returned by the REFERENCE's own eval loop (tests/golden/make_eval_golden.py,
    reference models/mm_late.py:534-638): three ragged batches (4 + 4 + 3 posts), class weights, ITM negatives re-sampled per
    batch from numpy's global stream: data ids and labels exact, mean loss within the dtype's band, predictions equal wherever
    the reference's top-2 logit margin exceeds the dtype's logit error"""
    import types
    z, cfg = load("eval_small_xlmr.npz")
    cfgd = types.SimpleNamespace(batch_size=4, num_labels=cfg.num_labels, use_clip_loss=itc, beta_itc=0.1, use_tim_loss=itm, beta_itm=0.1,
                                 max_length=int(z["T"]), dropout=cfg.p_head)
    arch = dict(layers_txt=cfg.layers_txt, layers_img=cfg.layers_img, vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab,
                p_hidden=cfg.p_hidden, p_attn=cfg.p_attn)
    tr = MMLate_Model(cfgd, "bernice", "vit", cfg.fusion, arch=arch, dtype=dtype)
    load_oracle_params(tr.model, O.make_params(cfg, int(z["seed_w"])))
    batches, first = [], 5000
    for B, seed in z["batches"].tolist():
        ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, int(z["T"]), seed, True)
        batches.append({"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "pixel_values": pixels.unsqueeze(1),
                        "labels": onehot, "data_id": torch.arange(first, first + B)})
        first += B
    np.random.seed(30)
    res = tr.eval(batches, torch.nn.CrossEntropyLoss(weight=t(z, "class_w")), tim_loss_fn=torch.nn.CrossEntropyLoss() if itm else None)
    assert not tr.model.training
    assert np.array_equal(res["data_id"], z[tag + ".data_id"]) and np.array_equal(res["labels"], z[tag + ".labels"])
    ref_loss = float(z[tag + ".loss"])
    rel = abs(res["loss"] - ref_loss) / ref_loss
    top2 = np.sort(z["out_cls"], axis=1)
    margin = top2[:, -1] - top2[:, -2]
    # (the logit error of THIS two-layer fixture, about twice what it measures -- not the thirteen-golden band of TOL_OUT, which full-depth cases set)
    sure = margin > 4 * {"bf16": 1.6e-2, "f16": 3.2e-3, "bf16x3": 1e-3}[dtype] * np.abs(z["out_cls"]).max()
    print(tag, dtype, "loss rel err", rel, "sure", int(sure.sum()), "of", len(sure))
    assert rel < {"bf16x3": 1e-4, "bf16": 5e-3, "f16": 1e-3}[dtype], (res["loss"], ref_loss)
    assert sure.sum() >= (len(sure) - 2 if dtype == "bf16x3" else 1)
    assert np.array_equal(res["predictions"][sure], z[tag + ".predictions"][sure])


def test_lockstep_forward_of_both_towers_matches_reference_golden():
    """MMHIP_LOCKSTEP=1 (opt-in): both towers layer by layer on one stream, same-named GEMMs of a layer paired in one persistent
    launch (csrc/gemm8.hip GemmNTPair) -- same goldens, same bands; the switch is read once per process, hence the subprocess"""
    import subprocess, sys
    env = dict(os.environ, MMHIP_LOCKSTEP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k",
                        "(forward_matches_reference_golden or train_losses_and_grads or ragged_shapes) and not bf16x3 and not lockstep"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_gemm_timing_counts_the_steps_nt_launches():
    """mmhip_gemm_timing / mmhip_gemm_timing_by_shape (bench.py's roofline source): events around every NT GEMM launch of a step,
    both stream configurations; launches, FLOPs and the per-shape table must agree with the architecture"""
    import ctypes as C
    import types
    from smtc_amd import _lib
    cfgd = types.SimpleNamespace(batch_size=8, num_labels=3, use_clip_loss=False, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=64, dropout=0.05)
    Lt, Lv, H, I = 2, 1, 768, 3072
    tr = MMLate_Model(cfgd, "bernice", "vit", "attention", arch=dict(layers_txt=Lt, layers_img=Lv, vocab=500, max_pos=130), seed=3)
    ocfg = O.OracleConfig(layers_txt=Lt, layers_img=Lv, vocab=500, max_pos=130, num_labels=3)
    ids, mask, pixels, onehot = O.synthetic_batch(ocfg, 8, 64, 5, False)
    lib, h = _lib.lib(), tr.model._handle
    for mode in (1, 2):
        _lib.check(lib.mmhip_gemm_timing(h, mode, 1, None, None, None))
        tr.train_step(ids.cuda(), mask.cuda(), pixels, onehot, None, 1e-3, 0.00025, 1)
        buf = C.create_string_buffer(1 << 14)
        _lib.check(lib.mmhip_gemm_timing_by_shape(h, buf, len(buf)))
        ms, n, fl = C.c_double(), C.c_uint64(), C.c_double()
        _lib.check(lib.mmhip_gemm_timing(h, 0, 1, C.byref(ms), C.byref(n), C.byref(fl)))
        # forward: patch embedding + 4 per image layer + 4 per text layer; backward: 4 activation-gradient GEMMs per text layer
        assert n.value == 1 + 4 * Lv + 4 * Lt + 4 * Lt, n.value
        Mt, Mv, P = 8 * 64, 8 * 197, 197
        per_layer = lambda M: 2.0 * M * (3 * H * H + H * H + 2 * H * I)
        # the last text layer runs QKV (forward) and d QKV (backward) on all rows, everything else on the 8 CLS rows
        last = 2 * (2.0 * Mt * 3 * H * H) + 2 * (2.0 * 8 * (H * H + 2 * H * I))
        want = 2.0 * 8 * (P - 1) * H * 768 + Lv * per_layer(Mv) + 2 * (Lt - 1) * per_layer(Mt) + last
        assert abs(fl.value - want) < 1e-6 * want, (fl.value, want)
        assert ms.value > 0
        rows = [l.split() for l in buf.value.decode().strip().splitlines()[1:]]
        # columns: M N K flags tile cus n avg_us total_ms TFLOP/s TF/CU-share
        assert sum(int(r[6]) for r in rows) == n.value and all(float(r[9]) > 0 and 0 < int(r[5]) <= 256 for r in rows)
        assert {(int(r[0]), int(r[1]), int(r[2])) for r in rows} >= {(Mv, 3 * H, H), (Mv, H, I), (Mt, 3 * H, H), (Mt, I, H)}
