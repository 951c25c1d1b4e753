"""-m gpu: early-fusion LXMERT (BASELINE config 5; round 4: the native engine csrc/early.hip) against the vectors of
the reference's own `mm_early.Lxmert` module (tests/golden/lxmert_small.npz, make_lxmert_golden.py) and the oracle."""
import ast
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

import smtc_amd  # noqa: E402,F401
from smtc_amd.mm_early import Lxmert, MMEarly_Model  # noqa: E402
from oracle import lxmert_oracle as L  # noqa: E402

# measured (MI355X, round 2): bf16x3 outputs <= 2.4e-5, losses <= 3e-7, gradients <= 2e-5; f16 outputs <= 2.5e-3, gradient rows <= 8e-2
# (the small attention-query gradients), gradient norms <= 1.2e-3; bf16 outputs <= 3.3e-2, rows <= 0.145, norms <= 1.05e-2 (two builds:
# the realisation of the bf16 rounding noise moves with where the kernels round).  Bands = about twice the measurement; bf16x3 = north_star's 1e-3.
TOL_OUT = {"bf16x3": 1e-3, "f16": 5e-3, "bf16": 6e-2}
TOL_GRAD = {"bf16x3": 1e-3, "f16": 0.16, "bf16": 0.3}
TOL_NORM = {"bf16x3": 1e-3, "f16": 2.5e-3, "bf16": 2.5e-2}


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def build(z, dtype, p=0.0):
    c = L.LxmertConfig(**ast.literal_eval(str(z["cfg"])))
    arch = dict(l_layers=c.l_layers, r_layers=c.r_layers, x_layers=c.x_layers, vocab=c.vocab, max_pos=c.max_pos, type_vocab=c.type_vocab,
                p_hidden=p, p_attn=p)
    m = Lxmert(None, c.num_labels, dropout=p, arch=arch, dtype=dtype)
    P = L.make_params(c, int(z["seed_w"]))
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not missing and not unexpected, (missing[:3], unexpected[:3])
    return c, m, P


@pytest.mark.parametrize("dtype", ["bf16x3", "f16", "bf16"])
def test_forward_matches_reference_golden(dtype):
    z = np.load(os.path.join(GOLD, "lxmert_small.npz"), allow_pickle=False)
    c, m, _ = build(z, dtype)
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, int(z["B"]), int(z["T"]), int(z["seed_x"]))
    tim = (torch.from_numpy(z["tim_ids"]), torch.from_numpy(z["tim_mask"]), torch.zeros_like(ids))
    m.eval()
    with torch.no_grad():
        out, et, ev, otim = m(ids, mask, tt, feats, boxes, tim_inputs=tim)
        lpt = m.get_logits_per_text(et, ev)
    errs = {k: rel(v, z[k]) for k, v in (("out_cls", out), ("emb_t", et), ("emb_v", ev), ("out_tim", otim), ("logits_per_text", lpt))}
    print("lxmert fwd", dtype, errs)
    for k, e in errs.items():
        assert e < TOL_OUT[dtype], (k, e)


@pytest.mark.parametrize("dtype", ["bf16x3", "f16", "bf16"])
def test_loss_mixes_and_gradients_match_reference_golden(dtype):
    """train mode with dropout 0: the three loss mixes of mm_early.py:366-379; for ITC + ITM the gradients of the watched parameters
    (first rows + Frobenius norm), `grad is None` for the pooler, zero rows for position 0 / token type 0 (padding_idx=0)"""
    z = np.load(os.path.join(GOLD, "lxmert_small.npz"), allow_pickle=False)
    c, m, _ = build(z, dtype)
    cfg = types.SimpleNamespace(batch_size=4, num_labels=c.num_labels, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=20, dropout=0.0)
    tr = MMEarly_Model.__new__(MMEarly_Model)
    tr.__dict__.update(batch_size=4, num_labels=c.num_labels, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=20,
                       model=m, device=m.device_, _opt=None)
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, int(z["B"]), int(z["T"]), int(z["seed_x"]))
    tim = (torch.from_numpy(z["tim_ids"]), torch.from_numpy(z["tim_mask"]), torch.zeros_like(ids))
    w, lbl = torch.from_numpy(z["class_w"]), torch.from_numpy(z["lbl_tim"]).cuda()
    m.train()
    for tag, itc, itm in (("cls", False, False), ("itc", True, False), ("itcitm", True, True)):
        tr.use_clip_loss, tr.use_tim_loss = itc, itm
        m.zero_grad()
        out, et, ev, otim = m(ids, mask, tt, feats, boxes, tim_inputs=tim if itm else None)
        loss = tr.loss(out, onehot, w, et, ev, otim, lbl)
        e = abs(loss.item() - float(z["loss." + tag])) / float(z["loss." + tag])
        print("lxmert loss", dtype, tag, e)
        assert e < {"bf16x3": 1e-4, "f16": 2e-3, "bf16": 1e-2}[dtype], (tag, e)
    loss.backward()
    m.finish_backward()
    named = dict(m.named_parameters())
    errs = {}
    for k in [f[5:] for f in z.files if f.startswith("grad.")]:
        g = named[k].grad
        got = g[:4] if g.dim() == 2 else g
        errs[k] = (rel(got, z["grad." + k]), abs(g.double().norm().item() - float(z["gradnorm." + k])) / float(z["gradnorm." + k]))
    print("lxmert grads", dtype, {k: (round(a, 5), round(b, 5)) for k, (a, b) in errs.items()})
    # the vision stream of the LAST cross-modality layer reaches the loss only through max(x_v) over the 36 boxes (mm_early.py:142): a 16-bit
    # rounding can move an arg-max to another box and with it the rows that receive gradient -- element-wise comparison of those (tiny)
    # gradients is meaningless in the 16-bit modes (measured 0.02 ... 0.65 for the same build), their norm is stable
    fragile = {"model.encoder.x_layers.%d.visn_output.dense.weight" % (c.x_layers - 1)} if dtype != "bf16x3" else set()
    for k, (a, b) in errs.items():
        # a: max-norm error over the first rows (small-gradient matrices such as the attention queries are noisy in 16 bits),
        # b: error of the Frobenius norm of the whole gradient
        assert (k in fragile or a < TOL_GRAD[dtype]) and b < TOL_NORM[dtype], (k, a, b)
    # the pooler is off the path: its slices of the flat gradient stay zero and AdamW's ranges leave it out (torch: grad is None)
    assert not named["model.pooler.dense.weight"].grad.any() and not named["model.pooler.dense.bias"].grad.any()
    covered = lambda name: any(b <= m._offs[name] < e for b, e in m.grad_ranges(True, True))
    assert not covered("model.pooler.dense.weight") and not covered("model.pooler.dense.bias") and covered("linear_tim.weight") and covered("logit_scale")
    assert not any(b <= m._offs["linear_tim.weight"] < e for b, e in m.grad_ranges(True, False))
    assert not named["model.embeddings.position_embeddings.weight"].grad[0].any()


def test_train_steps_reduce_the_loss_and_leave_the_pooler_alone():
    cfg = types.SimpleNamespace(batch_size=8, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=24, dropout=0.05)
    arch = dict(l_layers=2, r_layers=1, x_layers=1, vocab=500, max_pos=64)
    tr = MMEarly_Model(cfg, "lxmert", arch=arch, seed=3)
    c = L.LxmertConfig(l_layers=2, r_layers=1, x_layers=1, vocab=500, max_pos=64, num_labels=3)
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, 8, 24, 5)
    pool0 = tr.model._P("model.pooler.dense.weight").detach().clone()
    np.random.seed(30)
    losses = [float(tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 2e-4, 0.00025, s)) for s in range(1, 9)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert torch.equal(tr.model._P("model.pooler.dense.weight").detach(), pool0)       # never receives a gradient: AdamW skips it
    res = tr.eval([{"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "token_type_ids": tt.unsqueeze(1), "features": feats,
                    "normalized_boxes": boxes, "labels": onehot, "data_id": torch.arange(8)}])
    assert res["predictions"].shape == (8,) and np.isfinite(res["loss"]) and np.array_equal(res["labels"], onehot.argmax(1).numpy())


def test_adamw_ranges_are_merged_in_address_order():
    """the ranges AdamW runs over cover exactly the parameters that receive gradients (never the pooler; linear_tim / logit_scale by flag) and
    are merged where adjacent in the flat buffer: a handful of launches, not one per parameter group"""
    m = Lxmert(None, 3, dropout=0.0, arch=dict(l_layers=2, r_layers=1, x_layers=1, vocab=300, max_pos=64), dtype="bf16", seed=1)
    for itc, itm in ((False, False), (True, True)):
        ranges = m.grad_ranges(itc, itm)
        assert len(ranges) <= 6 and all(b < e for b, e in ranges) and all(ranges[i][1] <= ranges[i + 1][0] for i in range(len(ranges) - 1))
        covered = torch.zeros(m._flat.numel(), dtype=torch.bool)
        for b, e in ranges:
            covered[b:e] = True
        for name, shape in m._shapes.items():
            o, n = m._offs[name], int(np.prod(shape)) if shape else 1
            want = not (name.startswith("model.pooler.") or (name.startswith("linear_tim.") and not itm) or (name == "logit_scale" and not itc))
            assert bool(covered[o:o + n].all()) == want and (want or not bool(covered[o:o + n].any())), name


def test_cross_attention_with_text_longer_than_boxes_matches_the_oracle():
    """the orientation of BASELINE config 5 (48 tokens > 36 boxes; the golden vectors have the text shorter): the native engine -- cross-attention
    blocks projecting straight into the packed tensor, row copies for the shorter stream, the ONE cross module's gradient accumulated from both
    directions -- against the oracle (oracle/lxmert_oracle.py, pinned by the reference's own module): outputs and every gradient, parity mode"""
    c = L.LxmertConfig(l_layers=1, r_layers=1, x_layers=2, vocab=300, max_pos=64, num_labels=3)
    arch = dict(l_layers=1, r_layers=1, x_layers=2, vocab=300, max_pos=64, p_hidden=0.0, p_attn=0.0)
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, 5, 48, 11)
    m = Lxmert(None, 3, dropout=0.0, arch=arch, dtype="bf16x3", seed=4)
    P = L.make_params(c, 9)
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not missing and not unexpected
    m.train()
    m.zero_grad()
    out, et, ev, _ = m(ids, mask, tt, feats, boxes)
    (out.square().sum() + ev.square().sum()).backward()
    torch.cuda.synchronize()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    r_out, r_et, r_ev, _ = L.early_forward(Pg, ids, mask, tt, feats, boxes, c)
    (r_out.square().sum() + r_ev.square().sum()).backward()
    for a_, b_ in ((out, r_out), (et, r_et), (ev, r_ev)):
        assert rel(a_.detach(), b_.detach()) < 1e-4
    named = dict(m.named_parameters())
    worst = 0.0
    for k, q in Pg.items():
        if q.grad is None or k.startswith("model.pooler") or k == "logit_scale" or k.startswith("linear_tim") or k.endswith("key.bias"):
            continue                              # (a key bias shifts every score of a row alike: its true gradient is zero, what is left is round-off)
        g = named[k].grad.detach().cpu()
        if "embeddings.position_embeddings" in k or "token_type_embeddings" in k:
            assert not g[0].any()                     # padding_idx = 0: no gradient for position 0 / token type 0
            continue
        err = (g - q.grad).norm().item() / max(q.grad.norm().item(), 1e-30)
        worst = max(worst, err)
        assert err < 1e-3, (k, err)
    print("early cross orientation: worst gradient error", worst)


@pytest.mark.parametrize("Sq,Sk", [(36, 48), (48, 36), (40, 40), (128, 36), (36, 128), (70, 3)])
def test_cross_attention_block_takes_no_padding_passes(Sq, Sk):
    """round 5: the cross-attention block works on COMPACT tensors -- the projections write Mq = posts * Sq query rows and Mc = posts * Sk key / value
    rows, the attention kernels are told the two row pitches and lengths (AttnArgs::q_rps / kv_rps / ctx_rps, Sq_live / Sk_live) and skip key tiles
    past Sk -- no clear, no row remap, no row copies (round 4: S-row blocks with a remapping GEMM epilogue; round 3: padded copies).  Every work
    buffer is poisoned with NaN before the call: a single read of a row that no projection wrote would surface in y / dxq / dxc / the
    weight-gradient operands.  Reference: the same block in torch fp32 on the same bf16-rounded operands (HF LxmertCrossAttentionLayer arithmetic)."""
    from smtc_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Sq * 100 + Sk)
    posts, heads, H, S = 3, 2, 128, max(Sq, Sk)
    rb = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16)
    xq, xc = rb(posts * Sq, H), rb(posts * Sk, H)
    wqkv, wo = rb(3 * H, H, sc=0.08), rb(H, H, sc=0.08)
    bqkv, bo = torch.randn(3 * H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    live = torch.ones(posts, Sk)
    live[1, Sk - 5:] = 0                                              # one post with masked keys
    keybias = torch.full((posts, S), float("-inf"))
    keybias[:, :Sk] = torch.where(live > 0, 0.0, float("-inf"))
    dy = rb(posts * Sq, H)
    # ---- reference (fp32 on the rounded operands)
    Xq, Xc, W, Wo = (t.float().requires_grad_(True) for t in (xq, xc, wqkv, wo))
    q = (Xq @ W[:H].T + bqkv[:H]).view(posts, Sq, heads, 64).transpose(1, 2)
    k = (Xc @ W[H:2 * H].T + bqkv[H:2 * H]).view(posts, Sk, heads, 64).transpose(1, 2)
    v = (Xc @ W[2 * H:].T + bqkv[2 * H:]).view(posts, Sk, heads, 64).transpose(1, 2)
    pr = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + keybias[:, None, None, :Sk], -1)
    ctx = (pr @ v).transpose(1, 2).reshape(posts * Sq, H)
    y_ref = torch.nn.functional.layer_norm(ctx @ Wo.T + bo + Xq, (H,), gamma, beta, 1e-12)
    (y_ref * dy.float()).sum().backward()
    # ---- the operator, every work buffer NaN
    cu = lambda t: t.to(dev).contiguous()
    nan16 = lambda *sh: torch.full(sh, float("nan"), dtype=torch.bfloat16, device=dev)
    nan32 = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float32, device=dev)
    d_xq, d_xc, d_w, d_wo, d_wT, d_woT = cu(xq), cu(xc), cu(wqkv), cu(wo), cu(wqkv.T), cu(wo.T)
    d_bqkv, d_bo, d_gamma, d_beta, d_kb, d_dy = cu(bqkv), cu(bo), cu(gamma), cu(beta), cu(keybias), cu(dy)
    qkv, att, lse = nan16(posts * S, 3 * H), nan16(posts * S, H), nan32(posts, heads, S)
    attq, pre, mean, rstd, y = nan16(posts * Sq, H), nan16(posts * Sq, H), nan32(posts * Sq), nan32(posts * Sq), nan16(posts * Sq, H)
    P_, st = _lib.ptr, _lib.stream_ptr()
    _lib.check(lib.mmhip_op_cross_att_block_fwd(_lib.BF16, P_(d_xq), P_(d_xc), P_(d_kb), P_(d_w), P_(d_bqkv), P_(d_wo), P_(d_bo), P_(d_gamma), P_(d_beta), 1e-12,
                                                posts, Sq, Sk, heads, 0.0, 0.0, 1, P_(qkv), P_(att), P_(lse), None, None, P_(attq), P_(pre), P_(mean), P_(rstd),
                                                P_(y), st))
    dgamma, dbeta = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    dpre, datt, dqkv = nan16(posts * Sq, H), nan16(posts * S, H), nan16(posts * S, 3 * H)
    dq, dkv, dxq, dxc = nan16(posts * Sq, H), nan16(posts * Sk, 2 * H), nan16(posts * Sq, H), nan16(posts * Sk, H)
    _lib.check(lib.mmhip_op_cross_att_block_bwd(_lib.BF16, P_(d_dy), P_(d_kb), P_(d_wT), P_(d_woT), P_(d_gamma), posts, Sq, Sk, heads, 0.0, 0.0, 1, P_(qkv),
                                                P_(att), P_(lse), P_(pre), P_(mean), P_(rstd), P_(dgamma), P_(dbeta), P_(dpre), P_(dpre), None, P_(datt),
                                                P_(dqkv), P_(dq), P_(dkv), P_(dxq), P_(dxc), st))
    torch.cuda.synchronize()
    for name, got, ref, tol in (("y", y, y_ref, 3e-2), ("dxq", dxq, Xq.grad, 4e-2), ("dxc", dxc, Xc.grad, 4e-2)):
        assert torch.isfinite(got.float()).all(), name
        assert rel(got, ref.detach()) < tol, (name, rel(got, ref.detach()))
    # the operands the weight-gradient GEMMs read, where the block left them: rows [0, Mq) of dqkv's Q columns, rows [0, Mc) of its K | V columns, att [Mq, H];
    # the scratch tensors of the earlier forms (attq, dq, dkv) must not have been touched
    src_q, src_kv, src_att = dqkv[:posts * Sq, :H], dqkv[:posts * Sk, H:], att[:posts * Sq]
    assert torch.isnan(attq.float()).all() and torch.isnan(dq.float()).all() and torch.isnan(dkv.float()).all()
    assert torch.isfinite(src_q.float()).all() and torch.isfinite(src_kv.float()).all() and torch.isfinite(src_att.float()).all()
    dWq = src_q.float().T @ d_xq.float()
    assert rel(dWq, W.grad[:H]) < 4e-2
    dWkv = src_kv.float().T @ d_xc.float()
    assert rel(dWkv, W.grad[H:]) < 4e-2
    assert rel(src_att, ctx.detach()) < 3e-2


@pytest.mark.parametrize("Sq,Sk", [(128, 36), (36, 128), (20, 100), (33, 1)])
def test_cross_attention_16bit_and_parity_forms_agree_with_dropout_on(Sq, Sk):
    """the 16-bit kernels against the parity-mode (fp32 tensors, three bf16 products) kernels of the same compact cross-attention operator with dropout ON
    in both places: the two share the element-index convention of the masks (indices of the S x S layout, S = max(Sq, Sk), whatever the row pitch), so
    they must drop the same attention probabilities and the same hidden units and agree to bf16 accuracy -- outputs and both input gradients"""
    from smtc_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Sq * 131 + Sk)
    posts, heads, H, S = 3, 2, 128, max(Sq, Sk)
    r32 = lambda *sh, sc=1.0: (torch.randn(*sh, generator=g) * sc).to(torch.bfloat16).float()      # bf16-representable values in both forms
    xq, xc, dy = r32(posts * Sq, H), r32(posts * Sk, H), r32(posts * Sq, H)
    wqkv, wo = r32(3 * H, H, sc=0.08), r32(H, H, sc=0.08)
    bqkv, bo = torch.randn(3 * H, generator=g) * 0.1, torch.randn(H, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    keybias = torch.full((posts, S), float("-inf"))
    keybias[:, :Sk] = 0.0
    if Sk > 4:
        keybias[1, Sk - 3:Sk] = float("-inf")
    res = {}
    for name, code, tdt in (("bf16", _lib.BF16, torch.bfloat16), ("f32", _lib.F32, torch.float32)):
        cu = lambda t: t.to(tdt).to(dev).contiguous()
        f = lambda t: t.to(dev).contiguous()
        z = lambda *sh: torch.zeros(sh, dtype=tdt, device=dev)
        z32 = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
        d = dict(xq=cu(xq), xc=cu(xc), w=cu(wqkv), wo=cu(wo), wT=cu(wqkv.T), woT=cu(wo.T), dy=cu(dy))
        qkv, att, lse = z(posts * S, 3 * H), z(posts * S, H), z32(posts, heads, S)
        tq, tkv, attq = z(posts * Sq, H), z(posts * Sk, 2 * H), z(posts * Sq, H)
        pre, mean, rstd, y = z(posts * Sq, H), z32(posts * Sq), z32(posts * Sq), z(posts * Sq, H)
        P_, st = _lib.ptr, _lib.stream_ptr()
        kb, bq, bo_, ga, be = f(keybias), f(bqkv), f(bo), f(gamma), f(beta)
        _lib.check(lib.mmhip_op_cross_att_block_fwd(code, P_(d["xq"]), P_(d["xc"]), P_(kb), P_(d["w"]), P_(bq), P_(d["wo"]), P_(bo_), P_(ga), P_(be), 1e-12, posts,
                                                    Sq, Sk, heads, 0.1, 0.1, 77, P_(qkv), P_(att), P_(lse), P_(tq), P_(tkv), P_(attq), P_(pre), P_(mean), P_(rstd), P_(y), st))
        dgamma, dbeta = z32(H), z32(H)
        dpre, dd, dattq, datt, dqkv = z(posts * Sq, H), z(posts * Sq, H), z(posts * Sq, H), z(posts * S, H), z(posts * S, 3 * H)
        dq, dkv, dxq, dxc = z(posts * Sq, H), z(posts * Sk, 2 * H), z(posts * Sq, H), z(posts * Sk, H)
        _lib.check(lib.mmhip_op_cross_att_block_bwd(code, P_(d["dy"]), P_(kb), P_(d["wT"]), P_(d["woT"]), P_(ga), posts, Sq, Sk, heads, 0.1, 0.1, 77, P_(qkv), P_(att),
                                                    P_(lse), P_(pre), P_(mean), P_(rstd), P_(dgamma), P_(dbeta), P_(dpre), P_(dd), P_(dattq), P_(datt), P_(dqkv), P_(dq),
                                                    P_(dkv), P_(dxq), P_(dxc), st))
        torch.cuda.synchronize()
        res[name] = (y.float().cpu(), dxq.float().cpu(), dxc.float().cpu(), dgamma.cpu())
    for a_, b_, what in zip(res["bf16"], res["f32"], ("y", "dxq", "dxc", "dgamma")):
        assert torch.isfinite(a_).all() and rel(a_, b_) < 5e-2, (what, rel(a_, b_))
    # the same hidden units are dropped in both forms: pre = dropout(att Wo^T + bo) + xq, so y differs from LN(xq)-like rows in the same places
    # (a mismatch of the mask indices would show as O(1) differences above, far outside the bf16 band)


def test_native_step_equals_the_autograd_path():
    """mmhip_early_train_step (one native call: forward, fused loss mix incl. the ITC similarity, backward, AdamW, refresh) against the
    reference-style path -- Lxmert.forward, MMEarly_Model.loss in torch, loss.backward(), AdamW over grad_ranges -- on the same weights, batch
    and ITM draw (dropout off), for the three loss mixes.  AdamW's first step is update = lr g / (|g| + eps): with the default eps = 1e-8 an
    attention-query gradient of 1e-8 turns fp32 round-off of the loss gradient into +- lr, so the comparison runs with eps = 1e-2 (update
    proportional to the gradient): the parameter DELTAS of the two paths must agree to 1e-4 of the largest delta."""
    from smtc_amd import _lib
    c = L.LxmertConfig(l_layers=2, r_layers=1, x_layers=1, vocab=400, max_pos=64, num_labels=3)
    arch = dict(l_layers=2, r_layers=1, x_layers=1, vocab=400, max_pos=64, p_hidden=0.0, p_attn=0.0)
    ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, 6, 24, 5)
    w = torch.tensor([0.7, 1.6, 0.9])
    EPS = 1e-2
    for itc, itm in ((False, False), (True, False), (True, True)):
        cfg = types.SimpleNamespace(batch_size=6, num_labels=3, use_clip_loss=itc, beta_itc=0.1, use_tim_loss=itm, beta_itm=0.1, max_length=24, dropout=0.0)
        a, b = MMEarly_Model(cfg, "lxmert", arch=arch, seed=3, dtype="bf16x3"), MMEarly_Model(cfg, "lxmert", arch=arch, seed=3, dtype="bf16x3")
        assert torch.equal(a.model._flat, b.model._flat)
        p0 = a.model._flat.clone()
        a.adam_eps = EPS
        np.random.seed(77)
        la = float(a.train_step(ids, mask, tt, feats, boxes, onehot, w, 1e-3, 0.0, 1))
        np.random.seed(77)
        m = b.model
        m.train(); m.zero_grad()
        tim, lbl = None, None
        if itm:
            t_ids, t_mask, t_tt, lbl = b.prepare_itm_inputs(ids.cuda(), mask.cuda(), tt.cuda())
            tim = (t_ids, t_mask, t_tt)
        out, et, ev, otim = m(ids, mask, tt, feats, boxes, tim_inputs=tim)
        loss = b.loss(out, onehot, w, et, ev, otim, lbl)
        loss.backward()
        mom = (torch.zeros_like(m._flat), torch.zeros_like(m._flat))
        at = lambda t, el: t.data_ptr() + el * 4
        for rb, re in m.grad_ranges(itc, itm):
            _lib.check(_lib.lib().mmhip_adamw(at(m._flat, rb), at(m._flat_grad, rb), at(mom[0], rb), at(mom[1], rb), re - rb, 1e-3, 0.9, 0.999, EPS, 0.0, 1, 1.0, 1,
                                              _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert abs(la - loss.item()) < 1e-5 * abs(loss.item()), (itc, itm, la, loss.item())
        da, db = a.model._flat - p0, m._flat - p0
        scale = db.abs().max().item()
        d = (da - db).abs().max().item()
        print("native vs autograd", itc, itm, la, loss.item(), d, scale)
        assert scale > 1e-6 and d < 1e-4 * scale + 2.5e-7, (itc, itm, d, scale)      # (+ two ulps of a parameter of magnitude 1: LayerNorm weights)
        for name in ("model.pooler.dense.weight",) + (() if itm else ("linear_tim.weight",)) + (() if itc else ("logit_scale",)):      # off the path: untouched on both
            o, n_ = m._offs[name], int(np.prod(m._shapes[name])) if m._shapes[name] else 1
            assert not da[o:o + n_].any() and not db[o:o + n_].any(), name


def test_full_size_step_properties():
    """BASELINE config 5 at full size (9 + 5 + 5 layers, bs 32, T = 128, 36 x 2048 ROI features, ITC + ITM, dropout on): finite decreasing loss over
    a few steps, the pooler untouched, replicas of the same seed bit-identical in the deterministic kernels' outputs (forward), every gradient range
    written (no all-zero layer) -- size-independent properties where the oracle would take minutes"""
    cfg = types.SimpleNamespace(batch_size=32, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=128, dropout=0.05)
    tr = MMEarly_Model(cfg, "lxmert", dtype="bf16", seed=0)
    m = tr.model
    g = torch.Generator().manual_seed(1)
    B, T, NB = 32, 128, 36
    ids = torch.randint(1, 30522, (B, T), generator=g)
    mask = (torch.arange(T)[None, :] < torch.randint(8, T + 1, (B, 1), generator=g)).long()
    ids = ids * mask
    tt = torch.zeros_like(ids)
    feats, boxes = torch.rand(B, NB, 2048, generator=g) * 2, torch.rand(B, NB, 4, generator=g)
    onehot = torch.nn.functional.one_hot(torch.randint(0, 3, (B,), generator=g), 3)
    pool0 = m._P("model.pooler.dense.weight").detach().clone()
    np.random.seed(30)
    losses = [float(tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 5e-5, 0.00025, s)) for s in range(1, 7)]
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0], losses
    assert torch.equal(m._P("model.pooler.dense.weight").detach(), pool0) and torch.isfinite(m._flat).all()
    m.eval()
    with torch.no_grad():
        o1 = m(ids, mask, tt, feats, boxes)
        o2 = m(ids, mask, tt, feats, boxes)
    assert all(torch.equal(x, y) for x, y in zip(o1[:3], o2[:3]))
    m.train(); m.zero_grad()
    out, et, ev, _ = m(ids, mask, tt, feats, boxes)
    (out.square().sum() + ev.square().sum()).backward()
    torch.cuda.synchronize()
    for st, (b, e) in enumerate(m._stage_ranges):
        assert m._flat_grad[b:e].abs().max().item() > 0, st


def test_cli_synthetic_run_writes_the_reference_files(tmp_path):
    """run_mm_early.py mirror on synthetic posts: metrics CSVs (reference layout), checkpoint with the reference's keys, predictions"""
    import subprocess, sys
    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = str(tmp_path) + "/"
    r = subprocess.run([sys.executable, "-m", "smtc_amd.run_mm_early", "--model", "lxmert", "--task", "3", "--epochs", "1", "--use_clip_loss", "--use_tim_loss",
                        "--synthetic", "--n_synthetic", "32", "--batch_size", "8", "--arch_layers", "1", "--results_dir", res, "--save_model", "--evaltest"],
                       cwd=root, env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = res + "lxmert_task3_seed30_itc0.1itm0.1_"
    mv = pd.read_csv(stem + "metrics_val.csv")
    assert list(mv.columns) == ["metric", "epoch-1"] and np.isfinite(mv["epoch-1"]).all()
    assert list(pd.read_csv(stem + "preds.csv").columns) == ["data_id", "label", "prediction"]
    sd = torch.load(stem + "net.pth", map_location="cpu")
    assert "model.encoder.x_layers.0.visual_attention.att.query.weight" in sd and "linear_fusion.weight" in sd and "logit_scale" in sd


def test_cli_data_key_run_reads_roi_feature_files(tmp_path):
    """run_mm_early.py on the data-key path: data key -> prepare_data -> Lxmert_Dataset over <task>_img_feats/{features,boxes} files
    (reference models/datasets.py:255-301, mm_early.py:228-258) -> training, validation, test files"""
    import subprocess, sys
    import pandas as pd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import make_dummy_task
    run_dir = make_dummy_task.main(str(tmp_path), 60, 1)
    make_dummy_task.add_roi(str(tmp_path), 60)
    res = str(tmp_path) + "/res/"
    r = subprocess.run([sys.executable, "-m", "smtc_amd.run_mm_early", "--model", "lxmert", "--task", "2", "--epochs", "1", "--use_clip_loss",
                        "--batch_size", "8", "--arch_layers", "1", "--results_dir", res, "--evaltest", "--num_workers", "2"],
                       cwd=run_dir, env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    stem = res + "lxmert_task2_seed30_itc0.1_"
    mv = pd.read_csv(stem + "metrics_val.csv")
    assert list(mv.columns) == ["metric", "epoch-1"] and np.isfinite(mv["epoch-1"]).all()
    preds = pd.read_csv(stem + "preds.csv")
    key = pd.read_csv(os.path.join(run_dir, "..", "data", "data_key_imgtxt_random.csv"))
    assert sorted(preds.data_id.tolist()) == sorted(key[key.split == "test"].tweet_id.tolist())


DP_EARLY = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.environ["ROOT"])
import smtc_amd
from smtc_amd import dist as mmdist
from smtc_amd.mm_early import MMEarly_Model
from oracle import lxmert_oracle as L
os.environ["LOCAL_RANK"] = "0"
mmdist.init_from_env(backend="gloo")
rank, world = mmdist.rank(), mmdist.world_size()
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=False, beta_itm=0.1, max_length=20, dropout=0.0)
arch = dict(l_layers=1, r_layers=1, x_layers=1, vocab=300, max_pos=64, p_hidden=0.0, p_attn=0.0)
c = L.LxmertConfig(l_layers=1, r_layers=1, x_layers=1, vocab=300, max_pos=64, num_labels=3)
ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, 8, 20, 5)
tr = MMEarly_Model(cfg, "lxmert", arch=arch, seed=5, dtype="bf16x3")
tr.adam_eps = 1e-2          # update ~ gradient: with the default 1e-8 an attention-query gradient of 1e-8 turns fp32 round-off into +- lr
p_init = tr.model._flat.cpu().clone()
sl = slice(rank * 4, rank * 4 + 4)
tr.train_step(ids[sl], mask[sl], tt[sl], feats[sl], boxes[sl], onehot[sl], None, 1e-3, 0.00025, 1)
torch.save(tr.model._flat.cpu(), os.environ["OUT"] + f"/p{rank}.pt")
torch.distributed.barrier()
if rank == 0:
    ref = MMEarly_Model(cfg, "lxmert", arch=arch, seed=5, dtype="bf16x3")
    m = ref.model
    m.train(); g = torch.zeros_like(m._flat_grad)
    for r in range(world):
        s2 = slice(r * 4, r * 4 + 4)
        m.zero_grad()
        out, et, ev, _ = m(ids[s2], mask[s2], tt[s2], feats[s2], boxes[s2])
        ref.loss(out, onehot[s2], None, et, ev, None, None).backward()
        m.finish_backward()
        g += m._flat_grad
    m._flat_grad.copy_(g / world)
    from smtc_amd import _lib
    import ctypes as C
    mom = (torch.zeros_like(m._flat), torch.zeros_like(m._flat))
    at = lambda t, el: C.c_void_p(t.data_ptr() + el * 4)
    for b, e in m.grad_ranges(True, False):
        _lib.check(_lib.lib().mmhip_adamw(at(m._flat, b), at(m._flat_grad, b), at(mom[0], b), at(mom[1], b), e - b, 1e-3, 0.9, 0.999, 1e-2, 0.00025, 1, 1.0, 1, _lib.stream_ptr()))
    torch.cuda.synchronize()
    p0, p1 = torch.load(os.environ["OUT"] + "/p0.pt"), torch.load(os.environ["OUT"] + "/p1.pt")
    scale = (m._flat.cpu() - p_init).abs().max().item()
    print("DP_EARLY", max(0.0, (p0 - m._flat.cpu()).abs().max().item() - 2.5e-7) / scale, torch.equal(p0, p1))      # (- two ulps of a parameter of magnitude 1)
torch.distributed.destroy_process_group()
'''


def test_two_rank_step_equals_averaged_single_process(tmp_path):
    """data parallelism of the early-fusion step: two ranks on the one card (gloo), each on its half of the batch, one all-reduce of the flat
    gradient -> identical replicas, equal to one process stepping on the averaged gradients"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_early.py"
    script.write_text(DP_EARLY)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        str(29300 + os.getpid() % 300), str(script)], cwd=root,
                       env=dict(os.environ, PYTHONPATH=root, ROOT=root, OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DP_EARLY")][0].split()
    assert float(line[1]) < 1e-4 and line[2] == "True", line          # parameter deltas agree to 1e-4 of the largest delta; replicas bit-identical


RCCL_EARLY = r'''
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.environ["ROOT"])
import smtc_amd
from smtc_amd.mm_early import MMEarly_Model
from oracle import lxmert_oracle as L
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:" + os.environ["PORT"], rank=0, world_size=1)
cfg = types.SimpleNamespace(batch_size=4, num_labels=3, use_clip_loss=True, beta_itc=0.1, use_tim_loss=True, beta_itm=0.1, max_length=20, dropout=0.0)
arch = dict(l_layers=1, r_layers=1, x_layers=1, vocab=300, max_pos=64, p_hidden=0.0, p_attn=0.0)
c = L.LxmertConfig(l_layers=1, r_layers=1, x_layers=1, vocab=300, max_pos=64, num_labels=3)
ids, mask, tt, feats, boxes, onehot = L.synthetic_batch(c, 4, 20, 5)
out = {}
for name, force in (("rccl", "1"), ("plain", "0")):
    os.environ["MMHIP_FORCE_EXCHANGE"] = force
    tr = MMEarly_Model(cfg, "lxmert", arch=arch, seed=5, dtype="bf16x3")
    tr.adam_eps = 1e-2
    p0 = tr.model._flat.clone()
    np.random.seed(30)
    losses = [float(tr.train_step(ids, mask, tt, feats, boxes, onehot, None, 1e-3, 0.00025, s)) for s in (1, 2)]
    torch.cuda.synchronize()
    out[name] = (tr.model._flat.clone() - p0, losses, int(getattr(tr.model, "_last_exchange_bytes", 0)))
d = (out["rccl"][0] - out["plain"][0]).abs().max().item() / out["plain"][0].abs().max().item()
print("RCCL_EARLY", d, out["rccl"][2] > 0, torch.distributed.get_backend(), abs(out["rccl"][1][1] - out["plain"][1][1]))
torch.distributed.destroy_process_group()
'''


def test_rccl_call_pattern_of_the_early_step_at_world_size_one(tmp_path):
    """the staged exchange of the early-fusion step issued through RCCL itself (backend "nccl", one rank, exchange forced): the engine calls back at
    every backward stage boundary, the stage ranges leave as bucketed all-reduces of slices of the flat gradient, AdamW waits for them -- the
    parameter deltas of two steps equal those of the step without any collective (atomics in the shared cross-modality gradients: 1e-4 of the
    largest delta, not bits).  Two ranks over gloo pin the arithmetic across ranks (test_two_rank_step_equals_averaged_single_process)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_early.py"
    script.write_text(RCCL_EARLY)
    r = subprocess.run([sys.executable, str(script)], cwd=root, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=root, ROOT=root, PORT=str(29950 + os.getpid() % 40), HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_EARLY")][0].split()
    assert float(line[1]) < 1e-4 and line[2] == "True" and line[3] == "nccl" and float(line[4]) < 1e-5, line
