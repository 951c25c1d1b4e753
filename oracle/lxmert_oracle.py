"""CPU oracle for the early-fusion LXMERT path (BASELINE config 5)  --  TEST INFRASTRUCTURE ONLY.

Groundwork for the next hot-path row: the HIP path for this configuration is NOT built yet; this file pins what it has to
reproduce.  Plain-PyTorch (fp32, CPU) restatement of
  * the LXMERT encoder            HF:models/lxmert/modeling_lxmert.py  LxmertEmbeddings, LxmertVisualFeatureEncoder,
                                  LxmertAttention / LxmertAttentionOutput, LxmertLayer (language and relational layers),
                                  LxmertXLayer (ONE cross-attention module used in both directions, then self-attention and the
                                  feed-forward block per stream), LxmertEncoder, LxmertPooler, LxmertModel.forward
                                  (transformers==4.25.1 pinned by timrel-env.yml:122, un-vendored; as installed 5.15.0, same mathematics)
  * the reference head            models/mm_early.py:105-172 (class Lxmert): CLS row of the language output -> linear_fusion -> ReLU ->
                                  dropout -> linear; max-pooled language (masked) / vision embeddings for the contrastive loss; the
                                  ITM pass = a second full encoder call on the swapped texts with the same visual features
  * the loss mixing               models/mm_early.py:366-379 with clip_loss (models/utils.py:225-231) on
                                  get_logits_per_text (mm_early.py:165-172)
Only `tests/` may import it.

Parity status: PINNED.  `tests/golden/make_lxmert_golden.py` loads `make_params()` into the reference's own `Lxmert` module
(imported behind the shim of make_golden.py plus stand-in modules for the `lxmert_scripts` package that the reference repository
itself lacks -- it only feeds the offline feature extraction, not this path) and commits inputs + outputs;
`tests/test_oracle_golden.py` checks this restatement against them.
Parameter names are the reference module's `state_dict` keys (`model.*` = LxmertModel).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict  # noqa: F401  (asdict: used by the golden generator)
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .mm_oracle import make_param, clip_loss, cls_loss, itm_loss

Tensor = torch.Tensor


@dataclass
class LxmertConfig:
    hidden: int = 768
    heads: int = 12
    inter: int = 3072
    l_layers: int = 9
    r_layers: int = 5
    x_layers: int = 5
    vocab: int = 30522
    max_pos: int = 512
    type_vocab: int = 2
    feat_dim: int = 2048          # ROI feature width
    pos_dim: int = 4              # normalised boxes
    n_boxes: int = 36
    num_labels: int = 2
    ln_eps: float = 1e-12


def param_shapes(c: LxmertConfig) -> Dict[str, Tuple[int, ...]]:
    H, I = c.hidden, c.inter
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(n, o, i):
        s[n + ".weight"], s[n + ".bias"] = (o, i), (o,)

    def ln(n):
        s[n + ".weight"], s[n + ".bias"] = (H,), (H,)

    def att(n):                         # LxmertAttention + LxmertAttentionOutput
        for p in ("query", "key", "value"):
            lin(f"{n}.{p}", H, H)

    def att_block(n, inner):            # Lxmert{Self,Cross}AttentionLayer
        att(f"{n}.{inner}")
        lin(f"{n}.output.dense", H, H)
        ln(f"{n}.output.LayerNorm")

    def ffn(inter_name, out_name):
        lin(inter_name + ".dense", I, H)
        lin(out_name + ".dense", H, I)
        ln(out_name + ".LayerNorm")

    e = "model.embeddings."
    s[e + "word_embeddings.weight"] = (c.vocab, H)
    s[e + "position_embeddings.weight"] = (c.max_pos, H)
    s[e + "token_type_embeddings.weight"] = (c.type_vocab, H)
    ln(e + "LayerNorm")
    v = "model.encoder.visn_fc."
    lin(v + "visn_fc", H, c.feat_dim); ln(v + "visn_layer_norm")
    lin(v + "box_fc", H, c.pos_dim); ln(v + "box_layer_norm")
    for kind, n in (("layer", c.l_layers), ("r_layers", c.r_layers)):
        for i in range(n):
            b = f"model.encoder.{kind}.{i}."
            att_block(b + "attention", "self")
            ffn(b + "intermediate", b + "output")
    for i in range(c.x_layers):
        b = f"model.encoder.x_layers.{i}."
        att_block(b + "visual_attention", "att")
        att_block(b + "lang_self_att", "self")
        att_block(b + "visn_self_att", "self")
        ffn(b + "lang_inter", b + "lang_output")
        ffn(b + "visn_inter", b + "visn_output")
    lin("model.pooler.dense", H, H)
    lin("linear_fusion", H, H)
    lin("linear", c.num_labels, H)
    lin("linear_tim", 2, H)
    s["logit_scale"] = ()
    return s


def make_params(c: LxmertConfig, seed: int = 0) -> Dict[str, Tensor]:
    return {k: make_param(k, shp, seed) for k, shp in param_shapes(c).items()}


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
def _lin(x, P, n):
    return F.linear(x, P[n + ".weight"], P[n + ".bias"])


def _ln(x, P, n, eps):
    return F.layer_norm(x, (x.shape[-1],), P[n + ".weight"], P[n + ".bias"], eps)


def _attention(P, n, x, ctx, bias, nh):
    """LxmertAttention: queries from x, keys / values from ctx, additive key mask `bias` [B,1,1,S_ctx]"""
    B, Sq, H = x.shape
    d = H // nh
    q = _lin(x, P, n + ".query").view(B, Sq, nh, d).transpose(1, 2)
    k = _lin(ctx, P, n + ".key").view(B, ctx.shape[1], nh, d).transpose(1, 2)
    v = _lin(ctx, P, n + ".value").view(B, ctx.shape[1], nh, d).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(d)
    if bias is not None:
        s = s + bias
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Sq, H)


def _att_block(P, n, inner, x, ctx, bias, c):
    """attention + LxmertAttentionOutput: LayerNorm(dense(att) + x)"""
    a = _attention(P, f"{n}.{inner}", x, ctx, bias, c.heads)
    return _ln(_lin(a, P, n + ".output.dense") + x, P, n + ".output.LayerNorm", c.ln_eps)


def _ffn(P, inter_name, out_name, x, c):
    h = F.gelu(_lin(x, P, inter_name + ".dense"))
    return _ln(_lin(h, P, out_name + ".dense") + x, P, out_name + ".LayerNorm", c.ln_eps)


def lxmert_forward(P: Dict[str, Tensor], ids: Tensor, mask: Tensor, token_type_ids: Optional[Tensor], feats: Tensor, boxes: Tensor,
                   c: LxmertConfig, visual_mask: Optional[Tensor] = None):
    """LxmertModel.forward in eval mode -> (language_output [B,T,H], vision_output [B,36,H], pooled_output [B,H])"""
    B, T = ids.shape
    e = "model.embeddings."
    tt = torch.zeros_like(ids) if token_type_ids is None else token_type_ids
    # all three tables are nn.Embedding(..., padding_idx=0) in LxmertEmbeddings: the forward reads row 0 like any other row, the
    # backward leaves its gradient at zero -- also for POSITION 0 (the CLS slot) and token type 0 (every token here)
    emb = lambda n, idx: F.embedding(idx, P[e + n], padding_idx=0)
    x = emb("word_embeddings.weight", ids) + emb("position_embeddings.weight", torch.arange(T))[None] + emb("token_type_embeddings.weight", tt)
    lang = _ln(x, P, e + "LayerNorm", c.ln_eps)
    v = "model.encoder.visn_fc."
    visn = (_ln(_lin(feats, P, v + "visn_fc"), P, v + "visn_layer_norm", c.ln_eps) + _ln(_lin(boxes, P, v + "box_fc"), P, v + "box_layer_norm", c.ln_eps)) / 2
    fmin = torch.finfo(torch.float32).min
    lbias = ((1.0 - mask.float()) * fmin)[:, None, None, :]
    vbias = None if visual_mask is None else ((1.0 - visual_mask.float()) * fmin)[:, None, None, :]
    for i in range(c.l_layers):
        b = f"model.encoder.layer.{i}."
        a = _att_block(P, b + "attention", "self", lang, lang, lbias, c)
        lang = _ffn(P, b + "intermediate", b + "output", a, c)
    for i in range(c.r_layers):
        b = f"model.encoder.r_layers.{i}."
        a = _att_block(P, b + "attention", "self", visn, visn, vbias, c)
        visn = _ffn(P, b + "intermediate", b + "output", a, c)
    for i in range(c.x_layers):
        b = f"model.encoder.x_layers.{i}."
        # ONE cross-attention module, both directions, both from the layer's inputs
        la = _att_block(P, b + "visual_attention", "att", lang, visn, vbias, c)
        va = _att_block(P, b + "visual_attention", "att", visn, lang, lbias, c)
        la = _att_block(P, b + "lang_self_att", "self", la, la, lbias, c)
        va = _att_block(P, b + "visn_self_att", "self", va, va, vbias, c)
        lang = _ffn(P, b + "lang_inter", b + "lang_output", la, c)
        visn = _ffn(P, b + "visn_inter", b + "visn_output", va, c)
    pooled = torch.tanh(_lin(lang[:, 0], P, "model.pooler.dense"))
    return lang, visn, pooled


# --------------------------------------------------------------------------------------
# the reference head and loss mixing
# --------------------------------------------------------------------------------------
def early_forward(P, ids, mask, token_type_ids, feats, boxes, c: LxmertConfig, tim_inputs=None):
    """reference models/mm_early.py:121-163 (eval mode: dropout off) -> (linear_output, max_embeddings_t, max_embeddings_v, out_tim)"""
    x_t, x_v, _ = lxmert_forward(P, ids, mask, token_type_ids, feats, boxes, c)
    out = _lin(torch.relu(_lin(x_t[:, 0], P, "linear_fusion")), P, "linear")
    last = x_t.clone().detach()                                   # :139-143: the text embedding carries no gradient
    last[mask.unsqueeze(-1).expand(x_t.shape).float() == 0] = -1e9
    emb_t = last.max(1)[0]
    emb_v = x_v.max(1)[0]
    out_tim = None
    if tim_inputs is not None:
        t_ids, t_mask, t_tt = tim_inputs
        x_t2, _, _ = lxmert_forward(P, t_ids, t_mask, t_tt, feats, boxes, c)
        out_tim = _lin(x_t2[:, 0], P, "linear_tim")
    return out, emb_t, emb_v, out_tim


def logits_per_text(P, emb_t, emb_v):
    """reference models/mm_early.py:165-172"""
    t = emb_t / emb_t.norm(p=2, dim=-1, keepdim=True)
    v = emb_v / emb_v.norm(p=2, dim=-1, keepdim=True)
    return (t @ v.t()) * P["logit_scale"].exp()


def mix_loss(P, out, onehot, weight, emb_t, emb_v, out_tim, lbl_tim, use_itc, use_itm, beta_itc=0.1, beta_itm=0.1):
    """reference models/mm_early.py:366-379"""
    lc = cls_loss(out, onehot, weight)
    if use_itc and use_itm:
        return (1 - (beta_itc + beta_itm)) * lc + beta_itc * clip_loss(logits_per_text(P, emb_t, emb_v)) + beta_itm * itm_loss(out_tim, lbl_tim)
    if use_itc:
        return (1 - beta_itc) * lc + beta_itc * clip_loss(logits_per_text(P, emb_t, emb_v))
    if use_itm:
        return (1 - beta_itm) * lc + beta_itm * itm_loss(out_tim, lbl_tim)
    return lc


def synthetic_batch(c: LxmertConfig, B: int, T: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, c.vocab, (B, T), generator=g)
    lens = torch.randint(3, T + 1, (B,), generator=g)
    lens[0] = T
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()
    ids = ids * mask                                              # pad id 0
    ids[:, 0] = 101 % c.vocab
    feats = torch.rand(B, c.n_boxes, c.feat_dim, generator=g) * 2.0
    boxes = torch.rand(B, c.n_boxes, c.pos_dim, generator=g)
    labels = torch.randint(0, c.num_labels, (B,), generator=g)
    onehot = F.one_hot(labels, c.num_labels)
    return ids, mask, torch.zeros_like(ids), feats, boxes, onehot
