"""CPU oracle for the late-fusion fine-tuning path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) restatement of the algorithm that the
reference executes for `models/mm_late.py:MM_Model.forward` and the train-step
body of `models/mm_late.py:MMLate_Model.train`.  It is the checker for the HIP
path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.  The product path (`smtc_amd`) never imports anything from
`oracle/` and fails loudly when the HIP library is missing.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the reference's
own `MM_Model` (behind a shim for packages the container lacks),
loads the weights produced by `make_params()` below into it, runs it, and
commits inputs + outputs under `tests/golden/*.npz`.  `tests/test_oracle_golden.py`
checks this restatement against those vectors on CPU.

The encoders' arithmetic is not in the reference repo; it lives in
`transformers==4.25.1` (timrel-env.yml:122), un-vendored.  The restated
algorithm follows (HF = transformers, as installed 5.15.0, same mathematics):
  * ViT-B/16        HF:models/vit/modeling_vit.py:42-69,72-161,192-301,336-388
  * BERT / XLM-R    HF:models/bert/modeling_bert.py:98-108,111-203,282-351,451-463
                    HF:models/xlm_roberta/modeling_xlm_roberta.py:56-155 (position ids)
  * dual encoder    HF:models/vision_text_dual_encoder/modeling_vision_text_dual_encoder.py:244-292
  * fusion + heads  reference models/mm_late.py:91-113,148-210
  * losses          reference models/utils.py:225-231, models/run_mm_late.py:85,97,
                    models/mm_late.py:471-487
  * ITM sampling    reference models/mm_late.py:389-414
  * optimizer       reference models/utils.py:280-292, models/mm_late.py:420-422 (torch AdamW)

Parameter names are the reference checkpoint's `state_dict` keys
(transformers 4.25.1 naming).
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field, asdict
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    hidden: int = 768           # models/config.py:82-84 hard-wires 768
    heads: int = 12
    inter: int = 3072
    layers_txt: int = 12
    layers_img: int = 12
    vocab: int = 250002         # Bernice (XLM-R tokenizer); BERT: 30522
    max_pos: int = 130          # XLM-R needs T+2 rows for T=128; BERT: 512
    type_vocab: int = 1         # XLM-R: 1, BERT: 2
    txt_kind: str = "xlmr"      # "xlmr" (bernice/roberta/bertweet) | "bert"
    pad_id: int = 1             # XLM-R pad id 1; BERT 0
    ln_eps_txt: float = 1e-5    # XLM-R 1e-5; BERT 1e-12
    ln_eps_img: float = 1e-12
    image: int = 224
    patch: int = 16
    proj_dim: int = 512
    num_labels: int = 2
    fusion: str = "attention"   # "attention" | "concat"
    p_hidden: float = 0.1       # text hidden dropout
    p_attn: float = 0.1         # text attention-prob dropout
    p_head: float = 0.05        # --dropout
    # image tower: "vit" (HF ViTModel, ViT-B/16) or "clip" (HF CLIPVisionModel; BASELINE config 4: CLIP-ViT-L/14 = 1024 wide,
    # 16 heads, 4096 MLP, patch 14, eps 1e-5).  0 = the text tower's sizes.
    img_kind: str = "vit"
    hidden_img: int = 0
    heads_img: int = 0
    inter_img: int = 0

    @property
    def Hv(self) -> int:
        return self.hidden_img or self.hidden

    @property
    def Iv(self) -> int:
        return self.inter_img or self.inter

    @property
    def heads_v(self) -> int:
        return self.heads_img or self.heads

    @property
    def n_patches(self) -> int:
        return (self.image // self.patch) ** 2

    @property
    def img_tokens(self) -> int:
        return self.n_patches + 1


# --------------------------------------------------------------------------------------
# deterministic parameter recipe (shared by golden generation, oracle and the HIP tests)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys (transformers 4.25.1 naming) -> shapes, in checkpoint order."""
    H, I = cfg.hidden, cfg.inter
    s: Dict[str, Tuple[int, ...]] = {}
    de = "dual_encoder."
    s[de + "logit_scale"] = ()
    if cfg.img_kind == "clip":
        return _param_shapes_clip(cfg, s)
    vm = de + "vision_model."
    s[vm + "embeddings.cls_token"] = (1, 1, H)
    s[vm + "embeddings.position_embeddings"] = (1, cfg.img_tokens, H)
    s[vm + "embeddings.patch_embeddings.projection.weight"] = (H, 3, cfg.patch, cfg.patch)
    s[vm + "embeddings.patch_embeddings.projection.bias"] = (H,)
    for l in range(cfg.layers_img):
        p = f"{vm}encoder.layer.{l}."
        for n in ("query", "key", "value"):
            s[p + f"attention.attention.{n}.weight"] = (H, H)
            s[p + f"attention.attention.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "layernorm_before.weight"] = (H,)
        s[p + "layernorm_before.bias"] = (H,)
        s[p + "layernorm_after.weight"] = (H,)
        s[p + "layernorm_after.bias"] = (H,)
    s[vm + "layernorm.weight"] = (H,)
    s[vm + "layernorm.bias"] = (H,)
    s[vm + "pooler.dense.weight"] = (H, H)
    s[vm + "pooler.dense.bias"] = (H,)
    return _param_shapes_text_heads(cfg, s)


def _param_shapes_clip(cfg: OracleConfig, s):
    """HF CLIPVisionModel inside the dual encoder, transformers 4.25.1 key names (vision_model.vision_model.*)."""
    Hv, Iv = cfg.Hv, cfg.Iv
    vm = "dual_encoder.vision_model.vision_model."
    s[vm + "embeddings.class_embedding"] = (Hv,)
    s[vm + "embeddings.patch_embedding.weight"] = (Hv, 3, cfg.patch, cfg.patch)
    s[vm + "embeddings.position_embedding.weight"] = (cfg.img_tokens, Hv)
    s[vm + "pre_layrnorm.weight"] = (Hv,)
    s[vm + "pre_layrnorm.bias"] = (Hv,)
    for l in range(cfg.layers_img):
        p = f"{vm}encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (Hv, Hv)
            s[p + f"self_attn.{n}.bias"] = (Hv,)
        s[p + "layer_norm1.weight"] = (Hv,)
        s[p + "layer_norm1.bias"] = (Hv,)
        s[p + "mlp.fc1.weight"] = (Iv, Hv)
        s[p + "mlp.fc1.bias"] = (Iv,)
        s[p + "mlp.fc2.weight"] = (Hv, Iv)
        s[p + "mlp.fc2.bias"] = (Hv,)
        s[p + "layer_norm2.weight"] = (Hv,)
        s[p + "layer_norm2.bias"] = (Hv,)
    s[vm + "post_layernorm.weight"] = (Hv,)
    s[vm + "post_layernorm.bias"] = (Hv,)
    return _param_shapes_text_heads(cfg, s)


def _param_shapes_text_heads(cfg: OracleConfig, s):
    H, I = cfg.hidden, cfg.inter
    de = "dual_encoder."
    tm = de + "text_model."
    s[tm + "embeddings.word_embeddings.weight"] = (cfg.vocab, H)
    s[tm + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    s[tm + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab, H)
    s[tm + "embeddings.LayerNorm.weight"] = (H,)
    s[tm + "embeddings.LayerNorm.bias"] = (H,)
    for l in range(cfg.layers_txt):
        p = f"{tm}encoder.layer.{l}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s[tm + "pooler.dense.weight"] = (H, H)
    s[tm + "pooler.dense.bias"] = (H,)
    s[de + "visual_projection.weight"] = (cfg.proj_dim, cfg.Hv)
    s[de + "text_projection.weight"] = (cfg.proj_dim, H)
    # heads, reference models/mm_late.py:73-89
    for n, shp in (("fc_Q", (H, H)), ("fc_K", (H, H)), ("fc_V", (H, H)), ("aspectattention", (1, H)),
                   ("linear_fusion", (H, H + cfg.Hv)), ("linear_cls", (cfg.num_labels, H)),
                   ("linear_tim", (2, H)), ("linear_iadds", (2, H)),
                   ("linear_gmu_t", (2 * H, H)), ("linear_gmu_v", (2 * H, H))):
        s[n + ".weight"] = shp
        s[n + ".bias"] = (shp[0],)
    return s


def make_param(name: str, shape: Tuple[int, ...], seed: int) -> Tensor:
    """Value of one parameter: depends only on (name, shape, seed).

    Weights ~ N(0, 0.02) (HF initializer_range), LayerNorm weights 1 + 0.1 N(0,1),
    biases 0.02 N(0,1) (non-zero on purpose so bias paths are exercised), the word-embedding
    padding row is left random (the forward reads it for pad tokens; its gradient is zero).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 1000003 * seed) % (2 ** 63))
    if name.endswith("logit_scale"):
        return torch.tensor(2.6592)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if ("LayerNorm.weight" in name or ("layernorm" in name or "layer_norm" in name or "layrnorm" in name) and name.endswith(".weight")):
        return 1.0 + 0.1 * x
    return 0.02 * x


def make_params(cfg: OracleConfig, seed: int = 0) -> Dict[str, Tensor]:
    return {k: make_param(k, shp, seed) for k, shp in param_shapes(cfg).items()}


# --------------------------------------------------------------------------------------
# dropout: counter-based hash shared bit-for-bit with the HIP kernels
# (socialmedia-textimage-classification-auxlosses_amd/csrc/mmhip_common.h: mm_rng_u32)
# --------------------------------------------------------------------------------------
STREAM_EMBED = 1
STREAM_HEAD = 2


def stream_attn(layer: int) -> int:
    return 16 + 4 * layer


def stream_attn_out(layer: int) -> int:
    return 16 + 4 * layer + 1


def stream_ffn_out(layer: int) -> int:
    return 16 + 4 * layer + 2


def rng_u32(idx: np.ndarray, stream: int, seed: int) -> np.ndarray:
    x = idx.astype(np.uint32) ^ np.uint32(seed & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = x * np.uint32(0x9E3779B1)
        x ^= x >> np.uint32(15)
        x = x + np.uint32((stream * 0x85EBCA77 + ((seed >> 32) & 0xFFFFFFFF)) & 0xFFFFFFFF)
        x ^= x >> np.uint32(13)
        x = x * np.uint32(0xC2B2AE3D)
        x ^= x >> np.uint32(16)
    return x


def drop_threshold16(p: float) -> int:
    """element dropped when its 16 random bits < thresh16 (kernels: DropCfg.thresh16)."""
    return min(int(round(p * 65536.0)), 65535)


def hash_keep_mask(n: int, offset: int, stream: int, seed: int, p: float) -> np.ndarray:
    """keep[i] for linear element indices offset..offset+n-1 (True = kept).  Element e uses 16 bits of
    rng_u32(e >> 1): the low half for even e, the high half for odd e (mmhip_common.h: mm_keep)."""
    e = (np.arange(n, dtype=np.uint64) + np.uint64(offset)).astype(np.uint32)
    h = rng_u32(e >> np.uint32(1), stream, seed)
    r = np.where((e & np.uint32(1)) == 1, h >> np.uint32(16), h & np.uint32(0xFFFF))
    return r >= np.uint32(drop_threshold16(p))


def keep_scale(p: float) -> float:
    return 1.0 / (1.0 - drop_threshold16(p) / 65536.0)


class Dropout:
    """mode 'none' (eval / p=0), 'torch' (torch RNG), 'hash' (the HIP kernels' masks)."""

    def __init__(self, mode: str = "none", seed: int = 0):
        self.mode, self.seed = mode, seed

    def __call__(self, x: Tensor, p: float, stream: int, offset: int = 0) -> Tensor:
        if self.mode == "none" or p <= 0.0:
            return x
        if self.mode == "torch":
            return F.dropout(x, p, training=True)
        keep = torch.from_numpy(hash_keep_mask(x.numel(), offset, stream, self.seed, p)).view(x.shape)
        return x * keep.to(x.dtype) * keep_scale(p)


# --------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# rounding emulation: the 16-bit HIP modes round BOTH operands of every matrix product to bf16 / f16,
# accumulate in fp32 and round every activation they STORE between kernels (and, in the backward, every stored gradient).
# `with rounding("bf16"):` makes this restatement do the same at the same places, so that a 16-bit HIP path can be compared
# against an execution with its own rounding policy (expected agreement: the noise of single rounding decisions) instead of
# against fp32 through a band wide enough for twelve layers of operand rounding.  Policy None = exact fp32 (the pinned oracle).
#   stored activation   _q   forward: round            backward: round the gradient (it is stored in 16 bits too)
#   MFMA operand only   _qo  forward: round            backward: identity          (softmax weights P; weights via _qw)
#   gradient operand    _qg  forward: identity         backward: round             (dS, an operand of the dQ / dK products)
# f16 carries text-tower gradients multiplied by `loss_scale` (engine.hip gscale(): 1024) -- range, not precision.
# --------------------------------------------------------------------------------------
class _Policy:
    # what is rounded, each one of None (fp32) | "bf16" | "f16" | "bf16x2" / "f16x2" (hi + lo pair = 16 / 22 significant bits):
    op_a = None       # activation operand of a matrix product
    op_w = None       # weight operand
    st_act = None     # stored activations (qkv, ctx, FC1 output and stash) and stored gradients
    st_resid = None   # stored residual-stream tensors: LayerNorm INPUTS (pre1, pre2), the image tower's x
    st_ln = None      # stored LayerNorm OUTPUTS (x of the next layer, a1)
    loss_scale = 1.0


_POLICY = _Policy()
_FIELDS = ("op_a", "op_w", "st_act", "st_resid", "st_ln")


class rounding:
    """context manager: `with rounding("bf16" | "f16" | None, loss_scale=..., **overrides)`; overrides name single fields of the
    policy (op_a, op_w, st_act, st_resid, st_ln) for the numerics study of tools/numerics_study.py."""

    def __init__(self, round_operands: Optional[str] = None, loss_scale: Optional[float] = None, **over):
        base = None if round_operands in (None, "none", "fp32") else round_operands
        self.new = {k: over.get(k, base) for k in _FIELDS}
        any16 = any(v == "f16" for v in self.new.values())
        self.new["loss_scale"] = loss_scale if loss_scale is not None else (1024.0 if any16 else 1.0)

    def __enter__(self):
        self.prev = {k: getattr(_POLICY, k) for k in _FIELDS + ("loss_scale",)}
        for k, v in self.new.items():
            setattr(_POLICY, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            setattr(_POLICY, k, v)
        return False


def _rnd(x: Tensor, kind: str) -> Tensor:
    if kind == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if kind == "f16":
        return x.to(torch.float16).to(torch.float32)
    if kind == "bf16x2":
        hi = x.to(torch.bfloat16).to(torch.float32)
        return hi + (x - hi).to(torch.bfloat16).to(torch.float32)
    if kind == "f16x2":
        hi = x.to(torch.float16).to(torch.float32)
        return hi + (x - hi).to(torch.float16).to(torch.float32)
    raise ValueError(kind)


class _RoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, fwd, bwd, gs):
        ctx.kind, ctx.bwd, ctx.gs = kind, bwd, gs
        return _rnd(x, kind) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.bwd:
            g = _rnd(g * ctx.gs, ctx.kind) / ctx.gs
        return g, None, None, None, None


def _q(x: Tensor, where: str = "act") -> Tensor:
    """a tensor the 16-bit path stores between kernels (where: act | resid | ln); its gradient is stored the same way"""
    kind = getattr(_POLICY, "st_" + where)
    return x if kind is None else _RoundFn.apply(x, kind, True, True, _POLICY.loss_scale)


def _qo(x: Tensor) -> Tensor:
    return x if _POLICY.op_a is None else _RoundFn.apply(x, _POLICY.op_a, True, False, 1.0)


def _qg(x: Tensor) -> Tensor:
    return x if _POLICY.op_a is None else _RoundFn.apply(x, _POLICY.op_a, False, True, _POLICY.loss_scale)


def _qw(w: Tensor) -> Tensor:
    """weight as a 16-bit GEMM operand: the fp32 master receives the unrounded fp32 weight gradient."""
    return w if _POLICY.op_w is None else _RoundFn.apply(w, _POLICY.op_w, True, False, 1.0)


class _GeluStash(torch.autograd.Function):
    """FC1 epilogue of the 16-bit modes: h = gelu(v) on the fp32 accumulator value; the backward multiplies by gelu'(u) of the
    STORED (rounded) pre-activation u (gemm8.hip EP_GELU / EP_MULG)."""

    @staticmethod
    def forward(ctx, v, kind):
        ctx.save_for_backward(_rnd(v, kind))
        return F.gelu(v)

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(u * (2.0 ** -0.5)))
        pdf = torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))
        return g * (cdf + u * pdf), None


def _gelu(v: Tensor) -> Tensor:
    return F.gelu(v) if _POLICY.st_act is None else _GeluStash.apply(v, _POLICY.st_act)


# --------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------
def _lin(x: Tensor, P: Dict[str, Tensor], name: str) -> Tensor:
    """Linear inside a tower: 16-bit operands under a rounding policy (fp32 accumulate, fp32 bias); the caller rounds the result
    where the HIP path stores it."""
    return F.linear(_qo(x), _qw(P[name + ".weight"]), P.get(name + ".bias"))


def _lin32(x: Tensor, P: Dict[str, Tensor], name: str) -> Tensor:
    """Linear of the heads: fp32 in every mode (heads.hip small_gemm on v_mfma_f32_32x32x2_f32)."""
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _ln(x: Tensor, P: Dict[str, Tensor], name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def _heads(x: Tensor, nh: int) -> Tensor:
    B, S, H = x.shape
    return x.view(B, S, nh, H // nh).permute(0, 2, 1, 3)


def _attend(q: Tensor, k: Tensor, v: Tensor, scale: float, bias: Optional[Tensor], dropf) -> Tensor:
    """softmax(q k^T * scale + bias) v per head; under a rounding policy q, k, v arrive rounded (stored qkv), the dropped softmax
    weights are rounded as the P.V operand and dS as the operand of the dQ / dK products (attention.hip)."""
    s = _qg(q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    a = dropf(torch.softmax(s, dim=-1))
    return _qo(a) @ v


def vit_forward(P: Dict[str, Tensor], pixels: Tensor, cfg: OracleConfig,
                collect: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """HF:models/vit/modeling_vit.py:373-388 (ViTModel.forward); all dropouts are 0.0."""
    vm = "dual_encoder.vision_model."
    assert cfg.Hv == cfg.hidden and cfg.Iv == cfg.inter, "the ViT restatement shares the text tower's sizes"
    B = pixels.shape[0]
    w = P[vm + "embeddings.patch_embeddings.projection.weight"]
    b = P[vm + "embeddings.patch_embeddings.projection.bias"]
    x = F.conv2d(_qo(pixels), _qw(w), b, stride=cfg.patch).flatten(2).transpose(1, 2)  # :60,69
    x = torch.cat([P[vm + "embeddings.cls_token"].expand(B, -1, -1), _q(x)], dim=1)    # :146-150
    x = _q(x + P[vm + "embeddings.position_embeddings"], "resid")                               # :153-155
    scale = (cfg.hidden // cfg.heads) ** -0.5
    ident = lambda a: a
    for l in range(cfg.layers_img):
        p = f"{vm}encoder.layer.{l}."
        h = _q(_ln(x, P, p + "layernorm_before", cfg.ln_eps_img), "ln")                       # :274
        q = _heads(_q(_lin(h, P, p + "attention.attention.query")), cfg.heads)          # :216-218
        k = _heads(_q(_lin(h, P, p + "attention.attention.key")), cfg.heads)
        v = _heads(_q(_lin(h, P, p + "attention.attention.value")), cfg.heads)
        c = _q(_attend(q, k, v, scale, None, ident)).permute(0, 2, 1, 3).reshape(B, -1, cfg.hidden)   # :164-189
        x = _q(x + _lin(c, P, p + "attention.output.dense"), "resid")                            # :236,276-277
        h = _q(_ln(x, P, p + "layernorm_after", cfg.ln_eps_img), "ln")                        # :281
        h = _q(_gelu(_lin(h, P, p + "intermediate.dense")))                             # :250-251 (erf)
        x = _q(x + _lin(h, P, p + "output.dense"), "resid")                                   # :252,283-284
        if collect is not None:
            collect.append(x)
    x = _q(_ln(x, P, vm + "layernorm", cfg.ln_eps_img), "ln")                                 # :385
    pooled = torch.tanh(_lin32(x[:, 0], P, vm + "pooler.dense"))                        # :289-301
    return x, pooled


def clip_vision_forward(P: Dict[str, Tensor], pixels: Tensor, cfg: OracleConfig,
                        collect: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """HF:models/clip/modeling_clip.py -- CLIPVisionEmbeddings (bias-free patch conv, class embedding, learned positions),
    CLIPVisionTransformer.forward (pre_layrnorm; pre-LN encoder layers with quick-GELU MLPs; last_hidden_state is the encoder
    output as it is, pooler_output = post_layernorm of its CLS row).  -> (last_hidden_state [B,P,Hv], pooler_output [B,Hv])"""
    vm = "dual_encoder.vision_model.vision_model."
    B, Hv, nh = pixels.shape[0], cfg.Hv, cfg.heads_v
    x = F.conv2d(_qo(pixels), _qw(P[vm + "embeddings.patch_embedding.weight"]), None, stride=cfg.patch).flatten(2).transpose(1, 2)
    x = torch.cat([P[vm + "embeddings.class_embedding"].expand(B, 1, -1), _q(x)], dim=1)
    x = _q(x + P[vm + "embeddings.position_embedding.weight"], "resid")
    x = _q(_ln(x, P, vm + "pre_layrnorm", cfg.ln_eps_img), "resid")
    scale = (Hv // nh) ** -0.5
    ident = lambda a: a
    for l in range(cfg.layers_img):
        p = f"{vm}encoder.layers.{l}."
        h = _q(_ln(x, P, p + "layer_norm1", cfg.ln_eps_img), "ln")
        q = _heads(_q(_lin(h, P, p + "self_attn.q_proj")), nh)
        k = _heads(_q(_lin(h, P, p + "self_attn.k_proj")), nh)
        v = _heads(_q(_lin(h, P, p + "self_attn.v_proj")), nh)
        c = _q(_attend(q, k, v, scale, None, ident)).permute(0, 2, 1, 3).reshape(B, -1, Hv)
        x = _q(x + _lin(c, P, p + "self_attn.out_proj"), "resid")
        h = _lin(_q(_ln(x, P, p + "layer_norm2", cfg.ln_eps_img), "ln"), P, p + "mlp.fc1")
        h = _q(h * torch.sigmoid(1.702 * h))                                 # quick_gelu
        x = _q(x + _lin(h, P, p + "mlp.fc2"), "resid")
        if collect is not None:
            collect.append(x)
    pooled = _ln(x[:, 0], P, vm + "post_layernorm", cfg.ln_eps_img)
    return x, pooled


def text_position_ids(ids: Tensor, cfg: OracleConfig) -> Tensor:
    if cfg.txt_kind == "xlmr":   # HF:models/xlm_roberta/modeling_xlm_roberta.py:142-155
        m = (ids != cfg.pad_id).to(torch.int64)
        return torch.cumsum(m, dim=1) * m + cfg.pad_id
    return torch.arange(ids.shape[1], dtype=torch.int64).unsqueeze(0).expand_as(ids)


def text_forward(P: Dict[str, Tensor], ids: Tensor, mask: Tensor, cfg: OracleConfig,
                 drop: Optional[Dropout] = None, post_offset: int = 0,
                 collect: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """BertModel / XLMRobertaModel forward (post-LN).  `post_offset` = index of this call's first
    post inside the batched text pass of the HIP engine (ITM rows follow the original rows)."""
    drop = drop or Dropout("none")
    tm = "dual_encoder.text_model."
    B, T = ids.shape
    H, nh = cfg.hidden, cfg.heads
    pos = text_position_ids(ids, cfg)
    x = (P[tm + "embeddings.word_embeddings.weight"][ids]
         + P[tm + "embeddings.token_type_embeddings.weight"][0]
         + P[tm + "embeddings.position_embeddings.weight"][pos])
    x = _ln(x, P, tm + "embeddings.LayerNorm", cfg.ln_eps_txt)
    x = _q(drop(x, cfg.p_hidden, STREAM_EMBED, post_offset * T * H), "ln")
    bias = (1.0 - mask.to(torch.float32))[:, None, None, :] * torch.finfo(torch.float32).min
    scale = (H // nh) ** -0.5
    for l in range(cfg.layers_txt):
        p = f"{tm}encoder.layer.{l}."
        q = _heads(_q(_lin(x, P, p + "attention.self.query")), nh)
        k = _heads(_q(_lin(x, P, p + "attention.self.key")), nh)
        v = _heads(_q(_lin(x, P, p + "attention.self.value")), nh)
        dropf = lambda a, l=l: drop(a, cfg.p_attn, stream_attn(l), post_offset * nh * T * T)
        c = _q(_attend(q, k, v, scale, bias, dropf)).permute(0, 2, 1, 3).reshape(B, T, H)
        o = drop(_lin(c, P, p + "attention.output.dense"), cfg.p_hidden, stream_attn_out(l), post_offset * T * H)
        a1 = _q(_ln(_q(o + x, "resid"), P, p + "attention.output.LayerNorm", cfg.ln_eps_txt), "ln")
        h = _q(_gelu(_lin(a1, P, p + "intermediate.dense")))
        f = drop(_lin(h, P, p + "output.dense"), cfg.p_hidden, stream_ffn_out(l), post_offset * T * H)
        x = _q(_ln(_q(f + a1, "resid"), P, p + "output.LayerNorm", cfg.ln_eps_txt), "ln")
        if collect is not None:
            collect.append(x)
    pooled = torch.tanh(_lin32(x[:, 0], P, tm + "pooler.dense"))
    return x, pooled


def itc_logits(P: Dict[str, Tensor], t_pool: Tensor, v_pool: Tensor) -> Tensor:
    """logits_per_text; HF dual encoder :261-274."""
    img = F.linear(v_pool, P["dual_encoder.visual_projection.weight"])
    txt = F.linear(t_pool, P["dual_encoder.text_projection.weight"])
    img = img / img.norm(dim=-1, keepdim=True)
    txt = txt / txt.norm(dim=-1, keepdim=True)
    return txt @ img.t() * P["dual_encoder.logit_scale"].exp()


def _relu(pre: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ReLU; with `mask` (tests only) the given 0/1 pattern decides which units pass instead of the sign of `pre`: two executions that
    differ by 16-bit rounding flip units whose pre-activation is near zero, and a per-tensor gradient comparison then measures those
    flips instead of the backward under test"""
    return F.relu(pre) if mask is None else pre * mask


def mm_fusion(P: Dict[str, Tensor], x_t: Tensor, x_v: Tensor, cfg: OracleConfig, relu_mask: Optional[Tensor] = None) -> Tensor:
    """reference models/mm_late.py:91-113 + Scaled_Dot_Product_Attention :195-210."""
    if cfg.fusion == "concat":
        z = torch.cat((x_t[:, 0, :], x_v[:, 0, :]), dim=1)
        return _relu(_lin32(z, P, "linear_fusion"), relu_mask)
    if cfg.fusion == "attention":
        N, L, E = x_t.shape
        Q, K, V = _lin32(x_t, P, "fc_Q"), _lin32(x_v, P, "fc_K"), _lin32(x_v, P, "fc_V")
        scale = K.shape[-1] ** -0.5
        att = torch.softmax(Q @ K.permute(0, 2, 1) * scale, dim=-1)
        ctx = (att @ V).view(N, L, E)
        z = torch.cat((x_t[:, 0, :], ctx[:, 0, :]), dim=1)
        return _relu(_lin32(z, P, "linear_fusion"), relu_mask)
    raise ValueError(cfg.fusion)


def mm_forward(P: Dict[str, Tensor], ids: Tensor, mask: Tensor, pixels: Tensor, cfg: OracleConfig,
               tim_inputs: Optional[Tuple[Tensor, Tensor]] = None, drop: Optional[Dropout] = None,
               collect: Optional[dict] = None, relu_mask: Optional[Tensor] = None):
    """reference models/mm_late.py:148-193 -> (out_cls, logits_per_text, out_tim, None, mm_features)."""
    drop = drop or Dropout("none")
    cv = collect.setdefault("vit_layers", []) if collect is not None else None
    ct = collect.setdefault("txt_layers", []) if collect is not None else None
    B = ids.shape[0]
    x_v, v_pool = (clip_vision_forward if cfg.img_kind == "clip" else vit_forward)(P, pixels, cfg, cv)
    x_t, t_pool = text_forward(P, ids, mask, cfg, drop, 0, ct)
    logits_per_text = itc_logits(P, t_pool, v_pool)
    feats = mm_fusion(P, x_t, x_v, cfg, relu_mask)
    out_cls = _lin32(drop(feats, cfg.p_head, STREAM_HEAD, 0), P, "linear_cls")
    out_tim = None
    if tim_inputs is not None:
        tim_ids, tim_mask = tim_inputs
        # second dual-encoder call; the frozen, dropout-free ViT output is identical
        x_t2, _ = text_forward(P, tim_ids, tim_mask, cfg, drop, B, None)
        out_tim = _lin32(mm_fusion(P, x_t2, x_v, cfg), P, "linear_tim")      # no dropout, :181-182
    return out_cls, logits_per_text, out_tim, None, feats


# --------------------------------------------------------------------------------------
# losses, ITM sampling, optimizer
# --------------------------------------------------------------------------------------
def clip_loss(sim: Tensor) -> Tensor:
    """reference models/utils.py:225-231."""
    tgt = torch.arange(len(sim), device=sim.device)
    return (F.cross_entropy(sim, tgt) + F.cross_entropy(sim.t(), tgt)) / 2.0


def cls_loss(out: Tensor, onehot: Tensor, weight: Optional[Tensor]) -> Tensor:
    """nn.CrossEntropyLoss(weight=w)(out, float one-hot): -(1/B) sum_b sum_c w_c y_bc logsoftmax(out)_bc
    (models/run_mm_late.py:85 with models/mm_late.py:471; SURVEY §8c known answer (1))."""
    lsm = torch.log_softmax(out, dim=1)
    w = torch.ones(out.shape[1], device=out.device) if weight is None else weight.to(out.device)
    return -(onehot.to(out.device, out.dtype) * lsm * w).sum() / out.shape[0]


def itm_loss(out_tim: Tensor, lbl: Tensor) -> Tensor:
    return F.cross_entropy(out_tim, lbl.to(out_tim.device))


def mix_loss(out_cls, onehot, weight, logits_per_text, out_tim, lbl_tim,
             use_itc: bool, use_itm: bool, beta_itc: float = 0.1, beta_itm: float = 0.1) -> Tensor:
    """reference models/mm_late.py:473-487."""
    lc = cls_loss(out_cls, onehot, weight)
    if use_itc and use_itm:
        return (1 - (beta_itc + beta_itm)) * lc + beta_itc * clip_loss(logits_per_text) + beta_itm * itm_loss(out_tim, lbl_tim)
    if use_itc:
        return (1 - beta_itc) * lc + beta_itc * clip_loss(logits_per_text)
    if use_itm:
        return (1 - beta_itm) * lc + beta_itm * itm_loss(out_tim, lbl_tim)
    return lc


def prepare_itm_inputs(ids: Tensor, mask: Tensor):
    """reference models/mm_late.py:389-414 (numpy global RNG; same call order)."""
    tim_ids, tim_mask = ids.clone(), mask.clone()
    labels = []
    B = ids.shape[0]
    if B > 1:
        for idx in range(B):
            if np.random.choice([True, False]):
                labels.append(0)
                new_idx = np.random.choice(list(set(range(B)) - {idx}))
                tim_ids[idx] = ids[new_idx]
                tim_mask[idx] = mask[new_idx]
            else:
                labels.append(1)
    else:
        labels.append(1)
    return tim_ids, tim_mask, torch.tensor(labels, dtype=torch.long)


def trainable(name: str) -> bool:
    """reference models/mm_late.py:67-69: every dual_encoder parameter with 'vision' in its name is frozen."""
    return not (name.startswith("dual_encoder.") and "vision" in name)


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, wd: float,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.AdamW single-tensor update (in place)."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# synthetic posts (BASELINE.md §3 / SURVEY §8d)
# --------------------------------------------------------------------------------------
def synthetic_batch(cfg: OracleConfig, B: int, T: int = 128, seed: int = 1234, pad: bool = False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg.vocab, (B, T), generator=g, dtype=torch.int64)
    cls_id, eos_id = (0, 2) if cfg.txt_kind == "xlmr" else (101, 102)
    ids[:, 0] = cls_id
    mask = torch.ones(B, T, dtype=torch.int64)
    if pad:
        lens = torch.randint(4, T + 1, (B,), generator=g)
        lens[0] = T
        for b in range(B):
            n = int(lens[b])
            ids[b, n - 1] = eos_id
            ids[b, n:] = cfg.pad_id
            mask[b, n:] = 0
    else:
        ids[:, T - 1] = eos_id
    pixels = torch.rand(B, 3, cfg.image, cfg.image, generator=g) * 2 - 1
    labels = torch.randint(0, cfg.num_labels, (B,), generator=g)
    onehot = F.one_hot(labels, cfg.num_labels).to(torch.int64)
    return ids, mask, pixels, onehot
