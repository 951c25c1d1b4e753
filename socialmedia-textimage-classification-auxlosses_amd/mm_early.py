"""Early-fusion LXMERT (BASELINE config 5) -- reference models/mm_early.py:105-172 (class Lxmert) and :175-520 (MMEarly_Model).

FIRST VERSION (round 2): every matrix product, LayerNorm and attention of the model runs on the library's HIP kernels through
its operator-level C ABI (`mmhip_op_gemm_nt / _gemm_tn / _layernorm_* / _attn_*`, `mmhip_adamw`), chained by torch autograd
(one `autograd.Function` per operator; residual adds, GELU, dropout, embedding gathers, max-pooling and the small losses are torch
tensor ops on the GPU).  It is not a fused engine like mm_late.py: the step is launch-bound and there is no weight-gradient /
optimizer overlap.  There is no CPU path: the operators raise when the HIP library is missing.

Cross attention (queries and keys of different lengths: T tokens x 36 boxes) runs on the self-attention kernels: Q, K and V are
separate column blocks of one packed [rows, 3H] tensor, so row i carries query i and row j carries key / value j of the OTHER
stream; S = max(T, 36) rows per post, keys past the context length masked, query rows past the query length discarded (their
upstream gradient is zero).  S <= 128 (the attention backward's limit) covers the reference's max_length = 128.

State-dict keys are the reference module's (`model.*` = HF LxmertModel 4.25.1 naming, `linear_fusion`, `linear`, `linear_tim`,
`logit_scale`).  Parity: tests/test_gpu_early.py against tests/golden/lxmert_small.npz (the reference's own module) and the oracle.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dist as mmdist
from .utils import clip_loss, agg_metrics_val
from .config import metric_names

_DT = {"bf16": (_lib.BF16, torch.bfloat16), "f16": (_lib.F16, torch.float16), "bf16x3": (_lib.F32, torch.float32)}


def _p(t):
    # plain integers: the argtypes (c_void_p) convert them; building a c_void_p object per argument was a third of the step's host time
    return None if t is None else t.data_ptr()


_STREAM = [None]


def _s():
    """raw handle of the stream the operators launch on: torch's current stream, looked up once per forward / step
    (torch.cuda.current_stream() costs ~8 us; the step makes ~3000 launches)"""
    if _STREAM[0] is None:
        _STREAM[0] = torch.cuda.current_stream().cuda_stream
    return _STREAM[0]


def _refresh_stream():
    _STREAM[0] = torch.cuda.current_stream().cuda_stream


class _Ctx:
    """per-model operator context: activation dtype, 16-bit (or fp32) copies of the weights keyed by parameter version, dropout seeds"""

    def __init__(self, dtype):
        self.code, self.tdt = _DT[dtype]
        self.cache = {}
        self.calls = 0
        self.seed = 0x5DEECE66D
        self.gview = {}            # id(parameter) -> its slice of the flat gradient buffer: the operators accumulate there directly
        self.persistent = {}       # id(parameter) or tuple of ids (fused Q/K/V) -> (copy, transposed copy) refreshed once per step
        self.tn_queue = []         # weight-gradient products of the running backward pass (dy, x, C, colsum, M, Nn, Nc, lda, A address)

    def tn(self, dy, x, gw, gb, M, Nn, Nc, lda=None, a_ptr=None):
        """queue C[Nn,Nc] += dy[:, block]^T x (+ column sums into gb): all weight gradients of a backward pass leave in a few grouped launches
        (flush_tn): one 768 x 768 gradient alone is 18 tiles on a 256-CU chip"""
        self.tn_queue.append((dy, x, gw, gb, M, Nn, Nc, lda if lda is not None else Nn, a_ptr if a_ptr is not None else dy.data_ptr()))

    def flush_tn(self):
        q = self.tn_queue
        if not q:
            return
        arr = (_lib.TNProblem * len(q))()
        for i, (dy, x, gw, gb, M, Nn, Nc, lda, a_ptr) in enumerate(q):
            arr[i] = _lib.TNProblem(a_ptr, x.data_ptr(), gw.data_ptr(), M, Nn, Nc, lda, Nc, Nc, None if gb is None else gb.data_ptr())
        _lib.check(_lib.lib().mmhip_op_gemm_tn_group(self.code, C.cast(arr, C.c_void_p), len(q), 1, _s()), "gemm_tn_group")
        self.tn_queue = []

    def weight(self, w, transpose=False):
        """activation-typed copy of an fp32 [N,K] weight ([K,N] when transpose): the persistent copy (Lxmert.refresh_weights, one grouped
        launch per step) when the matrix has one, else cast on demand and cached by parameter version"""
        pers = self.persistent.get(id(w))
        if pers is not None:
            return pers[1 if transpose else 0]
        key = (id(w), transpose)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == w._version:
            return hit[1]
        N, K = w.shape
        if self.tdt == torch.float32:
            out = w.detach().t().contiguous() if transpose else w.detach()
        else:
            out = torch.empty((K, N) if transpose else (N, K), dtype=self.tdt, device=w.device)
            _lib.check(_lib.lib().mmhip_op_cast(self.code, _p(w.detach()), _p(out), N * K, N if transpose else 0, K if transpose else 0, _s()), "cast")
        self.cache[key] = (w._version, out)
        return out

    def weight_cat(self, ws, transpose=False):
        """activation-typed copy of the row-wise concatenation of fp32 [N_i, K] weights (fused Q / K / V projection)"""
        pers = self.persistent.get(tuple(id(w) for w in ws))
        if pers is not None:
            return pers[1 if transpose else 0]
        key = (tuple(id(w) for w in ws), transpose)
        ver = tuple(w._version for w in ws)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        cat = torch.cat([w.detach() for w in ws], dim=0)
        N, K = cat.shape
        if self.tdt == torch.float32:
            out = cat.t().contiguous() if transpose else cat
        else:
            out = torch.empty((K, N) if transpose else (N, K), dtype=self.tdt, device=cat.device)
            _lib.check(_lib.lib().mmhip_op_cast(self.code, _p(cat), _p(out), N * K, N if transpose else 0, K if transpose else 0, _s()), "cast")
        self.cache[key] = (ver, out)
        return out

    def bias_cat(self, bs):
        pers = self.persistent.get(("b",) + tuple(id(b) for b in bs))
        if pers is not None:
            return pers
        key = ("b",) + tuple(id(b) for b in bs)
        ver = tuple(b._version for b in bs)
        hit = self.cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        out = torch.cat([b.detach() for b in bs])
        self.cache[key] = (ver, out)
        return out

    def next_seed(self):
        self.calls += 1
        return (self.seed * 0x9E3779B97F4A7C15 + self.calls) & 0xFFFFFFFFFFFFFFFF


class _Linear(torch.autograd.Function):
    """y = x W^T + b on gemm_nt; dx = dy W (gemm_nt on the transposed copy), dW = dy^T x and db = column sums of dy (one gemm_tn)"""

    @staticmethod
    def forward(ctx, x, w, b, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=oc.tdt, device=x.device)
        _lib.check(_lib.lib().mmhip_op_gemm_nt(oc.code, _p(x), K, _p(oc.weight(w)), K, _p(y), N, M, N, K, _p(b), 0, None, 0, None, 0, 0.0, 0, 0,
                                               None, 0, 0, 0, _s()), "gemm_nt")
        ctx.save_for_backward(x, w)
        ctx.oc, ctx.has_b, ctx.b = oc, b is not None, b
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, w = ctx.saved_tensors
        oc, lib = ctx.oc, _lib.lib()
        dy = dy.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=oc.tdt, device=x.device)
            _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(dy), N, _p(oc.weight(w, True)), N, _p(dx), K, M, K, N, None, 0, None, 0, None, 0, 0.0, 0, 0,
                                            None, 0, 0, 0, _s()), "gemm_nt dx")
        # weight / bias gradients are ADDED into the parameters' slices of the flat gradient buffer (zeroed by AdamW); autograd gets None
        gw, gb = oc.gview[id(w)], (oc.gview[id(ctx.b)] if ctx.has_b else None)
        fused_db = ctx.has_b and N % 4 == 0                  # the column-sum leg of gemm_tn wants 4-element columns; the 3- / 2-wide heads sum in torch
        oc.tn(dy, x, gw, gb if fused_db else None, M, N, K)
        if ctx.has_b and not fused_db:
            gb.add_(dy.float().sum(0))
        return dx, None, None, None


class _LinearQKV(torch.autograd.Function):
    """[q | k | v] = x [Wq; Wk; Wv]^T + [bq | bk | bv] in ONE gemm_nt (self-attention: the packed rows the attention kernels read);
    backward: one gemm_nt for dx, one gemm_tn for the three weight gradients and their bias gradients"""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        M, K = x.shape
        N = wq.shape[0] + wk.shape[0] + wv.shape[0]
        y = torch.empty(M, N, dtype=oc.tdt, device=x.device)
        b = torch.cat([bq.detach(), bk.detach(), bv.detach()])
        _lib.check(_lib.lib().mmhip_op_gemm_nt(oc.code, _p(x), K, _p(oc.weight_cat((wq, wk, wv))), K, _p(y), N, M, N, K, _p(b), 0, None, 0, None, 0, 0.0, 0, 0,
                                               None, 0, 0, 0, _s()), "gemm_nt qkv")
        ctx.save_for_backward(x, wq, wk, wv)
        ctx.oc, ctx.bs = oc, (bq, bk, bv)
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, wq, wk, wv = ctx.saved_tensors
        oc, lib = ctx.oc, _lib.lib()
        dy = dy.contiguous()
        M, K = x.shape
        N = dy.shape[1]
        dx = torch.empty(M, K, dtype=oc.tdt, device=x.device)
        _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(dy), N, _p(oc.weight_cat((wq, wk, wv), True)), N, _p(dx), K, M, K, N, None, 0, None, 0, None, 0, 0.0, 0, 0,
                                        None, 0, 0, 0, _s()), "gemm_nt dx")
        n = wq.shape[0]
        for i, (w, b) in enumerate(((wq, ctx.bs[0]), (wk, ctx.bs[1]), (wv, ctx.bs[2]))):      # column block i of dy against x, added into the flat gradient
            oc.tn(dy, x, oc.gview[id(w)], oc.gview[id(b)], M, n, K, lda=N, a_ptr=dy.data_ptr() + i * n * dy.element_size())
        return dx, None, None, None, None, None, None, None


class _FFN(torch.autograd.Function):
    """y = GELU(x W1^T + b1) W2^T + b2: GELU and the pre-activation stash in the first GEMM's epilogue; backward: gelu' in the
    epilogue of the dh GEMM (the same fused epilogues mm_late's engine uses)"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        M, K = x.shape
        I = w1.shape[0]
        lib = _lib.lib()
        h = torch.empty(M, I, dtype=oc.tdt, device=x.device)
        u = torch.empty(M, I, dtype=oc.tdt, device=x.device)
        _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(x), K, _p(oc.weight(w1)), K, _p(h), I, M, I, K, _p(b1), 1, _p(u), I, None, 0, 0.0, 0, 0, None, 0, 0, 0, _s()), "ffn fc1")
        y = torch.empty(M, K, dtype=oc.tdt, device=x.device)
        _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(h), I, _p(oc.weight(w2)), I, _p(y), K, M, K, I, _p(b2), 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 0, _s()), "ffn fc2")
        ctx.save_for_backward(x, w1, w2, h, u)
        ctx.oc, ctx.bs = oc, (b1, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, w1, w2, h, u = ctx.saved_tensors
        oc, lib = ctx.oc, _lib.lib()
        dy = dy.contiguous()
        M, K = x.shape
        I = w1.shape[0]
        du = torch.empty(M, I, dtype=oc.tdt, device=x.device)          # (dy W2) * gelu'(u)
        _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(dy), K, _p(oc.weight(w2, True)), K, _p(du), I, M, I, K, None, 0, None, 0, _p(u), I, 0.0, 0, 0, None, 0, 0, 0, _s()), "ffn du")
        dx = torch.empty(M, K, dtype=oc.tdt, device=x.device)
        _lib.check(lib.mmhip_op_gemm_nt(oc.code, _p(du), I, _p(oc.weight(w1, True)), I, _p(dx), K, M, K, I, None, 0, None, 0, None, 0, 0.0, 0, 0, None, 0, 0, 0, _s()), "ffn dx")
        g = oc.gview
        oc.tn(dy, h, g[id(w2)], g[id(ctx.bs[1])], M, K, I)
        oc.tn(du, x, g[id(w1)], g[id(ctx.bs[0])], M, I, K)
        return dx, None, None, None, None, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        rows, width = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mmhip_op_layernorm_fwd(oc.code, _p(x), _p(y), _p(g), _p(b), _p(mean), _p(rstd), rows, width, eps, _s()), "ln_fwd")
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.oc, ctx.b = oc, b
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, g, mean, rstd = ctx.saved_tensors
        oc = ctx.oc
        rows, width = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().mmhip_op_layernorm_bwd(oc.code, _p(dy), _p(x), _p(g), _p(mean), _p(rstd), _p(dx), None, _p(oc.gview[id(g)]), _p(oc.gview[id(ctx.b)]),
                                                     rows, width, _s()), "ln_bwd")       # d gamma / d beta are added into the flat gradient
        return dx, None, None, None, None


class _Attention(torch.autograd.Function):
    """softmax(Q K^T / 8 + maskbias) V per (post, head) on packed [posts*S, 3H] rows; hash dropout on the probabilities"""

    @staticmethod
    def forward(ctx, qkv, maskbias, posts, S, heads, p_drop, seed, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        H = heads * 64
        out = torch.empty(posts * S, H, dtype=oc.tdt, device=qkv.device)
        lse = torch.empty(posts * heads * S, dtype=torch.float32, device=qkv.device)
        _lib.check(_lib.lib().mmhip_op_attn_fwd(oc.code, _p(qkv), _p(maskbias), _p(out), _p(lse), posts, S, heads, p_drop, seed, 7, _s()), "attn_fwd")
        ctx.save_for_backward(qkv, maskbias, out, lse)
        ctx.cfg = (posts, S, heads, p_drop, seed, oc)
        return out

    @staticmethod
    def backward(ctx, dctx):
        _STREAM[0] = ctx.stream
        qkv, maskbias, out, lse = ctx.saved_tensors
        posts, S, heads, p_drop, seed, oc = ctx.cfg
        dqkv = torch.empty_like(qkv)
        _lib.check(_lib.lib().mmhip_op_attn_bwd(oc.code, _p(qkv), _p(maskbias), _p(out), _p(dctx.contiguous()), _p(lse), _p(dqkv), posts, S, heads,
                                                p_drop, seed, 7, _s()), "attn_bwd")
        return dqkv, None, None, None, None, None, None, None


class _SelfAttBlock(torch.autograd.Function):
    """LayerNorm(dropout(dense(attention(x))) + x) of one stream as ONE autograd node and ONE native call per direction
    (mmhip_op_self_att_block_fwd / _bwd: fused Q/K/V GEMM, attention, output GEMM with bias + hash dropout + residual in its epilogue,
    LayerNorm; backward: LayerNorm backward emitting the dropout-backward copy, two input-gradient GEMMs, attention backward)"""

    @staticmethod
    def forward(ctx, x, maskbias, wq, bq, wk, bk, wv, bv, wo, bo, g, b, posts, S, heads, p_att, p_hid, seed, eps, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        M, H = x.shape
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        f = lambda n: torch.empty(n, dtype=torch.float32, device=x.device)
        qkv, att, pre, y, lse, mean, rstd = e(M, 3 * H), e(M, H), e(M, H), e(M, H), f(posts * heads * S), f(M), f(M)
        _lib.check(_lib.lib().mmhip_op_self_att_block_fwd(oc.code, _p(x), _p(maskbias), _p(oc.weight_cat((wq, wk, wv))), _p(oc.bias_cat((bq, bk, bv))),
                                                          _p(oc.weight(wo)), _p(bo), _p(g), _p(b), eps, posts, S, heads, p_att, p_hid, seed,
                                                          _p(qkv), _p(att), _p(lse), _p(pre), _p(mean), _p(rstd), _p(y), _s()), "self_att_block_fwd")
        ctx.save_for_backward(x, maskbias, qkv, att, lse, pre, mean, rstd)
        ctx.params, ctx.cfg = (wq, bq, wk, bk, wv, bv, wo, bo, g, b), (posts, S, heads, p_att, p_hid, seed, oc)
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, maskbias, qkv, att, lse, pre, mean, rstd = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.params
        posts, S, heads, p_att, p_hid, seed, oc = ctx.cfg
        gv = oc.gview
        M, H = x.shape
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        dpre, datt, dqkv, dx = e(M, H), e(M, H), e(M, 3 * H), e(M, H)
        dd = e(M, H) if p_hid > 0 else dpre
        _lib.check(_lib.lib().mmhip_op_self_att_block_bwd(oc.code, _p(dy.contiguous()), _p(maskbias), _p(oc.weight_cat((wq, wk, wv), True)), _p(oc.weight(wo, True)),
                                                          _p(g), posts, S, heads, p_att, p_hid, seed, _p(qkv), _p(att), _p(lse), _p(pre), _p(mean), _p(rstd),
                                                          _p(gv[id(g)]), _p(gv[id(b)]), _p(dpre), _p(dd), _p(datt), _p(dqkv), _p(dx), _s()), "self_att_block_bwd")
        oc.tn(dd, att, gv[id(wo)], gv[id(bo)], M, H, H)
        for i, (w, bb) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
            oc.tn(dqkv, x, gv[id(w)], gv[id(bb)], M, H, H, lda=3 * H, a_ptr=dqkv.data_ptr() + i * H * dqkv.element_size())
        return (dx,) + (None,) * 19


class _CrossAttBlock(torch.autograd.Function):
    """LayerNorm(dropout(dense(attention(queries of x, keys / values of ctx))) + x) -- a cross-modality attention block of LXMERT as ONE
    autograd node and ONE native call per direction (mmhip_op_cross_att_block_fwd / _bwd).  Returns the block output; the gradient of
    both inputs comes back from the one backward call."""

    @staticmethod
    def forward(ctx, x, c, keybias, wq, bq, wk, bk, wv, bv, wo, bo, g, b, posts, Sq, Sk, heads, p_att, p_hid, seed, eps, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        H = x.shape[1]
        S = max(Sq, Sk)
        Mq, Mc, M = posts * Sq, posts * Sk, posts * S
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        f = lambda n: torch.empty(n, dtype=torch.float32, device=x.device)
        qkv, att, pre, y, lse, mean, rstd = e(M, 3 * H), e(M, H), e(Mq, H), e(Mq, H), f(posts * heads * S), f(Mq), f(Mq)
        tq, attq = (e(Mq, H), e(Mq, H)) if Sq < S else (None, None)
        tkv = e(Mc, 2 * H) if Sk < S else None
        _lib.check(_lib.lib().mmhip_op_cross_att_block_fwd(oc.code, _p(x), _p(c), _p(keybias), _p(oc.weight_cat((wq, wk, wv))), _p(oc.bias_cat((bq, bk, bv))),
                                                           _p(oc.weight(wo)), _p(bo), _p(g), _p(b), eps, posts, Sq, Sk, heads, p_att, p_hid, seed,
                                                           _p(qkv), _p(att), _p(lse), _p(tq), _p(tkv), _p(attq), _p(pre), _p(mean), _p(rstd), _p(y), _s()),
                   "cross_att_block_fwd")
        ctx.save_for_backward(x, c, keybias, qkv, att, lse, pre, mean, rstd)
        ctx.attq = attq
        ctx.params, ctx.cfg = (wq, bq, wk, bk, wv, bv, wo, bo, g, b), (posts, Sq, Sk, heads, p_att, p_hid, seed, oc)
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, c, keybias, qkv, att, lse, pre, mean, rstd = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, g, b = ctx.params
        posts, Sq, Sk, heads, p_att, p_hid, seed, oc = ctx.cfg
        gv = oc.gview
        H = x.shape[1]
        S = max(Sq, Sk)
        Mq, Mc, M = posts * Sq, posts * Sk, posts * S
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        dpre, datt, dqkv, dxq, dxc = e(Mq, H), e(M, H), e(M, 3 * H), e(Mq, H), e(Mc, H)
        dd = e(Mq, H) if p_hid > 0 else dpre
        dattq, dq = (e(Mq, H), e(Mq, H)) if Sq < S else (None, None)
        dkv = e(Mc, 2 * H) if Sk < S else None
        _lib.check(_lib.lib().mmhip_op_cross_att_block_bwd(oc.code, _p(dy.contiguous()), _p(keybias), _p(oc.weight_cat((wq, wk, wv), True)), _p(oc.weight(wo, True)),
                                                           _p(g), posts, Sq, Sk, heads, p_att, p_hid, seed, _p(qkv), _p(att), _p(lse), _p(pre), _p(mean), _p(rstd),
                                                           _p(gv[id(g)]), _p(gv[id(b)]), _p(dpre), _p(dd), _p(dattq), _p(datt), _p(dqkv), _p(dq), _p(dkv), _p(dxq),
                                                           _p(dxc), _s()), "cross_att_block_bwd")
        es = dqkv.element_size()
        oc.tn(dd, ctx.attq if Sq < S else att, gv[id(wo)], gv[id(bo)], Mq, H, H)
        if Sq < S:
            oc.tn(dq, x, gv[id(wq)], gv[id(bq)], Mq, H, H)
        else:
            oc.tn(dqkv, x, gv[id(wq)], gv[id(bq)], Mq, H, H, lda=3 * H)
        for i, (w, bb) in enumerate(((wk, bk), (wv, bv))):
            if Sk < S:
                oc.tn(dkv, c, gv[id(w)], gv[id(bb)], Mc, H, H, lda=2 * H, a_ptr=dkv.data_ptr() + i * H * es)
            else:
                oc.tn(dqkv, c, gv[id(w)], gv[id(bb)], Mc, H, H, lda=3 * H, a_ptr=dqkv.data_ptr() + (1 + i) * H * es)
        return (dxq, dxc) + (None,) * 20


class _FFNBlock(torch.autograd.Function):
    """LayerNorm(dropout(W2 GELU(W1 x + b1) + b2) + x) as ONE autograd node and ONE native call per direction (mmhip_op_ffn_block_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, g, b, p_hid, seed, eps, oc):
        ctx.stream = _s()                  # the backward runs where the forward ran (autograd orders the streams on that assumption)
        M, H = x.shape
        I = w1.shape[0]
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        f = lambda n: torch.empty(n, dtype=torch.float32, device=x.device)
        h, u, pre, y, mean, rstd = e(M, I), e(M, I), e(M, H), e(M, H), f(M), f(M)
        _lib.check(_lib.lib().mmhip_op_ffn_block_fwd(oc.code, _p(x), _p(oc.weight(w1)), _p(b1), _p(oc.weight(w2)), _p(b2), _p(g), _p(b), eps, M, H, I, p_hid, seed,
                                                     _p(h), _p(u), _p(pre), _p(mean), _p(rstd), _p(y), _s()), "ffn_block_fwd")
        ctx.save_for_backward(x, h, u, pre, mean, rstd)
        ctx.params, ctx.cfg = (w1, b1, w2, b2, g, b), (p_hid, seed, oc)
        return y

    @staticmethod
    def backward(ctx, dy):
        _STREAM[0] = ctx.stream
        x, h, u, pre, mean, rstd = ctx.saved_tensors
        w1, b1, w2, b2, g, b = ctx.params
        p_hid, seed, oc = ctx.cfg
        gv = oc.gview
        M, H = x.shape
        I = w1.shape[0]
        e = lambda *shape: torch.empty(*shape, dtype=oc.tdt, device=x.device)
        dpre, du, dx = e(M, H), e(M, I), e(M, H)
        dd = e(M, H) if p_hid > 0 else dpre
        _lib.check(_lib.lib().mmhip_op_ffn_block_bwd(oc.code, _p(dy.contiguous()), _p(oc.weight(w1, True)), _p(oc.weight(w2, True)), _p(g), M, H, I, p_hid, seed,
                                                     _p(u), _p(pre), _p(mean), _p(rstd), _p(gv[id(g)]), _p(gv[id(b)]), _p(dpre), _p(dd), _p(du), _p(dx), _s()),
                   "ffn_block_bwd")
        oc.tn(dd, h, gv[id(w2)], gv[id(b2)], M, H, I)
        oc.tn(du, x, gv[id(w1)], gv[id(b1)], M, I, H)
        return (dx,) + (None,) * 10

class _Node(nn.Module):
    pass


class Lxmert(nn.Module):
    """reference models/mm_early.py:105-172.  `arch`: l_layers / r_layers / x_layers / vocab / max_pos / type_vocab (defaults: HF
    lxmert-base-uncased: 9 / 5 / 5 / 30522 / 512 / 2); weights are random-init unless `model_dir` holds a saved LxmertModel."""

    def __init__(self, model_dir, num_labels, max_length=None, dropout=0.1, logit_scale_init_value=2.6592, arch=None, dtype="bf16", seed=0):
        super().__init__()
        a = dict(hidden=768, heads=12, inter=3072, l_layers=9, r_layers=5, x_layers=5, vocab=30522, max_pos=512, type_vocab=2, feat_dim=2048,
                 pos_dim=4, p_hidden=0.1, p_attn=0.1, ln_eps=1e-12)
        a.update(arch or {})
        if a["hidden"] != a["heads"] * 64:
            raise ValueError("the attention kernels serve 64-wide heads")
        if not torch.cuda.is_available():
            raise RuntimeError("mm_early.Lxmert runs on the HIP kernels only (no CPU path)")
        self.arch, self.num_labels, self.p_head, self.dtype_name = a, num_labels, dropout, dtype
        self.oc = _Ctx(dtype)
        self.device_ = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        g = torch.Generator(device="cpu").manual_seed(seed)
        shapes = self.param_shapes(a, num_labels)
        # one flat fp32 buffer for the parameters, one for the gradients (every parameter 16-byte aligned): AdamW is a few launches
        # over flat ranges, and the operators add weight gradients straight into the gradient slices
        order, seen = [], set()
        for name in shapes:                                   # Q / K / V weights (and biases) of a block adjacent: [Wq; Wk; Wv] is a VIEW of the flat buffer
            if name in seen:
                continue
            if name.endswith(".query.weight"):
                stem = name[: -len("query.weight")]
                group = [stem + "query.weight", stem + "key.weight", stem + "value.weight", stem + "query.bias", stem + "key.bias", stem + "value.bias"]
                order += group
                seen.update(group)
            else:
                order.append(name)
                seen.add(name)
        offs, total = {}, 0
        for name in order:
            shape = shapes[name]
            offs[name] = total
            total += (int(np.prod(shape, dtype=np.int64)) if shape else 1) + 3 & ~3
        self._flat = torch.zeros(total, dtype=torch.float32, device=self.device_)
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device_)
        self._offs, self._shapes = offs, shapes
        for name, shape in shapes.items():
            if name == "logit_scale":
                val = torch.ones([]) * logit_scale_init_value
            elif name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
                val = torch.ones(shape)
            elif name.endswith(".bias"):
                val = torch.zeros(shape)
            else:
                val = torch.randn(shape, generator=g) * 0.02
            n = val.numel()
            view = self._flat[offs[name]: offs[name] + n].view(shape)
            view.copy_(val)
            node, parts = self, name.split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], nn.Parameter(view))
        self._attach_grads()
        self._build_persistent()
        if model_dir and os.path.isdir(model_dir):
            self._load_hf(model_dir)

    def _build_persistent(self):
        """activation-typed copy + transposed copy of every weight matrix (and of the fused [Wq; Wk; Wv] views), refreshed by ONE grouped
        native call per step instead of ~230 cast launches"""
        oc, H = self.oc, self.arch["hidden"]
        named = dict(self.named_parameters())
        mats = []

        def add(key, src):
            N, K = src.shape
            if N % 4 or K % 4:
                return
            dst = torch.empty(N, K, dtype=oc.tdt, device=src.device)
            dst_t = torch.empty(K, N, dtype=oc.tdt, device=src.device)
            oc.persistent[key] = (dst, dst_t)
            mats.append((src, dst, dst_t))

        for name, p in named.items():
            if p.dim() == 2 and "embeddings" not in name:
                add(id(p), p.detach())
            if name.endswith(".query.weight"):
                stem = name[: -len("query.weight")]
                trio = (named[stem + "query.weight"], named[stem + "key.weight"], named[stem + "value.weight"])
                o = self._offs[name]
                add(tuple(id(t) for t in trio), self._flat[o: o + 3 * H * H].view(3 * H, H))
                ob = self._offs[stem + "query.bias"]
                oc.persistent[("b",) + tuple(id(named[stem + n]) for n in ("query.bias", "key.bias", "value.bias"))] = self._flat[ob: ob + 3 * H]
        arr = (_lib.CastMat * len(mats))()
        for i, (src, dst, dst_t) in enumerate(mats):
            arr[i] = _lib.CastMat(src.data_ptr(), dst.data_ptr(), dst_t.data_ptr(), src.shape[0], src.shape[1])
        self._cast_arr, self._cast_keep, self._wsig = arr, mats, None

    def refresh_weights(self, force=False):
        sig = sum(p._version for p in self._pcache_list())
        if force or sig != self._wsig:
            _lib.check(_lib.lib().mmhip_op_cast_group(self.oc.code, C.cast(self._cast_arr, C.c_void_p), len(self._cast_keep), _s()), "cast_group")
            self.oc.cache.clear()
            self._wsig = sig

    def _pcache_list(self):
        try:
            return self._plist
        except AttributeError:
            object.__setattr__(self, "_plist", list(self.parameters()))
            return self._plist

    def _attach_grads(self):
        """every parameter's .grad is its slice of the flat gradient buffer (autograd accumulates in place for the few parameters torch ops
        consume: embeddings, logit_scale; the HIP operators add into the slices themselves)"""
        self.oc.gview = {}
        for name, p in self.named_parameters():
            n = p.numel()
            gv = self._flat_grad[self._offs[name]: self._offs[name] + n].view(p.shape)
            p.grad = gv
            self.oc.gview[id(p)] = gv

    def finish_backward(self):
        """launch the weight-gradient products queued by the backward pass (call after loss.backward(), before reading gradients)"""
        _refresh_stream()                  # the backward nodes left their own streams in the cache; autograd has joined them into this one
        self.oc.flush_tn()

    def zero_grad(self, set_to_none=False):
        self._flat_grad.zero_()
        self._attach_grads()

    def grad_ranges(self, use_itc, use_itm):
        """[begin, end) element ranges of the flat buffers that receive a gradient for this flag set: never the pooler (mm_early.py:132
        takes the CLS row itself), linear_tim only with ITM, logit_scale only with ITC -- torch's AdamW skips `grad is None` tensors"""
        key = (bool(use_itc), bool(use_itm))
        cached = getattr(self, "_ranges", {}).get(key)
        if cached is not None:
            return cached
        spans = []
        for name, shape in self._shapes.items():
            if name.startswith("model.pooler.") or (name.startswith("linear_tim.") and not use_itm) or (name == "logit_scale" and not use_itc):
                continue
            b = self._offs[name]
            spans.append((b, b + ((int(np.prod(shape, dtype=np.int64)) if shape else 1) + 3 & ~3)))
        out = []
        for b, e in sorted(spans):          # in ADDRESS order (the fused Q/K/V groups are laid out apart from the naming order): a handful of ranges, not ~150
            if out and out[-1][1] == b:
                out[-1][1] = e
            else:
                out.append([b, e])
        if not hasattr(self, "_ranges"):
            object.__setattr__(self, "_ranges", {})
        self._ranges[key] = out
        return out

    @staticmethod
    def param_shapes(a, num_labels):
        H, I = a["hidden"], a["inter"]
        s = {}

        def lin(n, o, i):
            s[n + ".weight"], s[n + ".bias"] = (o, i), (o,)

        def ln(n):
            s[n + ".weight"], s[n + ".bias"] = (H,), (H,)

        def att_block(n, inner):
            for p in ("query", "key", "value"):
                lin(f"{n}.{inner}.{p}", H, H)
            lin(f"{n}.output.dense", H, H)
            ln(f"{n}.output.LayerNorm")

        def ffn(i_, o_):
            lin(i_ + ".dense", I, H)
            lin(o_ + ".dense", H, I)
            ln(o_ + ".LayerNorm")

        e = "model.embeddings."
        s[e + "word_embeddings.weight"] = (a["vocab"], H)
        s[e + "position_embeddings.weight"] = (a["max_pos"], H)
        s[e + "token_type_embeddings.weight"] = (a["type_vocab"], H)
        ln(e + "LayerNorm")
        v = "model.encoder.visn_fc."
        lin(v + "visn_fc", H, a["feat_dim"]); ln(v + "visn_layer_norm")
        lin(v + "box_fc", H, a["pos_dim"]); ln(v + "box_layer_norm")
        for kind, n in (("layer", a["l_layers"]), ("r_layers", a["r_layers"])):
            for i in range(n):
                b = f"model.encoder.{kind}.{i}."
                att_block(b + "attention", "self")
                ffn(b + "intermediate", b + "output")
        for i in range(a["x_layers"]):
            b = f"model.encoder.x_layers.{i}."
            att_block(b + "visual_attention", "att")
            att_block(b + "lang_self_att", "self")
            att_block(b + "visn_self_att", "self")
            ffn(b + "lang_inter", b + "lang_output")
            ffn(b + "visn_inter", b + "visn_output")
        lin("model.pooler.dense", H, H)
        lin("linear_fusion", H, H)
        lin("linear", num_labels, H)
        lin("linear_tim", 2, H)
        s["logit_scale"] = ()
        return s

    def _load_hf(self, model_dir):
        """a saved LxmertModel directory (LxmertModel.from_pretrained layout): safetensors or pytorch_model.bin"""
        sd = None
        st = os.path.join(model_dir, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        elif os.path.exists(os.path.join(model_dir, "pytorch_model.bin")):
            sd = torch.load(os.path.join(model_dir, "pytorch_model.bin"), map_location="cpu")
        if sd is None:
            raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {model_dir!r}")
        sd = {("model." + k[len("lxmert."):] if k.startswith("lxmert.") else ("model." + k)): v for k, v in sd.items()}
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                if k in own and tuple(own[k].shape) == tuple(v.shape):
                    own[k].copy_(v.to(own[k].device, torch.float32))

    # ---- operators
    def _P(self, name):
        try:
            return self._pcache[name]
        except (AttributeError, KeyError):
            object.__setattr__(self, "_pcache", dict(self.named_parameters()))
            return self._pcache[name]

    def _lin(self, x, n):
        """Linear on [rows, K]; the fast GEMM wants N, K multiples of 8 (the 3- / 2-wide heads and the 4-wide box input take the generic kernel)"""
        return _Linear.apply(x.contiguous(), self._P(n + ".weight"), self._P(n + ".bias"), self.oc)

    def _ln(self, x, n):
        return _LayerNorm.apply(x.contiguous(), self._P(n + ".weight"), self._P(n + ".bias"), self.arch["ln_eps"], self.oc)

    def _drop(self, x, p):
        return F.dropout(x, p, self.training) if p > 0 else x

    def _attend(self, q_in, ctx_in, ctx_bias, n, B, Sq, Sk):
        """LxmertAttention(q_in, ctx_in): q_in [B*Sq, H], ctx_in [B*Sk, H], ctx_bias [B, Sk] additive key mask -> [B*Sq, H]"""
        H, nh = self.arch["hidden"], self.arch["heads"]
        S = max(Sq, Sk)
        p = self.arch["p_attn"] if self.training else 0.0
        if q_in is ctx_in:                                   # self-attention: one GEMM writes the packed [rows, 3H] tensor
            qkv = _LinearQKV.apply(q_in.contiguous(), self._P(n + ".query.weight"), self._P(n + ".query.bias"), self._P(n + ".key.weight"),
                                   self._P(n + ".key.bias"), self._P(n + ".value.weight"), self._P(n + ".value.bias"), self.oc)
            return _Attention.apply(qkv, ctx_bias.contiguous(), B, S, nh, p, self.oc.next_seed() if p > 0 else 0, self.oc)
        q = self._lin(q_in, n + ".query").view(B, Sq, H)
        k = self._lin(ctx_in, n + ".key").view(B, Sk, H)
        v = self._lin(ctx_in, n + ".value").view(B, Sk, H)
        pad = lambda t, L: t if L == S else F.pad(t, (0, 0, 0, S - L))
        qkv = torch.cat([pad(q, Sq), pad(k, Sk), pad(v, Sk)], dim=2).view(B * S, 3 * H)
        bias = ctx_bias if Sk == S else F.pad(ctx_bias, (0, S - Sk), value=float("-inf"))
        out = _Attention.apply(qkv, bias.contiguous(), B, S, nh, p, self.oc.next_seed() if p > 0 else 0, self.oc)
        return out.view(B, S, H)[:, :Sq].reshape(B * Sq, H)

    def _att_block(self, n, inner, x, ctx, ctx_bias, B, Sq, Sk):
        if x is ctx and os.environ.get("MMHIP_EARLY_FUSED", "1") != "0":          # self-attention: the whole block is one autograd node
            P, a = self._P, self.arch
            p_att = a["p_attn"] if self.training else 0.0
            q = f"{n}.{inner}."
            return _SelfAttBlock.apply(x.contiguous(), ctx_bias.contiguous(), P(q + "query.weight"), P(q + "query.bias"), P(q + "key.weight"), P(q + "key.bias"),
                                       P(q + "value.weight"), P(q + "value.bias"), P(n + ".output.dense.weight"), P(n + ".output.dense.bias"),
                                       P(n + ".output.LayerNorm.weight"), P(n + ".output.LayerNorm.bias"), B, Sq, a["heads"], p_att,
                                       a["p_hidden"] if self.training else 0.0, self.oc.next_seed() if self.training else 0, a["ln_eps"], self.oc)
        if x is not ctx and os.environ.get("MMHIP_EARLY_FUSED", "1") != "0":       # cross attention: one node as well
            P, a = self._P, self.arch
            S = max(Sq, Sk)
            q = f"{n}.{inner}."
            kb = ctx_bias if Sk == S else self._padded_bias(ctx_bias, S)
            return _CrossAttBlock.apply(x.contiguous(), ctx.contiguous(), kb.contiguous(), P(q + "query.weight"), P(q + "query.bias"), P(q + "key.weight"),
                                        P(q + "key.bias"), P(q + "value.weight"), P(q + "value.bias"), P(n + ".output.dense.weight"), P(n + ".output.dense.bias"),
                                        P(n + ".output.LayerNorm.weight"), P(n + ".output.LayerNorm.bias"), B, Sq, Sk, a["heads"],
                                        a["p_attn"] if self.training else 0.0, a["p_hidden"] if self.training else 0.0,
                                        self.oc.next_seed() if self.training else 0, a["ln_eps"], self.oc)
        a = self._attend(x, ctx, ctx_bias, f"{n}.{inner}", B, Sq, Sk)
        return self._ln(self._drop(self._lin(a, n + ".output.dense"), self.arch["p_hidden"]) + x, n + ".output.LayerNorm")

    def _padded_bias(self, bias, S):
        """[B, Sk] additive key mask -> [B, S], keys past Sk masked; one pad per mask tensor and forward (the five cross layers share it)"""
        key = (id(bias), S)
        hit = getattr(self, "_bias_pad", None)
        if hit is None or hit[0] != key or hit[1] is not bias:
            object.__setattr__(self, "_bias_pad", (key, bias, F.pad(bias, (0, S - bias.shape[1]), value=float("-inf")).contiguous()))
        return self._bias_pad[2]

    def _ffn(self, i_, o_, x):
        if os.environ.get("MMHIP_EARLY_FUSED", "1") != "0":
            P = self._P
            return _FFNBlock.apply(x.contiguous(), P(i_ + ".dense.weight"), P(i_ + ".dense.bias"), P(o_ + ".dense.weight"), P(o_ + ".dense.bias"),
                                   P(o_ + ".LayerNorm.weight"), P(o_ + ".LayerNorm.bias"), self.arch["p_hidden"] if self.training else 0.0,
                                   self.oc.next_seed() if self.training else 0, self.arch["ln_eps"], self.oc)
        y = _FFN.apply(x.contiguous(), self._P(i_ + ".dense.weight"), self._P(i_ + ".dense.bias"), self._P(o_ + ".dense.weight"), self._P(o_ + ".dense.bias"), self.oc)
        return self._ln(self._drop(y, self.arch["p_hidden"]) + x, o_ + ".LayerNorm")

    def encode(self, ids, mask, token_type_ids, features, boxes):
        """HF LxmertModel.forward -> (language_output [B,T,H], vision_output [B,36,H])"""
        a, oc = self.arch, self.oc
        B, T = ids.shape
        Nb = features.shape[1]
        if max(T, Nb) > 128:
            raise ValueError("max(text length, boxes) <= 128 (attention backward)")
        e = "model.embeddings."
        tt = torch.zeros_like(ids) if token_type_ids is None else token_type_ids
        emb = lambda n, idx: F.embedding(idx, self._P(e + n), padding_idx=0)      # HF: padding_idx=0 on all three tables
        x = emb("word_embeddings.weight", ids) + emb("position_embeddings.weight", torch.arange(T, device=ids.device))[None] + emb("token_type_embeddings.weight", tt)
        lang = self._drop(self._ln(x.to(oc.tdt).view(B * T, -1), e + "LayerNorm"), a["p_hidden"])
        v = "model.encoder.visn_fc."
        f = self._ln(self._lin(features.to(oc.tdt).reshape(B * Nb, -1).contiguous(), v + "visn_fc"), v + "visn_layer_norm")
        bx = self._ln(self._lin(boxes.to(oc.tdt).reshape(B * Nb, -1).contiguous(), v + "box_fc"), v + "box_layer_norm")
        visn = self._drop((f + bx) / 2, a["p_hidden"])
        lbias = torch.where(mask.bool(), 0.0, float("-inf")).to(torch.float32).contiguous()
        vbias = torch.zeros(B, Nb, dtype=torch.float32, device=ids.device)
        # The language and the vision stream are independent between their meeting points (the cross attentions): the vision stream's
        # operators -- 36 rows per post, launches that fill a fifth of the chip -- run on a second HIP stream beside the language stream's
        # (MMHIP_EARLY_STREAMS=0: one stream).  The backward follows by itself: every node runs on its forward's stream.
        main = torch.cuda.current_stream()
        two = lang.is_cuda and os.environ.get("MMHIP_EARLY_STREAMS", "1") != "0"
        side = self._side_stream() if two else None
        if side is not None:
            side.wait_stream(main)
            for t in (visn, vbias, lbias):            # made on the caller's stream, read on the side stream: the allocator must not hand the
                t.record_stream(side)                 # block out again before the side stream is done with it
        with self._on(side):
            for i in range(a["r_layers"]):
                b = f"model.encoder.r_layers.{i}."
                visn = self._ffn(b + "intermediate", b + "output", self._att_block(b + "attention", "self", visn, visn, vbias, B, Nb, Nb))
        for i in range(a["l_layers"]):
            b = f"model.encoder.layer.{i}."
            lang = self._ffn(b + "intermediate", b + "output", self._att_block(b + "attention", "self", lang, lang, lbias, B, T, T))
        for i in range(a["x_layers"]):
            b = f"model.encoder.x_layers.{i}."
            if side is not None:                      # each side needs the other's output of the previous layer
                main.wait_stream(side)
                side.wait_stream(main)
                lang.record_stream(side)
                visn.record_stream(main)
            la = self._att_block(b + "visual_attention", "att", lang, visn, vbias, B, T, Nb)      # ONE module, both directions
            la = self._att_block(b + "lang_self_att", "self", la, la, lbias, B, T, T)
            lang_next = self._ffn(b + "lang_inter", b + "lang_output", la)
            with self._on(side):
                va = self._att_block(b + "visual_attention", "att", visn, lang, lbias, B, Nb, T)
                va = self._att_block(b + "visn_self_att", "self", va, va, vbias, B, Nb, Nb)
                visn = self._ffn(b + "visn_inter", b + "visn_output", va)
            lang = lang_next
        if side is not None:
            main.wait_stream(side)
            visn.record_stream(main)                  # allocated on the side stream, consumed on the caller's (pooling, ITM slices)
        return lang.view(B, T, -1), visn.view(B, Nb, -1)

    def _side_stream(self):
        st = getattr(self, "_side", None)
        if st is None:
            st = torch.cuda.Stream(device=self.device_)
            object.__setattr__(self, "_side", st)
        return st

    def _on(self, stream):
        """context: torch's current stream = `stream` (None: unchanged), with the operators' cached stream handle following it"""
        import contextlib

        @contextlib.contextmanager
        def ctxm():
            if stream is None:
                yield
                return
            with torch.cuda.stream(stream):
                _refresh_stream()
                try:
                    yield
                finally:
                    pass
            _refresh_stream()
        return ctxm()

    def forward(self, ids, mask, token_type_ids, features, normalized_boxes, tim_inputs=None):
        """reference :121-163 -> (linear_output, max_embeddings_t, max_embeddings_v, out_tim), fp32"""
        dev = self.device_
        _refresh_stream()
        self.refresh_weights()
        ids, mask = ids.to(dev), mask.to(dev)
        tt = None if token_type_ids is None else token_type_ids.to(dev)
        features, boxes = features.to(dev, torch.float32), normalized_boxes.to(dev, torch.float32)
        B = ids.shape[0]
        x_t2 = None
        if tim_inputs is not None and os.environ.get("MMHIP_EARLY_ITM_BATCHED", "1") != "0":
            # ITM (reference :146-161 runs the whole encoder a second time on the swapped texts): both passes as ONE pass of 2B posts -- the same
            # per-post arithmetic (no operator mixes posts), half the launches, twice the rows per GEMM
            t_ids, t_mask, t_tt = tim_inputs
            t_tt = torch.zeros_like(t_ids) if t_tt is None else t_tt
            tt0 = torch.zeros_like(ids) if tt is None else tt
            x_t_all, x_v_all = self.encode(torch.cat([ids, t_ids.to(dev)]), torch.cat([mask, t_mask.to(dev)]), torch.cat([tt0, t_tt.to(dev)]),
                                           torch.cat([features, features]), torch.cat([boxes, boxes]))
            x_t, x_v, x_t2 = x_t_all[:B], x_v_all[:B], x_t_all[B:]
        else:
            x_t, x_v = self.encode(ids, mask, tt, features, boxes)
        xt = torch.relu(self._lin(x_t[:, 0].contiguous(), "linear_fusion"))
        out = self._lin(self._drop(xt, self.p_head).contiguous(), "linear").float()
        last = x_t.detach().float().clone()                        # :139-143: no gradient into the text embedding
        last[mask.unsqueeze(-1).expand(last.shape) == 0] = -1e9
        emb_t = last.max(1)[0]
        emb_v = x_v.float().max(1)[0]
        out_tim = None
        if tim_inputs is not None:
            if x_t2 is None:
                t_ids, t_mask, t_tt = tim_inputs
                x_t2, _ = self.encode(t_ids.to(dev), t_mask.to(dev), None if t_tt is None else t_tt.to(dev), features, boxes)
            out_tim = self._lin(x_t2[:, 0].contiguous(), "linear_tim").float()
        return out, emb_t, emb_v, out_tim

    def get_logits_per_text(self, text_embeds, image_embeds):
        """reference :165-172"""
        image_embeds = image_embeds / image_embeds.norm(p=2, dim=-1, keepdim=True)
        text_embeds = text_embeds / text_embeds.norm(p=2, dim=-1, keepdim=True)
        return torch.matmul(text_embeds, image_embeds.t()) * self._P("logit_scale").exp()


class MMEarly_Model(object):
    """reference models/mm_early.py:175-520, LXMERT branch: loss mixing :366-379, ITM sampling (same numpy stream as mm_late),
    AdamW over every parameter that received a gradient (the pooler never does: torch skips `grad is None`)."""

    def __init__(self, config, model_name="lxmert", multilabel=False, **model_kw):
        if model_name != "lxmert":
            raise NotImplementedError("early fusion: only the LXMERT branch (BASELINE config 5); ViLT is out of scope (SURVEY.md 2)")
        if multilabel:
            raise NotImplementedError("multilabel BCE branch: no task enables it in the reference")
        self.batch_size, self.num_labels = config.batch_size, config.num_labels
        self.use_clip_loss, self.beta_itc = config.use_clip_loss, config.beta_itc
        self.use_tim_loss, self.beta_itm = config.use_tim_loss, config.beta_itm
        self.max_length = config.max_length
        self.model = Lxmert(model_kw.pop("model_dir", None), self.num_labels, self.max_length, dropout=config.dropout, **model_kw)
        self.device = self.model.device_
        self._opt = None

    def prepare_itm_inputs(self, ids, mask, token_type_ids=None):
        """reference :300-330 (same draws as mm_late.prepare_itm_inputs, plus the token type ids of the swapped rows)"""
        B = ids.shape[0]
        src, labels = list(range(B)), [1] * B
        if B > 1:
            for idx in range(B):
                if np.random.choice(2) == 0:
                    labels[idx] = 0
                    j = int(np.random.choice(B - 1))
                    src[idx] = j if j < idx else j + 1
        sel = torch.tensor(src, device=ids.device)
        tt = None if token_type_ids is None else token_type_ids.index_select(0, sel)
        return ids.index_select(0, sel), mask.index_select(0, sel), tt, torch.tensor(labels, device=self.device)

    def loss(self, out, onehot, class_weight, emb_t, emb_v, out_tim, lbl_tim):
        """reference :366-379"""
        label = onehot.to(out.device).type_as(out)
        lc = F.cross_entropy(out, label, weight=None if class_weight is None else class_weight.to(out.device, torch.float32))
        bi = self.beta_itc if self.use_clip_loss else 0.0
        bm = self.beta_itm if self.use_tim_loss else 0.0
        total = (1 - (bi + bm)) * lc
        if self.use_clip_loss:
            total = total + bi * clip_loss(self.model.get_logits_per_text(emb_t, emb_v))
        if self.use_tim_loss:
            total = total + bm * F.cross_entropy(out_tim, lbl_tim)
        return total

    def train_step(self, ids, mask, token_type_ids, features, boxes, onehot, class_weight, lr, weight_decay, step):
        m = self.model
        if not m.training:
            m.train()                                          # walks ~1500 submodules: only on a mode change
        dev = self.device
        ids, mask = ids.to(dev), mask.to(dev)
        tt = None if token_type_ids is None else token_type_ids.to(dev)
        tim, lbl = None, None
        if self.use_tim_loss:
            t_ids, t_mask, t_tt, lbl = self.prepare_itm_inputs(ids, mask, tt)
            tim = (t_ids, t_mask, t_tt)
        out, et, ev, otim = m(ids, mask, tt, features, boxes, tim_inputs=tim)
        loss = self.loss(out, onehot, class_weight, et, ev, otim, lbl)
        loss.backward()
        m.finish_backward()
        lib = _lib.lib()
        world = mmdist.world_size()
        if world > 1:                                          # data parallel: one all-reduce of the flat gradient (RCCL: backend "nccl"); AdamW averages
            torch.distributed.all_reduce(m._flat_grad)
        if self._opt is None or not isinstance(self._opt, tuple):
            self._opt = (torch.zeros_like(m._flat), torch.zeros_like(m._flat))
        at = lambda t, el: t.data_ptr() + el * 4
        for b, e in m.grad_ranges(self.use_clip_loss, self.use_tim_loss):
            _lib.check(lib.mmhip_adamw(at(m._flat, b), at(m._flat_grad, b), at(self._opt[0], b), at(self._opt[1], b), e - b, lr, 0.9, 0.999, 1e-8,
                                       weight_decay, step, 1.0 / world, 1, _s()), "adamw")  # zero_grad fused: the slices are clean for the next step
        m._wsig = None                                         # the kernels updated the weights through raw pointers: the copies are refreshed by the next forward
        m.oc.cache.clear()
        return loss.detach()

    def load_saved_model(self, model_path):
        self.model.load_state_dict(torch.load(model_path, map_location=self.device))

    def save_model(self, model_path):
        torch.save(self.model.state_dict(), model_path)

    def train(self, dataloader, val_dataloader, epochs, loss_fn=None, lr=1e-5, weight_decay=0.00025, tim_loss_fn=None, te_dataloader=None,
              model_path=None, val_filename=None, te_filename=None, class_weight=None, log_every=50):
        """reference :332-428 (LXMERT branch): epochs of train steps, validation / test metrics CSVs every even epoch and at the end,
        checkpoint = plain state_dict with the reference's keys"""
        import pandas as pd
        if class_weight is None and loss_fn is not None and getattr(loss_fn, "weight", None) is not None:
            class_weight = loss_fn.weight
        res_val, res_te, step = [], [], 0
        for epoch in range(epochs):
            if mmdist.rank() == 0:
                print("Epoch:", epoch + 1)
            if hasattr(getattr(dataloader, "sampler", None), "set_epoch"):
                dataloader.sampler.set_epoch(epoch)
            for it, b in enumerate(dataloader):
                step += 1
                sq = lambda t: t.squeeze(1) if t.dim() == 3 else t
                loss = self.train_step(sq(b["input_ids"]), sq(b["attention_mask"]), sq(b["token_type_ids"]) if "token_type_ids" in b else None,
                                       b["features"], b["normalized_boxes"], b["labels"], class_weight, lr, weight_decay, step)
                if log_every and it % log_every == 0 and mmdist.rank() == 0:
                    print(f"loss {float(loss):.4f}")
            for loader, store, fname in ((val_dataloader, res_val, val_filename), (te_dataloader, res_te, te_filename)):
                if loader is None:
                    continue
                r = self.eval(loader, class_weight=class_weight)
                r["epoch"] = epoch
                store.append(r)
                if fname is not None and (epoch % 2 == 0 or epoch == epochs - 1) and mmdist.rank() == 0:
                    pd.DataFrame(agg_metrics_val(store, metric_names, self.num_labels)).to_csv(fname, index=False)
        if model_path is not None and mmdist.rank() == 0:
            self.save_model(model_path)

    def eval(self, batches, loss_fn=None, tim_loss_fn=None, class_weight=None):
        """reference :430-520 for the LXMERT branch: eval-mode forward, loss mix with re-sampled ITM negatives, argmax"""
        m = self.model
        if class_weight is None and loss_fn is not None and getattr(loss_fn, "weight", None) is not None:
            class_weight = loss_fn.weight
        m.eval()
        preds, labels, ids_all, losses = [], [], [], []
        sq = lambda t: t.squeeze(1) if t.dim() == 3 else t
        with torch.no_grad():
            for b in batches:
                ids, mask = sq(b["input_ids"]).to(self.device), sq(b["attention_mask"]).to(self.device)
                tt = sq(b["token_type_ids"]).to(self.device) if "token_type_ids" in b else None
                tim, lbl = None, None
                if self.use_tim_loss:
                    t_ids, t_mask, t_tt, lbl = self.prepare_itm_inputs(ids, mask, tt)
                    tim = (t_ids, t_mask, t_tt)
                out, et, ev, otim = m(ids, mask, tt, b["features"], b["normalized_boxes"], tim_inputs=tim)
                losses.append(float(self.loss(out, b["labels"], class_weight, et, ev, otim, lbl)))
                preds.append(out.argmax(1).cpu()); labels.append(b["labels"].argmax(1).cpu())
                if "data_id" in b:
                    ids_all.append(b["data_id"])
        return {"data_id": torch.cat(ids_all).numpy() if ids_all else np.zeros(0, dtype=np.int64), "loss": float(np.mean(losses)),
                "predictions": torch.cat(preds).numpy(), "labels": torch.cat(labels).numpy()}
