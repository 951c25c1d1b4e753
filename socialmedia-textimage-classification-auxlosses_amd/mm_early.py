"""Early-fusion LXMERT (BASELINE config 5) -- reference models/mm_early.py:105-172 (class Lxmert) and :175-520 (MMEarly_Model).

Round 4: a NATIVE engine (csrc/early.hip, include/mmhip.h `mmhip_early_*`), the counterpart of mm_late.py's: the whole step -- embeddings,
visual-feature encoder, 9 language + 5 relational + 5 cross-modality layers on two internal HIP streams, heads, max-pooled ITC embeddings,
ITC similarity, fused loss mix, backward, grouped weight gradients, AdamW, operand refresh -- is ONE C call (`mmhip_early_train_step`); no
torch autograd graph, no ATen kernels inside the step.  (Rounds 2-3 chained the library's block operators by ~40 autograd nodes per step
from Python, with embeddings / dropout / max-pool / ITC / losses as ATen kernels.)  `Lxmert.forward` keeps the reference's signature and
return tuple; a reference-style caller's `loss.backward()` works through one autograd edge around the engine (explicit output gradients into
`mmhip_early_backward`).  There is no CPU path: the module raises when the HIP library or the GPU is missing.

Cross attention (queries and keys of different lengths: T tokens x 36 boxes) runs on the self-attention kernels: Q, K and V are separate
column blocks of one packed [rows, 3H] tensor, S = max(T, boxes) rows per post, keys past the context length masked, query rows past the
query length discarded.  S <= 128 (the attention backward's limit) covers the reference's max_length = 128.

State-dict keys are the reference module's (`model.*` = HF LxmertModel 4.25.1 naming, `linear_fusion`, `linear`, `linear_tim`,
`logit_scale`).  Parity: tests/test_gpu_early.py against tests/golden/lxmert_small.npz (the reference's own module) and the oracle.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dist as mmdist
from .utils import clip_loss, agg_metrics_val
from .config import metric_names

_DT = {"bf16": _lib.BF16, "f16": _lib.F16, "bf16x3": _lib.BF16X3}


class _Node(nn.Module):
    pass


class _EarlyFunction(torch.autograd.Function):
    """autograd edge around the engine so that a reference-style caller's loss.backward() (models/mm_early.py:381) works"""

    @staticmethod
    def forward(ctx, model, ids, mask, tt, feats, boxes, tim, *params):
        out, et, ev, otim = model._engine_forward(ids, mask, tt, feats, boxes, tim)
        ctx.model, ctx.token, ctx.has_tim = model, model._fwd_token, tim is not None
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(et)                       # reference :139-143: the text embedding is detached
        return (out, et, ev, otim) if ctx.has_tim else (out, et, ev)

    @staticmethod
    def backward(ctx, *douts):
        model = ctx.model
        if ctx.token != model._fwd_token:
            raise RuntimeError("Lxmert: backward() after another forward(); the engine keeps one set of activations")
        d_out, _, d_ev = douts[:3]
        d_tim = douts[3] if ctx.has_tim else None
        grads = model._engine_backward_autograd(d_out, d_ev, d_tim)
        return (None,) * 7 + tuple(grads)


class Lxmert(nn.Module):
    """reference models/mm_early.py:105-172.  `arch`: l_layers / r_layers / x_layers / vocab / max_pos / type_vocab (defaults: HF
    lxmert-base-uncased: 9 / 5 / 5 / 30522 / 512 / 2); weights are random-init unless `model_dir` holds a saved LxmertModel.
    Additive keywords: `dtype` ('bf16' | 'f16' | 'bf16x3' = strict parity), `max_posts` / `max_text_len` / `max_boxes` (capacity the
    workspace is sized for; it grows on demand), `seed`."""

    def __init__(self, model_dir, num_labels, max_length=None, dropout=0.1, logit_scale_init_value=2.6592, arch=None, dtype="bf16", seed=0,
                 max_posts=8, max_boxes=36):
        super().__init__()
        a = dict(hidden=768, heads=12, inter=3072, l_layers=9, r_layers=5, x_layers=5, vocab=30522, max_pos=512, type_vocab=2, feat_dim=2048,
                 pos_dim=4, p_hidden=0.1, p_attn=0.1, ln_eps=1e-12)
        a.update(arch or {})
        if a["hidden"] != a["heads"] * 64:
            raise ValueError("the attention kernels serve 64-wide heads")
        if not torch.cuda.is_available():
            raise _lib.MMHipError("mm_early.Lxmert needs an MI355X (gfx950) GPU: the HIP path has no CPU fallback")
        self.arch, self.num_labels, self.p_head, self.dtype_name = a, num_labels, dropout, dtype
        self.device_ = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
        self._cfg_kw = dict(hidden=a["hidden"], heads=a["heads"], inter=a["inter"], l_layers=a["l_layers"], r_layers=a["r_layers"], x_layers=a["x_layers"],
                            vocab=a["vocab"], max_pos=a["max_pos"], type_vocab=a["type_vocab"], feat_dim=a["feat_dim"], pos_dim=a["pos_dim"], num_labels=num_labels,
                            dtype=_DT[dtype], p_hidden=a["p_hidden"], p_attn=a["p_attn"], p_head=dropout, ln_eps=a["ln_eps"])
        self._handle, self._ws, self._capacity = None, None, (0, 0, 0)
        self._seed_base, self._calls, self._fwd_token, self._weights_version, self._grad_dirty = 0x5DEECE66D + seed, 0, 0, None, False
        self._create_engine(max_posts, min(128, max(int(max_length or 16), 16)), max_boxes, first=True)
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for inf in self._infos:
                name, shape, p = inf["name"], inf["shape"], inf["param"]
                if name == "logit_scale":
                    val = torch.ones([]) * logit_scale_init_value
                elif name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
                    val = torch.ones(shape)
                elif name.endswith(".bias"):
                    val = torch.zeros(shape)
                else:
                    val = torch.randn(shape, generator=g) * 0.02
                p.copy_(val.to(self.device_))
        if model_dir and os.path.isdir(model_dir):
            self._load_hf(model_dir)

    # ------------------------------------------------------------------ engine plumbing
    def _create_engine(self, max_posts, max_text_len, max_boxes, first=False):
        lib = _lib.lib()
        cfg = _lib.EarlyConfig(max_posts=int(max_posts), max_text_len=int(max_text_len), max_boxes=int(max_boxes), **self._cfg_kw)
        h = C.c_void_p()
        _lib.check(lib.mmhip_early_create(C.byref(cfg), C.byref(h)), "early_create")
        if self._handle is not None:
            lib.mmhip_early_destroy(self._handle)
        self._handle = h
        self._capacity = (int(max_posts), int(max_text_len), int(max_boxes))
        dev = self.device_
        if first:
            n = int(lib.mmhip_early_numel(h))
            self._flat = torch.zeros(n, dtype=torch.float32, device=dev)
            self._flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
            self._infos, self._offs, self._shapes = [], {}, {}
            pi = _lib.ParamInfo()
            for i in range(lib.mmhip_early_param_count(h)):
                _lib.check(lib.mmhip_early_param_info_at(h, i, C.byref(pi)), "early_param_info")
                inf = dict(name=pi.name.decode(), shape=tuple(pi.dims[: pi.ndim]), group=pi.group, offset=int(pi.offset), numel=int(pi.numel))
                view = self._flat[inf["offset"]: inf["offset"] + inf["numel"]].view(inf["shape"])
                p = nn.Parameter(view)
                node, parts = self, inf["name"].split(".")
                for part in parts[:-1]:
                    if part not in node._modules:
                        node.add_module(part, _Node())
                    node = node._modules[part]
                node.register_parameter(parts[-1], p)
                inf["param"] = p
                self._infos.append(inf)
                self._offs[inf["name"]], self._shapes[inf["name"]] = inf["offset"], inf["shape"]
            self._attach_grads()
        if first:
            # include/mmhip.h mmhip_early_set_index_counter: token ids / token types that had to be clamped into their tables (the reference's
            # nn.Embedding raises IndexError for them; MMEarly_Model.check_indices does, at the end of an epoch / an evaluation loop)
            self._bad_index = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.mmhip_early_set_index_counter(h, _lib.ptr(self._bad_index)), "early_set_index_counter")
        self._ws = None
        torch.cuda.empty_cache()
        self._ws = torch.empty(int(lib.mmhip_early_workspace_bytes(h)), dtype=torch.uint8, device=dev)
        _lib.check(lib.mmhip_early_bind(h, _lib.ptr(self._flat), _lib.ptr(self._flat_grad), _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "early_bind")
        self._stage_ranges = []
        b, e = C.c_uint64(), C.c_uint64()
        for st in range(lib.mmhip_early_num_stages(h)):
            _lib.check(lib.mmhip_early_stage_grad_range(h, st, C.byref(b), C.byref(e)), "early_stage_grad_range")
            self._stage_ranges.append((int(b.value), int(e.value)))
        self._weights_version = None

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().mmhip_early_destroy(self._handle)
        except Exception:
            pass

    def _attach_grads(self):
        """every parameter's .grad is its slice of the flat gradient buffer the engine writes"""
        for inf in self._infos:
            inf["param"].grad = self._flat_grad[inf["offset"]: inf["offset"] + inf["numel"]].view(inf["shape"])

    def refresh_weights(self, force=False):
        if force or self._weights_version != self._flat._version:
            _lib.check(_lib.lib().mmhip_early_refresh_weights(self._handle, _lib.stream_ptr()), "early_refresh_weights")
            self._weights_version = self._flat._version

    def _ensure(self, B, T, Nb):
        cb, ct, cn = self._capacity
        if B > cb or T > ct or Nb > cn:
            if max(T, Nb) > 128:
                raise ValueError("max(text length, boxes) <= 128 (attention backward)")
            self._create_engine(max(B, cb), max(T, ct), max(Nb, cn))
        self.refresh_weights()

    def finish_backward(self):
        """kept for callers of the first version (the weight gradients used to be queued): the engine's backward is complete when it returns"""

    def zero_grad(self, set_to_none=False):
        self._flat_grad.zero_()
        self._attach_grads()
        self._grad_dirty = False

    def _P(self, name):
        try:
            return self._pcache[name]
        except (AttributeError, KeyError):
            object.__setattr__(self, "_pcache", dict(self.named_parameters()))
            return self._pcache[name]

    def grad_ranges(self, use_itc, use_itm):
        """[begin, end) element ranges of the flat buffers that receive a gradient for this flag set, merged in address order: never the
        pooler (mm_early.py:132 takes the CLS row itself), linear_tim only with ITM, logit_scale only with ITC -- torch's AdamW skips
        `grad is None` tensors.  (The engine's step walks the same ranges.)"""
        groups = {_lib.G_ALWAYS} | ({_lib.G_ITC} if use_itc else set()) | ({_lib.G_ITM} if use_itm else set())
        spans = sorted((i["offset"], i["offset"] + ((i["numel"] + 3) & ~3)) for i in self._infos if i["group"] in groups)
        out = []
        for b, e in spans:
            if out and out[-1][1] == b:
                out[-1][1] = e
            else:
                out.append([b, e])
        return out

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

    # ------------------------------------------------------------------ engine calls
    def _next_seed(self):
        self._calls += 1
        return (self._seed_base * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF

    def _inputs(self, ids, mask, tt, feats, boxes):
        dev = self.device_
        i64 = lambda t: None if t is None else t.to(dev, torch.int64).contiguous()
        ids, mask, tt = i64(ids), i64(mask), i64(tt)
        feats, boxes = feats.to(dev, torch.float32).contiguous(), boxes.to(dev, torch.float32).contiguous()
        if ids.dim() != 2 or feats.dim() != 3 or feats.shape[0] != ids.shape[0] or tuple(boxes.shape[:2]) != tuple(feats.shape[:2]):
            raise ValueError(f"Lxmert.forward: ids {tuple(ids.shape)}, features {tuple(feats.shape)}, boxes {tuple(boxes.shape)}")
        if feats.shape[2] != self.arch["feat_dim"] or boxes.shape[2] != self.arch["pos_dim"]:
            raise ValueError("feature / box width does not match the model")
        return ids, mask, tt, feats, boxes

    def _engine_forward(self, ids, mask, tt, feats, boxes, tim=None, seed=None):
        ids, mask, tt, feats, boxes = self._inputs(ids, mask, tt, feats, boxes)
        B, T = ids.shape
        Nb = feats.shape[1]
        self._ensure(B, T, Nb)
        dev = self.device_
        t_ids = t_mask = t_tt = None
        if tim is not None:
            i64 = lambda t: None if t is None else t.to(dev, torch.int64).contiguous()
            t_ids, t_mask, t_tt = i64(tim[0]), i64(tim[1]), i64(tim[2] if len(tim) > 2 else None)
        out = torch.empty(B, self.num_labels, device=dev)
        et, ev = torch.empty(B, self.arch["hidden"], device=dev), torch.empty(B, self.arch["hidden"], device=dev)
        otim = torch.empty(B, 2, device=dev) if tim is not None else None
        _lib.check(_lib.lib().mmhip_early_forward(self._handle, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(tt), _lib.ptr(feats), _lib.ptr(boxes), _lib.ptr(t_ids),
                                                  _lib.ptr(t_mask), _lib.ptr(t_tt), B, T, Nb, int(self.training), seed if seed is not None else self._next_seed(),
                                                  _lib.ptr(out), _lib.ptr(et), _lib.ptr(ev), _lib.ptr(otim), _lib.stream_ptr()), "early_forward")
        self._fwd_token += 1
        self._last = dict(B=B, T=T, Nb=Nb, itm=tim is not None)
        return out, et, ev, otim

    def _engine_backward_autograd(self, d_out, d_ev, d_tim):
        """explicit output gradients into the engine; the parameter gradients land in the flat gradient buffer, whose slices ARE the
        parameters' .grad (torch semantics: they add to what is there -- call zero_grad() per step as the reference's loop does; weight
        matrices are plain stores, include/mmhip.h mmhip_early_backward)"""
        B, dev = self._last["B"], self.device_
        f = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
        d_out = f(d_out) if d_out is not None else torch.zeros(B, self.num_labels, device=dev)
        d_ev, d_tim = f(d_ev), f(d_tim)
        _lib.check(_lib.lib().mmhip_early_backward(self._handle, _lib.ptr(d_out), _lib.ptr(d_ev), _lib.ptr(d_tim), _lib.stream_ptr()), "early_backward")
        self._grad_dirty = True          # the fused step finds its entry condition (zero gradient) re-established by _clean_grad
        return [None] * len(self._infos)          # (already in place: nothing for autograd to accumulate)

    def _clean_grad(self):
        if self._grad_dirty:
            self._flat_grad.zero_()
            self._grad_dirty = False

    # ------------------------------------------------------------------ reference interface
    def forward(self, ids, mask, token_type_ids, features, normalized_boxes, tim_inputs=None):
        """reference :121-163 -> (linear_output, max_embeddings_t, max_embeddings_v, out_tim), fp32"""
        if torch.is_grad_enabled() and self.training:
            params = [i["param"] for i in self._infos]
            outs = _EarlyFunction.apply(self, ids, mask, token_type_ids, features, normalized_boxes, tim_inputs, *params)
            return (outs[0], outs[1], outs[2], outs[3]) if tim_inputs is not None else (outs[0], outs[1], outs[2], None)
        return self._engine_forward(ids, mask, token_type_ids, features, normalized_boxes, tim_inputs)

    def get_logits_per_text(self, text_embeds, image_embeds):
        """reference :165-172 (on the [B, H] embeddings the engine returned: the reference-style loss path; the fused step computes it natively)"""
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
        model_kw.setdefault("max_posts", config.batch_size)
        self.model = Lxmert(model_kw.pop("model_dir", None), self.num_labels, self.max_length, dropout=config.dropout, **model_kw)
        self.device = self.model.device_
        self._opt = None
        self.adam_eps = 1e-8                  # torch.optim.AdamW's default, as the reference uses it

    def prepare_itm_inputs(self, ids, mask, token_type_ids=None):
        """reference :300-330 (same draws as mm_late.prepare_itm_inputs, plus the token type ids of the swapped rows)"""
        src, labels = self._itm_draw(ids.shape[0])
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

    def loss_weights(self):
        bi = self.beta_itc if self.use_clip_loss else 0.0
        bm = self.beta_itm if self.use_tim_loss else 0.0
        return 1.0 - (bi + bm), bi, bm

    def train_step(self, ids, mask, token_type_ids, features, boxes, onehot, class_weight, lr, weight_decay, step):
        """one fused training step on the engine (include/mmhip.h mmhip_early_train_step): forward (the ITM pass batched with the main pass),
        loss mix, backward, AdamW over the ranges that receive a gradient, operand refresh -- one native call, no torch kernels inside.
        Under data parallelism the engine calls back at every backward stage boundary and the stage ranges leave as bucketed all-reduces
        (dist.StageBuckets) beside the stages below; AdamW averages (grad_scale = 1 / world)."""
        m, lib = self.model, _lib.lib()
        if not m.training:
            m.train()
        m._clean_grad()
        ids, mask, tt, feats, bx = m._inputs(ids, mask, token_type_ids, features, boxes)
        B, T = ids.shape
        Nb = feats.shape[1]
        m._ensure(B, T, Nb)
        dev = self.device
        src = lbl = None
        if self.use_tim_loss:
            srcs, labels = self._itm_draw(B)
            both = torch.tensor(srcs + labels, dtype=torch.int64).to(dev, non_blocking=True)      # one host -> device copy
            src, lbl = both[:B], both[B:]
        onehot = onehot.to(dev, torch.int64).contiguous()
        cw = None if class_weight is None else class_weight.to(dev, torch.float32).contiguous()
        if self._opt is None or not isinstance(self._opt, tuple):
            self._opt = (torch.zeros_like(m._flat), torch.zeros_like(m._flat))
        w_cls, w_itc, w_itm = self.loss_weights()
        world = mmdist.world_size()
        loss = torch.empty(4, device=dev)
        args = (m._handle, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(tt), _lib.ptr(feats), _lib.ptr(bx), _lib.ptr(src), _lib.ptr(lbl), _lib.ptr(onehot), _lib.ptr(cw),
                B, T, Nb, m._next_seed(), int(self.use_clip_loss), int(self.use_tim_loss), w_cls, w_itc, w_itm, _lib.ptr(self._opt[0]), _lib.ptr(self._opt[1]),
                lr, 0.9, 0.999, self.adam_eps, weight_decay, step, 1.0 / world, _lib.ptr(loss), _lib.stream_ptr())
        if world > 1 or mmdist.force_exchange():
            buckets = mmdist.StageBuckets(m._flat_grad)
            failure = []

            def on_stage(_user, st):
                try:
                    if mmdist.SKIP_EXCHANGE:
                        return 0
                    if st >= 0:
                        b, e = m._stage_ranges[st]
                        buckets.add(b, e, flush=st == len(m._stage_ranges) - 1)
                    elif st == _lib.CB_WAIT_DENSE:
                        buckets.flush()
                        for w in buckets.works:
                            if w is not None:
                                w.wait()
                    return 0
                except BaseException as exc:              # a Python exception must not unwind through the C frames
                    failure.append(exc)
                    return -2
            cb = _lib.EXCHANGE_CB(on_stage)
            rc = lib.mmhip_early_train_step(*args, cb, None)
            if failure:
                raise failure[0]
            _lib.check(rc, "early_train_step")
            m._last_exchange_bytes = buckets.bytes
        else:
            _lib.check(lib.mmhip_early_train_step(*args, None, None), "early_train_step")
        m._fwd_token += 1
        m._weights_version = m._flat._version                  # the refresh inside the call keeps the operand copies current
        return loss[0]

    def _itm_draw(self, B):
        """the draws of prepare_itm_inputs (reference :300-330) as plain lists: source row and label per post"""
        src, labels = list(range(B)), [1] * B
        if B > 1:
            for idx in range(B):
                if np.random.choice(2) == 0:
                    labels[idx] = 0
                    j = int(np.random.choice(B - 1))
                    src[idx] = j if j < idx else j + 1
        return src, labels

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
            self.check_indices()
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
        self.check_indices()
        return {"data_id": torch.cat(ids_all).numpy() if ids_all else np.zeros(0, dtype=np.int64), "loss": float(np.mean(losses)),
                "predictions": torch.cat(preds).numpy(), "labels": torch.cat(labels).numpy()}

    def check_indices(self):
        """the engine clamps token ids / token types into their tables and counts them (include/mmhip.h mmhip_early_set_index_counter): the
        reference raises IndexError for such an index, so does this -- at the end of an epoch or an evaluation loop (synchronises)"""
        n, seen = int(self.model._bad_index.item()), getattr(self, "_bad_seen", 0)
        if n > seen:
            self._bad_seen = n
            raise IndexError(f"index out of range in self: {n - seen} token id(s) / token type(s) outside their embedding tables reached the encoder; "
                             "the engine clamped them to a valid row instead of following them")
