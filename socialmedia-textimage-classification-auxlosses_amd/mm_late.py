"""Drop-in for the reference's late-fusion model and trainer (models/mm_late.py), backed by libmmhip.so.

  MM_Model        same constructor / forward signature / state_dict keys as reference models/mm_late.py:50-193;
                  forward + backward run the hand-written HIP kernels through the C ABI (include/mmhip.h).
  MMLate_Model    same role as reference models/mm_late.py:298-739 (ITM sampling :389-414, train :416-532, eval :534-638),
                  with a fused step (loss + backward + AdamW on flat buffers) and one-process-per-GPU data parallelism.

There is no CPU / eager fallback: without the built library (or without a GPU) construction raises.
"""
import ctypes as C
import json
import logging
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import dist as mmdist
from .config import MODEL_DIR_DICT, TEXT_ARCH, IMAGE_ARCH, metric_names
from .utils import agg_metrics_val

logger = logging.getLogger(__name__)


class Scaled_Dot_Product_Attention(nn.Module):
    """reference models/mm_late.py:195-210 (kept for API parity; MM_Model's attention fusion runs inside the engine)."""

    def forward(self, Q, K, V, scale=None):
        attention = torch.matmul(Q, K.permute(0, 2, 1))
        scores = attention
        if scale:
            attention = attention * scale
        attention = torch.softmax(attention, dim=-1)
        return torch.matmul(attention, V), scores


class _Node(nn.Module):
    """name-space node so that parameters carry the reference checkpoint's dotted keys"""


def default_arch(txt_model_name, img_model_name):
    if txt_model_name not in TEXT_ARCH:
        raise ValueError(f"text model {txt_model_name!r}: late-fusion path supports {sorted(TEXT_ARCH)}")
    if img_model_name not in IMAGE_ARCH:
        raise ValueError(f"image model {img_model_name!r}: late-fusion HIP path supports {sorted(IMAGE_ARCH)} (ViT-B/16, CLIP-ViT-L/14)")
    a = dict(hidden=768, heads=12, inter=3072, layers_txt=12, layers_img=12, proj_dim=512, p_hidden=0.1, p_attn=0.1,
             img_kind="vit", hidden_img=0, heads_img=0, inter_img=0)
    a.update(TEXT_ARCH[txt_model_name])
    a.update(IMAGE_ARCH[img_model_name])
    return a


def _read_hf_dir(path):
    """(config dict, state dict) of a local HuggingFace model directory, or (None, None)"""
    cfg_file = os.path.join(path, "config.json")
    if not os.path.isfile(cfg_file):
        return None, None
    with open(cfg_file) as f:
        cfg = json.load(f)
    sd = None
    if os.path.isfile(os.path.join(path, "model.safetensors")):
        from safetensors.torch import load_file
        sd = load_file(os.path.join(path, "model.safetensors"))
    elif os.path.isfile(os.path.join(path, "pytorch_model.bin")):
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    return cfg, sd


def _clip_key_to_ref(k):
    """HF CLIPVisionModel state-dict keys -> the dual encoder's 4.25.1 names (CLIPVisionModel wraps its transformer as
    `.vision_model`: dual_encoder.vision_model.vision_model.*; transformers >= 5 flattens one level)"""
    return k if k.startswith("vision_model.") else "vision_model." + k


def _vit_key_to_ref(k):
    """transformers >= 5 ViT names -> 4.25.1 checkpoint names"""
    if k.startswith("layers."):
        k = "encoder.layer." + k[len("layers."):]
        k = k.replace("attention.q_proj", "attention.attention.query").replace("attention.k_proj", "attention.attention.key")
        k = k.replace("attention.v_proj", "attention.attention.value").replace("attention.o_proj", "attention.output.dense")
        k = k.replace("mlp.fc1", "intermediate.dense").replace("mlp.fc2", "output.dense")
    return k


class _MMFunction(torch.autograd.Function):
    """autograd edge around the engine so that a reference-style caller's loss.backward() (mm_late.py:489) works."""

    @staticmethod
    def forward(ctx, model, ids, mask, pixels, tim_ids, tim_mask, *params):
        outs = model._engine_forward(ids, mask, pixels, tim_ids, tim_mask)
        ctx.model, ctx.token, ctx.has_tim = model, model._fwd_token, tim_ids is not None
        ctx.set_materialize_grads(False)
        out_cls, lpt, out_tim, feats = outs
        return (out_cls, lpt, out_tim, feats) if ctx.has_tim else (out_cls, lpt, feats)

    @staticmethod
    def backward(ctx, *douts):
        model = ctx.model
        if ctx.token != model._fwd_token:
            raise RuntimeError("MM_Model: backward() after another forward(); the engine keeps one set of activations")
        if ctx.has_tim:
            d_cls, d_lpt, d_tim, d_feats = douts
        else:
            (d_cls, d_lpt, d_feats), d_tim = douts, None
        grads = model._engine_backward_autograd(d_cls, d_lpt, d_tim, d_feats)
        return (None,) * 6 + tuple(grads)


class MM_Model(nn.Module):
    """reference models/mm_late.py:50-193.

    MM_Model(num_labels, txt_model_name, img_model_name, dropout, fusion_name='concat'); keyword-only extras are
    additive: `arch` overrides (layer counts, vocab ... for tests), `dtype` ('bf16' | 'f16' | 'bf16x3' = strict-parity mode; with it
    `backward_products` = 3 | 2 | 1 bf16 MFMA products in the backward's matrix products, the forward always three), `max_posts` /
    `max_text_len` (capacity the workspace is sized for), `device`, `seed`.
    Weights: loaded from the local directories of config.MODEL_DIR_DICT when they exist (HF layout), else random
    init at the architecture's true shapes (no network in this environment).
    """

    def __init__(self, num_labels, txt_model_name, img_model_name, dropout, fusion_name="concat", *, arch=None,
                 dtype="bf16", max_posts=64, max_text_len=128, device=None, seed=0, backward_products=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise _lib.MMHipError("MM_Model needs an MI355X (gfx950) GPU: the HIP path has no CPU fallback")
        if fusion_name not in ("concat", "attention"):
            raise NotImplementedError(f"fusion {fusion_name!r}: the HIP late-fusion path implements 'attention' and 'concat' "
                                      "(reference models/mm_late.py:92-113)")
        self.num_labels, self.fusion_name = num_labels, fusion_name
        self.txt_model_name, self.img_model_name = txt_model_name, img_model_name
        self.device_ = torch.device(device if device is not None else f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
        a = default_arch(txt_model_name, img_model_name)
        txt_cfg, txt_sd = _read_hf_dir(MODEL_DIR_DICT.get(txt_model_name, ""))
        img_cfg, img_sd = _read_hf_dir(MODEL_DIR_DICT.get(img_model_name, ""))
        if txt_cfg:
            a.update(vocab=txt_cfg["vocab_size"], max_pos=txt_cfg["max_position_embeddings"], type_vocab=txt_cfg.get("type_vocab_size", 1),
                     layers_txt=txt_cfg["num_hidden_layers"], ln_eps_txt=txt_cfg.get("layer_norm_eps", 1e-12),
                     pad_id=txt_cfg.get("pad_token_id", a["pad_id"]), p_hidden=txt_cfg.get("hidden_dropout_prob", 0.1),
                     p_attn=txt_cfg.get("attention_probs_dropout_prob", 0.1))
        if img_cfg:
            img_cfg = img_cfg.get("vision_config", img_cfg)            # a full CLIP checkpoint directory nests the tower's config
            a.update(layers_img=img_cfg["num_hidden_layers"], image=img_cfg.get("image_size", 224), patch=img_cfg.get("patch_size", 16),
                     ln_eps_img=img_cfg.get("layer_norm_eps", 1e-12))
            if a["img_kind"] == "clip":
                a.update(hidden_img=img_cfg.get("hidden_size", 1024), heads_img=img_cfg.get("num_attention_heads", 16),
                         inter_img=img_cfg.get("intermediate_size", 4096))
        a.update(arch or {})
        if fusion_name == "attention" and a["hidden_img"] not in (0, a["hidden"]):
            raise NotImplementedError("attention fusion needs image tokens as wide as the text tower (fc_K / fc_V are 768 x 768, reference "
                                      "models/mm_late.py:74-75): use fusion_name='concat' with the CLIP-ViT-L/14 tower")
        self.arch = a
        self.dtype_name = dtype
        # parity mode only (include/mmhip.h mmhip_set_backward_products): MFMA products per reduction slice in the backward's matrix products;
        # None = the library's default (3, or MMHIP_X3_BWD).  The forward -- logits and loss -- always takes three.
        if backward_products is not None and (dtype != "bf16x3" or int(backward_products) not in (1, 2, 3)):
            raise ValueError("backward_products is 1, 2 or 3 and belongs to dtype='bf16x3'")
        self.backward_products = None if backward_products is None else int(backward_products)
        self._cfg_kw = dict(hidden=a["hidden"], heads=a["heads"], inter=a["inter"], layers_txt=a["layers_txt"], layers_img=a["layers_img"],
                            vocab=a["vocab"], max_pos=a["max_pos"], type_vocab=a["type_vocab"],
                            txt_kind=_lib.TXT_XLMR if a["txt_kind"] == "xlmr" else _lib.TXT_BERT, pad_id=a["pad_id"],
                            ln_eps_txt=a["ln_eps_txt"], ln_eps_img=a["ln_eps_img"], image=a["image"], patch=a["patch"],
                            proj_dim=a["proj_dim"], num_labels=num_labels,
                            fusion=_lib.FUSION_ATTENTION if fusion_name == "attention" else _lib.FUSION_CONCAT,
                            p_hidden=a["p_hidden"], p_attn=a["p_attn"], p_head=float(dropout),
                            img_kind=_lib.IMG_CLIP if a["img_kind"] == "clip" else _lib.IMG_VIT, hidden_img=int(a["hidden_img"]),
                            heads_img=int(a["heads_img"]), inter_img=int(a["inter_img"]),
                            dtype={"bf16": _lib.BF16, "f16": _lib.F16, "bf16x3": _lib.BF16X3}[dtype])
        self._handle = None
        self._capacity = (0, 0)
        self._fwd_token = 0
        self._seed_base = int(seed) if seed is not None else int(torch.initial_seed())
        self._calls = 0
        self._ws = None
        self._last = {}
        self._create_engine(max_posts, max_text_len, first=True)
        self._init_weights()
        if txt_sd is not None:
            self._load_tower(txt_sd, "dual_encoder.text_model.", lambda k: k)
        if img_sd is not None:
            self._load_tower(img_sd, "dual_encoder.vision_model.", _clip_key_to_ref if a["img_kind"] == "clip" else _vit_key_to_ref)
        self._refresh_weights(3)

    # ------------------------------------------------------------------ engine / buffers
    def _create_engine(self, max_posts, max_text_len, first=False):
        lib = _lib.lib()
        cfg = _lib.Config(max_posts=int(max_posts), max_text_len=int(max_text_len), **self._cfg_kw)
        h = C.c_void_p()
        _lib.check(lib.mmhip_create(C.byref(cfg), C.byref(h)), "create")
        if self._handle is not None:
            lib.mmhip_destroy(self._handle)
        self._handle = h
        self._capacity = (int(max_posts), int(max_text_len))
        dev = self.device_
        if first:
            n_frozen, n_train = lib.mmhip_buffer_numel(h, 0), lib.mmhip_buffer_numel(h, 1)
            self._flat_frozen = torch.zeros(n_frozen, dtype=torch.float32, device=dev)
            self._flat_train = torch.zeros(n_train, dtype=torch.float32, device=dev)
            self._flat_grad = torch.zeros(n_train, dtype=torch.float32, device=dev)
            self._infos = []
            pi = _lib.ParamInfo()
            for i in range(lib.mmhip_param_count(h)):
                _lib.check(lib.mmhip_param_info_at(h, i, C.byref(pi)), "param_info")
                self._infos.append(dict(name=pi.name.decode(), shape=tuple(pi.dims[: pi.ndim]), buffer=pi.buffer, group=pi.group,
                                        offset=int(pi.offset), numel=int(pi.numel)))
            self._register_parameters()
            # word-table row flags shared by the backward pass and the row-lazy AdamW (include/mmhip.h: mmhip_adamw_rows)
            self._word_info = next(i for i in self._train_params if i["name"].endswith("word_embeddings.weight"))
            self._word_row_state = torch.zeros((self._word_info["shape"][0] + 3) // 4 * 4, dtype=torch.uint8, device=dev)
        _lib.check(lib.mmhip_set_row_state(h, _lib.ptr(self._word_row_state)), "set_row_state")
        if first:
            # device words of this handle: [0:2] include/mmhip.h mmhip_set_guard {non-finite counter, void-step flag}; [2] mmhip_set_index_counter
            # (token ids that had to be clamped into the word table: the reference raises IndexError for them)
            self._guard4 = torch.zeros(4, dtype=torch.int32, device=dev)
            self._nonfinite = self._guard4[:2]
            self._bad_index = self._guard4[2:3]
            self._loss_scale = 0.0
        _lib.check(lib.mmhip_set_guard(h, _lib.ptr(self._nonfinite)), "set_guard")
        _lib.check(lib.mmhip_set_index_counter(h, _lib.ptr(self._bad_index)), "set_index_counter")
        if self._loss_scale > 0:
            _lib.check(lib.mmhip_set_loss_scale(h, self._loss_scale), "set_loss_scale")
        if self.backward_products is not None:
            _lib.check(lib.mmhip_set_backward_products(h, self.backward_products), "set_backward_products")
        self._ws = None
        torch.cuda.empty_cache()
        self._ws = torch.empty(lib.mmhip_workspace_bytes(h), dtype=torch.uint8, device=dev)
        if os.environ.get("MMHIP_POISON_WS"):       # debugging aid: no kernel may read workspace it has not written
            self._ws.fill_(int(os.environ["MMHIP_POISON_WS"], 0))
        _lib.check(lib.mmhip_bind(h, _lib.ptr(self._flat_frozen), _lib.ptr(self._flat_train), _lib.ptr(self._flat_grad),
                                  _lib.ptr(self._ws), self._ws.numel()), "bind")
        self._stage_ranges = []
        b, e = C.c_uint64(), C.c_uint64()
        for st in range(lib.mmhip_num_backward_stages(h)):
            _lib.check(lib.mmhip_stage_grad_range(h, st, C.byref(b), C.byref(e)), "stage_grad_range")
            self._stage_ranges.append((int(b.value), int(e.value)))
        self._weights_version = None

    def _register_parameters(self):
        """nn.Parameters are views into the flat fp32 buffers, registered under the reference checkpoint's keys."""
        self._train_params = []
        for inf in self._infos:
            flat = self._flat_frozen if inf["buffer"] == 0 else self._flat_train
            view = flat[inf["offset"]: inf["offset"] + inf["numel"]].view(inf["shape"])
            p = nn.Parameter(view, requires_grad=inf["buffer"] == 1)     # 'vision' parameters frozen, mm_late.py:67-69
            node = self
            parts = inf["name"].split(".")
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            node.register_parameter(parts[-1], p)
            inf["param"] = p
            if inf["buffer"] == 1:
                self._train_params.append(inf)
        # transformers 4.25.1 checkpoints carry this buffer
        emb = self._modules["dual_encoder"]._modules["text_model"]._modules["embeddings"]
        emb.register_buffer("position_ids", torch.arange(self.arch["max_pos"], device=self.device_).unsqueeze(0))

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().mmhip_destroy(self._handle)
        except Exception:
            pass

    def _init_weights(self):
        """HF initializer_range 0.02 for the towers, nn.Linear defaults for the heads (reference: from_pretrained + nn.Linear)."""
        g = torch.Generator(device=self.device_).manual_seed(self._seed_base)
        with torch.no_grad():
            for inf in self._infos:
                n, p = inf["name"], inf["param"]
                if n.startswith("dual_encoder."):
                    if n.endswith("logit_scale"):
                        p.fill_(2.6592)
                    elif "LayerNorm.weight" in n or (("layernorm" in n or "layer_norm" in n or "layrnorm" in n) and n.endswith("weight")):
                        p.fill_(1.0)
                    elif n.endswith(".bias"):
                        p.zero_()
                    else:
                        p.normal_(0.0, 0.02, generator=g)
                else:
                    fan_in = p.shape[1] if p.dim() == 2 else self.arch["hidden"] * (2 if n.startswith("linear_fusion") else 1)
                    bound = 1.0 / math.sqrt(fan_in)
                    p.uniform_(-bound, bound, generator=g)
            pad = self.arch["pad_id"]
            self._modules["dual_encoder"]._modules["text_model"]._modules["embeddings"]._modules["word_embeddings"].weight[pad].zero_()

    def _load_tower(self, sd, prefix, keymap):
        own = {inf["name"]: inf["param"] for inf in self._infos}
        with torch.no_grad():
            for k, v in sd.items():
                for strip in ("vit.", "bert.", "roberta.", ""):
                    if k.startswith(strip):
                        name = prefix + keymap(k[len(strip):])
                        if name in own and tuple(own[name].shape) == tuple(v.shape):
                            own[name].copy_(v.to(own[name].device, torch.float32))
                            break

    def _refresh_weights(self, which):
        _lib.check(_lib.lib().mmhip_refresh_weights(self._handle, which, _lib.stream_ptr()), "refresh_weights")
        self._weights_version = (self._flat_train._version, self._flat_frozen._version)

    def _ensure(self, B, T):
        cap_b, cap_t = self._capacity
        if B > cap_b or T > cap_t:
            self._create_engine(max(B, cap_b), max(T, cap_t))
            self._refresh_weights(3)
        elif self._weights_version != (self._flat_train._version, self._flat_frozen._version):
            # parameters were modified in place (optimizer.step / load_state_dict): re-derive the 16-bit GEMM operands
            self._refresh_weights(3 if self._weights_version is None or self._weights_version[1] != self._flat_frozen._version else 2)

    # ------------------------------------------------------------------ engine calls
    # ---- image-tower output cache (include/mmhip.h: mmhip_vision_export / _import)
    def enable_vision_cache(self, capacity_posts):
        """keep the frozen, dropout-free image tower's outputs per post key (306 KB each, in HBM): a post seen again (every
        epoch after the first) skips the tower and the pixel transfer; results are bit-identical.  Keys are passed to
        `_engine_forward(..., vision_keys=...)` (MMLate_Model.train / eval pass the batch's data_id)."""
        rec = int(_lib.lib().mmhip_vision_record_bytes(self._handle))
        self._vcache = dict(rec=rec, cap=int(capacity_posts), slots={}, hits=0, misses=0,
                            buf=torch.empty(int(capacity_posts) * rec, dtype=torch.uint8, device=self.device_))

    def _vision_slots(self, keys, assign):
        vc = self._vcache
        out = []
        for k in keys:
            k = int(k)
            s = vc["slots"].get(k, -1)
            if s < 0 and assign and len(vc["slots"]) < vc["cap"]:
                s = vc["slots"][k] = len(vc["slots"])
            out.append(s)
        t = torch.tensor(out, dtype=torch.int64).pin_memory()
        return out, t.to(self.device_, non_blocking=True)

    def _engine_forward(self, ids, mask, pixels, tim_ids=None, tim_mask=None, seed=None, vision_keys=None):
        dev = self.device_
        ids = ids.to(dev, torch.int64).contiguous()
        mask = mask.to(dev, torch.int64).contiguous()
        if ids.dim() != 2:
            raise ValueError(f"MM_Model.forward: ids {tuple(ids.shape)}")
        B, T = ids.shape
        vc = getattr(self, "_vcache", None) if vision_keys is not None else None
        if vc is not None and len(vision_keys) != B:
            raise ValueError("vision_keys: one key per post")
        self._ensure(B, T)
        cached = False
        if vc is not None:
            slots, slots_dev = self._vision_slots(vision_keys, assign=False)
            if all(s >= 0 for s in slots):
                _lib.check(_lib.lib().mmhip_vision_import(self._handle, _lib.ptr(slots_dev), _lib.ptr(vc["buf"]), vc["cap"], B, _lib.stream_ptr()),
                           "vision_import")
                cached, pixels = True, None
                vc["hits"] += B
        if not cached:
            pixels = pixels.to(dev, torch.float32).contiguous()
            if pixels.dim() != 4 or pixels.shape[0] != B or tuple(pixels.shape[1:]) != (3, self.arch["image"], self.arch["image"]):
                raise ValueError(f"pixel_values must be [{B},3,{self.arch['image']},{self.arch['image']}], got {tuple(pixels.shape)}")
        if tim_ids is not None:
            tim_ids = tim_ids.to(dev, torch.int64).contiguous()
            tim_mask = tim_mask.to(dev, torch.int64).contiguous()
        if seed is None:
            self._calls += 1
            seed = (self._seed_base * 0x9E3779B97F4A7C15 + self._calls) & 0xFFFFFFFFFFFFFFFF
        out_cls = torch.empty(B, self.num_labels, device=dev)
        lpt = torch.empty(B, B, device=dev)
        out_tim = torch.empty(B, 2, device=dev) if tim_ids is not None else None
        feats = torch.empty(B, self.arch["hidden"], device=dev)
        _lib.check(_lib.lib().mmhip_forward(self._handle, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(pixels), _lib.ptr(tim_ids), _lib.ptr(tim_mask),
                                            B, T, int(self.training), seed, _lib.ptr(out_cls), _lib.ptr(lpt), _lib.ptr(out_tim),
                                            _lib.ptr(feats), _lib.stream_ptr()), "forward")
        self._fwd_token += 1
        self._last = dict(B=B, T=T, itm=tim_ids is not None, seed=seed, ids=ids)
        if vc is not None and not cached:
            slots, slots_dev = self._vision_slots(vision_keys, assign=True)         # posts beyond the capacity stay uncached (-1)
            _lib.check(_lib.lib().mmhip_vision_export(self._handle, _lib.ptr(slots_dev), _lib.ptr(vc["buf"]), vc["cap"], _lib.stream_ptr()),
                       "vision_export")
            vc["misses"] += B
        return out_cls, lpt, out_tim, feats

    def active_groups(self, use_itc, use_itm):
        g = {_lib.G_ALWAYS}
        if self.fusion_name == "attention":
            g.add(_lib.G_FUSION_ATT)
        if use_itc:
            g.add(_lib.G_ITC)
        if use_itm:
            g.add(_lib.G_ITM)
        return g

    def active_ranges(self, use_itc, use_itm):
        """merged [begin, end) element ranges of the trainable flat buffer that receive gradients"""
        groups = self.active_groups(use_itc, use_itm)
        spans = sorted((i["offset"], i["offset"] + ((i["numel"] + 3) & ~3)) for i in self._train_params if i["group"] in groups)
        out = []
        for b, e in spans:
            if out and out[-1][1] == b:
                out[-1][1] = e
            else:
                out.append([b, e])
        return [tuple(x) for x in out]

    def _engine_backward_autograd(self, d_cls, d_lpt, d_tim, d_feats):
        B = self._last["B"]
        dev = self.device_
        self._flat_grad.zero_()
        self._word_row_state.bitwise_and_(0xFE)
        f = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
        d_cls = f(d_cls) if d_cls is not None else torch.zeros(B, self.num_labels, device=dev)
        d_lpt, d_tim, d_feats = f(d_lpt), f(d_tim), f(d_feats)
        _lib.check(_lib.lib().mmhip_backward(self._handle, _lib.ptr(d_cls), _lib.ptr(d_lpt), _lib.ptr(d_tim), _lib.ptr(d_feats),
                                             _lib.stream_ptr()), "backward")
        groups = self.active_groups(d_lpt is not None, d_tim is not None)
        out = []
        for inf in self._train_params:
            if inf["group"] in groups:
                out.append(self._flat_grad[inf["offset"]: inf["offset"] + inf["numel"]].view(inf["shape"]).clone())
            else:
                out.append(None)
        # _flat_grad keeps this call's gradient (tests and callers inspect it); the fused step re-establishes the entry condition
        # of include/mmhip.h's backward contract (zero gradient, no "row has a gradient" flags) when it finds this mark
        self._grad_dirty = True
        return out

    def _clean_grad(self):
        if getattr(self, "_grad_dirty", False):
            self._flat_grad.zero_()
            self._word_row_state.bitwise_and_(0xFE)
            self._grad_dirty = False

    # ------------------------------------------------------------------ reference interface
    def forward(self, ids, mask, pixel_values, tim_inputs=None, iadds_task=False):
        """-> (out_cls, logits_per_text, out_tim | None, out_iadds = None, mm_features)   (mm_late.py:148-193)"""
        if iadds_task:
            raise NotImplementedError("iadds head is deprecated in the reference (models/config.py:65,68)")
        tim_ids, tim_mask = tim_inputs if tim_inputs is not None else (None, None)
        if torch.is_grad_enabled() and any(i["param"].requires_grad for i in self._train_params):
            params = [i["param"] for i in self._train_params]
            outs = _MMFunction.apply(self, ids, mask, pixel_values, tim_ids, tim_mask, *params)
            if tim_ids is not None:
                out_cls, lpt, out_tim, feats = outs
            else:
                (out_cls, lpt, feats), out_tim = outs, None
        else:
            out_cls, lpt, out_tim, feats = self._engine_forward(ids, mask, pixel_values, tim_ids, tim_mask)
        return out_cls, lpt, out_tim, None, feats


# =====================================================================================================================
class MMLate_Model(object):
    """reference models/mm_late.py:298-739 (late-fusion branch).  `train()`/`eval()` keep the reference's semantics
    (loss mixing :473-487, ITM sampling :389-414 with the same numpy RNG call order, metrics CSVs every even epoch);
    the step itself is fused: engine loss + backward + AdamW over flat buffers, gradients all-reduced per backward stage
    when torch.distributed is initialised (one process per GPU, RCCL)."""

    def __init__(self, config, txt_model_name, img_model_name, fusion_name, multilabel=False, **model_kw):
        if multilabel:
            raise NotImplementedError("multilabel BCE branch: no task enables it in the reference (models/config.py:10)")
        self.batch_size, self.num_labels = config.batch_size, config.num_labels
        self.multilabel = multilabel
        self.use_clip_loss, self.beta_itc = config.use_clip_loss, config.beta_itc
        self.use_tim_loss, self.beta_itm = config.use_tim_loss, config.beta_itm
        self.txt_model_name, self.img_model_name = txt_model_name, img_model_name
        self.max_length = config.max_length
        model_kw.setdefault("max_posts", config.batch_size)
        model_kw.setdefault("max_text_len", config.max_length)
        self.model = MM_Model(self.num_labels, txt_model_name, img_model_name, config.dropout, fusion_name=fusion_name, **model_kw)
        self.device = self.model.device_
        self._opt = None
        self.world = mmdist.world_size()
        # MMHIP_DP_OPT=shard: reduce-scatter -> sharded AdamW -> all-gather for the dense ranges instead of all-reduce + replicated AdamW (dist.ShardedBuckets)
        self.sharded_optimizer = os.environ.get("MMHIP_DP_OPT", "allreduce") == "shard"
        self.image_processor = None          # GpuImageProcessor when the loaders yield raw images (datasets.py)

    # ---- checkpoints (reference :343-345, :529-531): plain state_dict with the reference's keys
    def load_saved_model(self, model_path):
        self.model.load_state_dict(torch.load(model_path, map_location=self.device))

    def save_model(self, model_path):
        torch.save(self.model.state_dict(), model_path)

    # ---- ITM negative sampling, reference :389-414 (same numpy RNG stream: one choice([True, False]) per row and one
    # choice(list(others)) per swapped row; sources are read from the original ids, so swaps do not chain)
    def prepare_itm_inputs(self, ids, mask):
        B = ids.shape[0]
        src, labels = list(range(B)), [1] * B
        if B > 1:
            # choice(2) / choice(B-1) consume the legacy RandomState stream exactly like choice([True, False]) /
            # choice(list(set(range(B)) - {idx})) (one bounded integer each; index 0 is True; the others-list is
            # ascending), pinned by tests/golden/itm_sampling.npz
            choice = np.random.choice
            for idx in range(B):
                if choice(2) == 0:
                    labels[idx] = 0
                    j = int(choice(B - 1))
                    src[idx] = j if j < idx else j + 1
        # pinned staging + asynchronous copies: a pageable host-to-device copy would wait for the previous step's kernels
        # on this stream and stop the host from running ahead of the GPU
        host = torch.tensor([src, labels], dtype=torch.int64)
        if ids.is_cuda or self.device.type == "cuda":
            host = host.pin_memory()
        dev_pair = host.to(self.device, non_blocking=True)
        lbl = dev_pair[1]
        sel = dev_pair[0] if ids.is_cuda else host[0]
        return (ids.index_select(0, sel).to(self.device, non_blocking=True), mask.index_select(0, sel).to(self.device, non_blocking=True), lbl)

    def loss_weights(self):
        """reference :473-487 -> (w_cls, w_itc, w_itm)"""
        bi = self.beta_itc if self.use_clip_loss else 0.0
        bm = self.beta_itm if self.use_tim_loss else 0.0
        return 1.0 - (bi + bm), bi, bm

    # ---- one fused training step on device tensors; returns (loss[4] device tensor, n_correct device tensor)
    def train_step(self, ids, mask, pixel_values, onehot, class_weight, lr, weight_decay, step, tim=None, vision_keys=None):
        m, lib = self.model, _lib.lib()
        s = _lib.stream_ptr()
        if self.use_tim_loss and tim is None:
            tim = self.prepare_itm_inputs(ids, mask)
        tim_ids, tim_mask, lbl_tim = tim if tim is not None else (None, None, None)
        if not m.training:
            m.train()                              # walks ~370 submodules (1.3 ms of host time): only on a mode change
        m._clean_grad()                            # an autograd-path backward before this step left its gradient in the flat buffer
        self._poll_guard()                                   # the counter as it stood a step ago (pinned copy, no host sync)
        try:
            return self._train_step(ids, mask, pixel_values, onehot, class_weight, lr, weight_decay, step, tim_ids, tim_mask, lbl_tim, vision_keys)
        finally:
            self._post_guard()

    def _train_step(self, ids, mask, pixel_values, onehot, class_weight, lr, weight_decay, step, tim_ids, tim_mask, lbl_tim, vision_keys):
        m, lib = self.model, _lib.lib()
        s = _lib.stream_ptr()
        w_cls, w_itc, w_itm = self.loss_weights()
        onehot = onehot.to(self.device, torch.int64).contiguous()
        cw = None if class_weight is None else class_weight.to(self.device, torch.float32).contiguous()
        loss = torch.empty(4, device=self.device)
        ncorr = torch.empty(1, dtype=torch.int32, device=self.device)
        exchange = (self.world > 1 and not mmdist.SKIP_EXCHANGE) or mmdist.force_exchange()
        if not exchange and self.world == 1 and vision_keys is None and os.environ.get("MMHIP_NATIVE_STEP", "1") != "0":
            # single rank: the whole step is one native call (include/mmhip.h: mmhip_train_step) -- the host enqueues ~250
            # kernels from C++ instead of crossing ctypes ~40 times per step
            return self._native_step(ids, mask, pixel_values, tim_ids, tim_mask, lbl_tim, onehot, cw, lr, weight_decay, step, loss, ncorr)
        if exchange and vision_keys is None and os.environ.get("MMHIP_NATIVE_DP", "1") != "0":
            # data parallel: the same native enqueue (include/mmhip.h: mmhip_train_step_dp); the library calls back between launches so that
            # this process starts / finishes the collectives (dist.py) at the points the staged loop below would
            return self._native_step(ids, mask, pixel_values, tim_ids, tim_mask, lbl_tim, onehot, cw, lr, weight_decay, step, loss, ncorr, exchange=True)
        m._engine_forward(ids, mask, pixel_values, tim_ids, tim_mask, vision_keys=vision_keys)
        _lib.check(lib.mmhip_loss(m._handle, _lib.ptr(onehot), _lib.ptr(cw), _lib.ptr(lbl_tim), w_cls, w_itc, w_itm, _lib.ptr(loss),
                                  _lib.ptr(ncorr), s), "loss")
        _lib.check(lib.mmhip_backward_begin(m._handle, None, None, None, None, s), "backward_begin")
        works = []
        buckets = mmdist.StageBuckets(m._flat_grad) if exchange else None
        n_stage = len(m._stage_ranges)
        for st in range(n_stage):
            _lib.check(lib.mmhip_backward_stage(m._handle, st, s), "backward_stage")
            if exchange and st >= 1:
                # stage st-1's parameter gradients may still be running on the engine's side stream: join them into this
                # stream, then hand the range to RCCL (which orders itself after this stream) while stage st+1 computes;
                # ranges are merged into >= 48 MB buckets (dist.StageBuckets)
                _lib.check(lib.mmhip_backward_join_stage(m._handle, st - 1, s), "backward_join_stage")
                works += mmdist.exchange_stage(m, st - 1, n_stage, self.use_clip_loss, self.use_tim_loss, buckets=buckets)
        _lib.check(lib.mmhip_backward_finish(m._handle, s), "backward_finish")
        finishers = []
        if exchange:
            works += mmdist.exchange_stage(m, n_stage - 1, n_stage, self.use_clip_loss, self.use_tim_loss, finishers, buckets=buckets)
            works += buckets.works
        for w in works:
            w.wait()
        self._share_guard_flag(exchange)
        self._adamw(lr, weight_decay, step, rows=False)        # dense ranges first: the word-table rows are still travelling
        m._refresh_weights(2)                                  # 16-bit GEMM operand copies (no word-table dependence)
        for f in finishers:
            f()
        self._adamw(lr, weight_decay, step, dense=False)
        return loss, ncorr

    def warm_start(self, B=None, T=None):
        """one throw-away training step (learning rate 0, synthetic posts) BEFORE the DataLoader forks its workers: streams, events, the
        workspace and the runtime's launch machinery then exist in this process only.  Measured (tools/loader_bench.py, 8 workers on the
        16-core share of a GPU box): with the first step taken after the fork the process enqueues a step in 7.6 ms instead of 1.3 ms and
        the loader -> train_step path runs at 2 750 instead of 4 080 posts/s.  Parameters, moments, row flags and the dropout call counter
        are left as they were."""
        from .synthetic import synthetic_batch
        m = self.model
        a = m.arch
        B = B or self.batch_size
        T = T or self.max_length
        calls, rs, trs = m._calls, np.random.get_state(), torch.get_rng_state()      # (synthetic_batch draws from torch's global generator)
        ids, mask, px, oh = synthetic_batch(a["vocab"], self.num_labels, int(B), int(T), 0, a["txt_kind"], a["pad_id"], False, a["image"], self.device)
        before = m._flat_train.clone()
        self.train_step(ids, mask, px, oh, None, 0.0, 0.0, 1)
        torch.cuda.synchronize(self.device)
        np.random.set_state(rs)                      # the ITM sampling drew from numpy's global stream
        torch.set_rng_state(trs)                     # a run's shuffle order must not depend on whether it warm-started (num_workers > 0 does)
        m._calls = calls
        m._nonfinite.zero_()                         # the throw-away step leaves no trace in the overflow guard either
        self._nf_seen, self._nf_clean, self._nf_pending = 0, 0, False
        self._opt = None                             # the moments of the throw-away step are dropped (the next step starts from zeros)
        self._opt_rows = None
        m._word_row_state.zero_()
        if not torch.equal(before, m._flat_train):   # lr = 0: AdamW must have left every parameter untouched
            m._flat_train.copy_(before)
            m._refresh_weights(2)

    def _moments(self):
        m = self.model
        if self._opt is None:
            self._opt = (torch.zeros_like(m._flat_train), torch.zeros_like(m._flat_train))
            m._word_row_state.bitwise_and_(1)                      # fresh moments: no row has any yet
        return self._opt

    def _native_step(self, ids, mask, pixels, tim_ids, tim_mask, lbl_tim, onehot, cw, lr, weight_decay, step, loss, ncorr, exchange=False):
        m = self.model
        dev = self.device
        ids = ids.to(dev, torch.int64).contiguous()
        mask = mask.to(dev, torch.int64).contiguous()
        pixels = pixels.to(dev, torch.float32).contiguous()
        B, T = ids.shape
        if pixels.dim() != 4 or pixels.shape[0] != B or tuple(pixels.shape[1:]) != (3, m.arch["image"], m.arch["image"]):
            raise ValueError(f"pixel_values must be [{B},3,{m.arch['image']},{m.arch['image']}], got {tuple(pixels.shape)}")
        m._ensure(B, T)
        sharded = exchange and self.sharded_optimizer
        em, ev = self._row_moments() if sharded else self._moments()
        m._calls += 1
        seed = (m._seed_base * 0x9E3779B97F4A7C15 + m._calls) & 0xFFFFFFFFFFFFFFFF
        w_cls, w_itc, w_itm = self.loss_weights()
        if tim_ids is not None:
            tim_ids = tim_ids.to(dev, torch.int64).contiguous()
            tim_mask = tim_mask.to(dev, torch.int64).contiguous()
        m._last = dict(B=B, T=T, itm=tim_ids is not None, seed=seed, ids=ids)
        args = (m._handle, _lib.ptr(ids), _lib.ptr(mask), _lib.ptr(pixels), _lib.ptr(tim_ids), _lib.ptr(tim_mask), _lib.ptr(lbl_tim), _lib.ptr(onehot),
                _lib.ptr(cw), B, T, seed, int(bool(self.use_clip_loss)), int(bool(self.use_tim_loss)), w_cls, w_itc, w_itm, em if sharded else _lib.ptr(em),
                ev if sharded else _lib.ptr(ev),
                lr, 0.9, 0.999, 1e-8, weight_decay, step, 1.0 / self.world, _lib.ptr(loss), _lib.ptr(ncorr), _lib.stream_ptr())
        if not exchange:
            _lib.check(_lib.lib().mmhip_train_step(*args), "train_step")
        else:
            works, finishers, failure = [], [], []
            buckets = mmdist.ShardedBuckets(m._flat_grad) if sharded else mmdist.StageBuckets(m._flat_grad)
            n_stage = len(m._stage_ranges)

            # per-bucket optimizer (include/mmhip.h MMHIP_CB_BUCKET): a bucket that has just left as a collective is answered with CB_BUCKET; the
            # engine comes back with CB_WAIT_BUCKET, where ITS side stream is made to wait for that collective (RCCL: a stream-side wait) -- and, with
            # the sharded optimizer, where the rank's shard of the bucket is updated and gathered on that stream -- so the bucket's layers are
            # stepped and refreshed beside the backward stages below instead of behind one barrier after the last collective
            per_bucket = os.environ.get("MMHIP_DP_BUCKET_OPT", "1") != "0"
            seen = [0, 0]          # buckets.works / buckets.plan entries already answered | already waited for on the side stream

            def started():
                return len(buckets.plan) if sharded else len(buckets.works)

            def on_stage(_user, st):
                try:
                    if st >= 0:
                        works.extend(mmdist.exchange_stage(m, st, n_stage, self.use_clip_loss, self.use_tim_loss, finishers if st == n_stage - 1 else None,
                                                           buckets=buckets))
                        if per_bucket and st < n_stage - 1 and started() > seen[0]:
                            seen[0] = started()
                            return _lib.CB_BUCKET
                    elif st == _lib.CB_WAIT_BUCKET:
                        with torch.cuda.stream(self._engine_side_stream()):
                            if sharded:
                                entries = buckets.plan[seen[1]: seen[0]]
                                seen[1] = seen[0]
                                self._sharded_dense_update(buckets, lr, weight_decay, step, entries)
                                return _lib.CB_HANDLED
                            for w in buckets.works[seen[1]: seen[0]]:
                                w.wait()
                            seen[1] = seen[0]
                    elif st == _lib.CB_WAIT_DENSE:
                        for w in works + buckets.works:
                            w.wait()
                        self._share_guard_flag(True)
                        if sharded:
                            self._sharded_dense_update(buckets, lr, weight_decay, step, buckets.plan[seen[1]:])
                            return _lib.CB_HANDLED
                    elif st == _lib.CB_FINISH_ROWS:
                        for f in finishers:
                            f()
                    return 0
                except BaseException as exc:              # a Python exception must not unwind through the C frames
                    failure.append(exc)
                    return -2
            cb = _lib.EXCHANGE_CB(on_stage)
            rc = _lib.lib().mmhip_train_step_dp(*args, cb, None)
            if failure:
                raise failure[0]
            _lib.check(rc, "train_step_dp")
        m._fwd_token += 1
        m._flat_train._version                                        # (read only; the refresh inside the call keeps the 16-bit copies current)
        m._weights_version = (m._flat_train._version, m._flat_frozen._version)
        return loss, ncorr

    # ---- sharded optimizer of the dense ranges under data parallelism (MMHIP_DP_OPT=shard; dist.ShardedBuckets): reduce-scatter -> AdamW on the
    # rank's own shard (moments only it keeps) -> all-gather of the updated parameters.  The word table keeps its row-sparse exchange and the
    # replicated row-lazy update.
    def _row_moments(self):
        """moments of the word table only, handed to the engine as base pointers such that (base + word offset) is the buffer: with the dense
        optimizer run by the caller the engine dereferences adam_m / adam_v over the word table's range only (include/mmhip.h MMHIP_CB_HANDLED)"""
        m = self.model
        if getattr(self, "_opt_rows", None) is None:
            V, H = m._word_info["shape"]
            self._opt_rows = (torch.zeros(V * H, dtype=torch.float32, device=self.device), torch.zeros(V * H, dtype=torch.float32, device=self.device))
            self._opt_shards = mmdist.ShardMoments(self.device)
            m._word_row_state.bitwise_and_(1)
        w0 = m._word_info["offset"]
        return tuple(C.c_void_p(t.data_ptr() - w0 * 4) for t in self._opt_rows)

    def _engine_side_stream(self):
        """the engine's side stream of the backward as a torch stream (include/mmhip.h mmhip_side_stream)"""
        if getattr(self, "_side_ext", None) is None:
            ptr = _lib.lib().mmhip_side_stream(self.model._handle)
            if not ptr:
                raise _lib.MMHipError("the engine has no side stream (MMHIP_OVERLAP=0) yet asked for a per-bucket wait")
            self._side_ext = torch.cuda.ExternalStream(int(ptr), device=self.device)
        return self._side_ext

    def _sharded_dense_update(self, buckets, lr, weight_decay, step, entries=None):
        """`entries`: the slice of buckets.plan to update (per-bucket form); None = the whole plan.  Runs on torch's current stream."""
        m, lib = self.model, _lib.lib()
        at = lambda t, el: C.c_void_p(t.data_ptr() + el * 4)
        w0 = m._word_info["offset"]
        active = [(b, min(e, w0)) for b, e in m.active_ranges(self.use_clip_loss, self.use_tim_loss) if b < w0]
        plan = buckets.plan if entries is None else entries
        for ob, oe, _replicated in buckets.own_ranges(plan):
            mo, vo = self._opt_shards.get(ob, oe)
            for ab, ae in active:                       # AdamW touches only parameters that received a gradient (torch skips `grad is None`)
                b, e = max(ob, ab), min(oe, ae)
                if e > b:
                    _lib.check(lib.mmhip_adamw_guarded(at(m._flat_train, b), at(m._flat_grad, b), at(mo, b - ob), at(vo, b - ob), e - b, lr, 0.9, 0.999, 1e-8,
                                                       weight_decay, step, 1.0 / self.world, 1, _lib.stream_ptr(), _lib.ptr(m._nonfinite)), "adamw shard")
        # the shards of the other ranks hold this rank's unsummed gradient: clear the buckets whole (the entry condition of the next backward)
        for b, s_, e, _, _ in plan:
            m._flat_grad[b:e].zero_()
        buckets.gather_params(m._flat_train, plan)

    def _adamw(self, lr, weight_decay, step, dense=True, rows=True):
        m, lib = self.model, _lib.lib()
        em, ev = self._moments()
        at = lambda t, el: C.c_void_p(t.data_ptr() + el * 4)
        V, H = m._word_info["shape"]
        w0 = m._word_info["offset"]                                 # the word table closes the trainable buffer
        for b, e in m.active_ranges(self.use_clip_loss, self.use_tim_loss):
            dense_end = min(e, w0)
            if dense and dense_end > b:
                _lib.check(lib.mmhip_adamw_guarded(at(m._flat_train, b), at(m._flat_grad, b), at(em, b), at(ev, b), dense_end - b, lr, 0.9, 0.999,
                                                   1e-8, weight_decay, step, 1.0 / self.world, 1, _lib.stream_ptr(), _lib.ptr(m._nonfinite)), "adamw")
            if rows and e > w0:
                # rows without gradient and without moments only decay: same values as the dense update, 1/4 of its traffic
                _lib.check(lib.mmhip_adamw_rows_guarded(at(m._flat_train, w0), at(m._flat_grad, w0), at(em, w0), at(ev, w0), V, H,
                                                        _lib.ptr(m._word_row_state), lr, 0.9, 0.999, 1e-8, weight_decay, step,
                                                        1.0 / self.world, 1, _lib.stream_ptr(), _lib.ptr(m._nonfinite)), "adamw_rows")

    # ---- overflow guard, host side.  The device skips a void step by itself (include/mmhip.h: mmhip_set_step_guard); the host only adapts
    # the f16 loss scale, from a pinned copy of the counter taken at the end of the previous step -- no synchronisation in the step loop.
    # clean steps before the f16 loss scale doubles again.  torch.cuda.amp.GradScaler's default is 2000; the runs of this path are two epochs of a few
    # hundred steps (run_mm_late.py), hence 200 -- MMHIP_SCALE_GROWTH_INTERVAL (or the attribute) restores GradScaler's schedule.  Unlike GradScaler the
    # optimizer's step counter (bias correction) also advances on a void step: the reference's loop counts steps, not successful updates.
    GROWTH_INTERVAL = int(os.environ.get("MMHIP_SCALE_GROWTH_INTERVAL", "200"))
    MAX_LOSS_SCALE = 65536.0

    def _share_guard_flag(self, exchange):
        """data parallel: a step is void on every rank if it is on one (the replicas must take the same decision)"""
        if exchange and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(self.model._nonfinite[1:], op=torch.distributed.ReduceOp.MAX)

    def _post_guard(self):
        if getattr(self, "_nf_host", None) is None:
            self._nf_host = torch.zeros(4, dtype=torch.int32).pin_memory()
            self._nf_event = torch.cuda.Event()
            self._nf_seen, self._nf_clean, self._nf_pending = 0, 0, False
        if self.world > 1 and self.sharded_optimizer and not mmdist.SKIP_EXCHANGE:
            mmdist.sync_guard_counter(self.model._nonfinite)          # owner-only AdamW launches: the counter must not stay rank-local
        self._nf_host.copy_(self.model._guard4, non_blocking=True)
        self._nf_event.record()
        self._nf_pending = True

    def _poll_guard(self):
        if not getattr(self, "_nf_pending", False) or not self._nf_event.query():
            return
        self._nf_pending = False
        self._raise_on_clamped_indices(int(self._nf_host[2]))
        n = int(self._nf_host[0]) - self._nf_seen
        m = self.model
        if n <= 0:
            self._nf_clean += 1
            if m.dtype_name == "f16" and self._nf_clean >= self.GROWTH_INTERVAL and 0 < m._loss_scale < self.MAX_LOSS_SCALE:
                self._nf_clean = 0
                m._loss_scale = min(m._loss_scale * 2.0, self.MAX_LOSS_SCALE)
                _lib.check(_lib.lib().mmhip_set_loss_scale(m._handle, m._loss_scale), "set_loss_scale")
            return
        self._nf_seen += n
        self._nf_clean = 0
        self._react_to_overflow(n)

    def _react_to_overflow(self, n):
        m = self.model
        if m.dtype_name != "f16":
            raise FloatingPointError(f"non-finite gradients met {n} times ({m.dtype_name}): the run has diverged")
        cur = m._loss_scale if m._loss_scale > 0 else 1024.0
        m._loss_scale = max(cur / 2.0, 1.0)
        _lib.check(_lib.lib().mmhip_set_loss_scale(m._handle, m._loss_scale), "set_loss_scale")
        logger.warning("f16 gradient overflow (%d sightings; the steps were skipped on the device): loss scale %g -> %g", n, cur, m._loss_scale)

    def _raise_on_clamped_indices(self, count):
        """token ids outside [0, vocab) were clamped by the engine (include/mmhip.h mmhip_set_index_counter): the reference's nn.Embedding raises
        IndexError for them -- so does this, one step late (the count comes from the pinned copy of the guard words) or at the end of a loop"""
        seen = getattr(self, "_bad_seen", 0)
        if count > seen:
            self._bad_seen = count
            raise IndexError(f"index out of range in self: {count - seen} token id(s) outside [0, {self.model.arch['vocab']}) reached the text tower "
                             "(a tokenizer that does not match the checkpoint?); the engine clamped them to a valid row instead of following them")

    def check_indices(self):
        """synchronising form of the check above (end of an evaluation / feature loop)"""
        self._raise_on_clamped_indices(int(self.model._bad_index.item()))

    def check_overflow(self):
        """non-finite gradient elements since the last call (the AdamW kernels skipped and counted them, include/mmhip.h).
        f16: halve the loss scale (dynamic loss scaling) and go on; other dtypes: a real divergence -> FloatingPointError."""
        m = self.model
        n = int(m._nonfinite[0].item()) - getattr(self, "_nf_seen", 0)
        if n <= 0:
            return 0
        m._nonfinite.zero_()
        self._nf_seen, self._nf_clean, self._nf_pending = 0, 0, False
        self._react_to_overflow(n)
        return n

    @staticmethod
    def _unpack(batch):
        """reference :438-447: [B,1,T] -> [B,T], [B,1,3,H,W] -> [B,3,H,W] (with the B == 1 guard)"""
        ids, mask, px = batch["input_ids"], batch["attention_mask"], batch["pixel_values"]
        if ids.dim() == 3:
            ids, mask = ids.squeeze(1), mask.squeeze(1)
        if px.dim() == 5:
            px = px.squeeze(1)
        return ids, mask, px

    def _device_batches(self, dataloader):
        """batches staged to the GPU ahead of use (pinned, side stream); raw-image batches become pixel_values there"""
        if self.device.type != "cuda":
            return dataloader
        from .image_processing import DevicePrefetcher
        return DevicePrefetcher(dataloader, self.device, self.image_processor, ring=getattr(self, "image_ring", None))

    def train(self, dataloader, val_dataloader, epochs, loss_fn=None, lr=1e-5, weight_decay=0.00025, tim_loss_fn=None,
              iadds_loss_fn=None, te_dataloader=None, model_path=None, val_filename=None, te_filename=None, class_weight=None,
              log_every=50):
        """reference :416-532.  `loss_fn` is accepted for signature parity; the class weights it would carry are passed
        as `class_weight` (nn.CrossEntropyLoss(weight=w), run_mm_late.py:85)."""
        import pandas as pd
        if class_weight is None and loss_fn is not None and getattr(loss_fn, "weight", None) is not None:
            class_weight = loss_fn.weight
        res_val, res_te, step = [], [], 0
        # MMHIP_EPOCH_PREFETCH=1 (datasets.loaders_from_data_key hangs a twin loader on the training loader): epochs alternate between the two
        # loaders and the next epoch's loader is primed -- iterator created, its workers decoding -- once this epoch's workers have been handed
        # their last batches, so the epoch boundary costs no wait for a first batch.  Without the twin: the reference's one iteration per epoch.
        loaders = [dataloader] + ([dataloader.mmhip_twin] if getattr(dataloader, "mmhip_twin", None) is not None else [])
        feeds = [self._device_batches(l) for l in loaders]

        def new_epoch(i, epoch):
            if hasattr(getattr(loaders[i], "sampler", None), "set_epoch"):
                loaders[i].sampler.set_epoch(epoch)              # DistributedSampler: a new shuffle per epoch, same on all ranks
        for epoch in range(epochs):
            if mmdist.rank() == 0:
                print("Epoch:", epoch + 1)
            cur = epoch % len(loaders)
            if getattr(feeds[cur], "_primed", None) is None:
                new_epoch(cur, epoch)
            n_batches = len(loaders[cur])
            lead = int(getattr(loaders[cur], "num_workers", 0) or 0) * int(getattr(loaders[cur], "prefetch_factor", 0) or 0)
            for it, batch in enumerate(feeds[cur]):
                if len(loaders) > 1 and epoch + 1 < epochs and it == max(0, n_batches - 1 - lead) and hasattr(feeds[1 - cur], "prime"):
                    new_epoch(1 - cur, epoch + 1)
                    feeds[1 - cur].prime()
                ids, mask, px = self._unpack(batch)
                step += 1
                keys = batch["data_id"].tolist() if (getattr(self.model, "_vcache", None) is not None and "data_id" in batch) else None
                loss, ncorr = self.train_step(ids.to(self.device), mask.to(self.device), px, batch["labels"], class_weight, lr, weight_decay, step,
                                              vision_keys=keys)
                if log_every and it % log_every == 0:
                    self.check_overflow()
                if log_every and it % log_every == 0 and mmdist.rank() == 0:     # the reference syncs every step (:496-498)
                    n = ids.shape[0]
                    print(f"Got {int(ncorr.item())} / {n} with accuracy {float(ncorr.item()) / n * 100:.2f} loss {loss[0].item():.4f}")
            for loader, store, fname, tag in ((val_dataloader, res_val, val_filename, "val"), (te_dataloader, res_te, te_filename, "test")):
                if loader is None:
                    continue
                r = self.eval(loader, class_weight=class_weight)
                r["epoch"] = epoch
                store.append(r)
                if fname is not None and (epoch % 2 == 0 or epoch == epochs - 1) and mmdist.rank() == 0:
                    pd.DataFrame(agg_metrics_val(store, metric_names, self.num_labels)).to_csv(fname, index=False)
                    logger.info("%s saved!", fname)
        if model_path is not None and mmdist.rank() == 0:
            self.save_model(model_path)
            logger.info("%s saved", model_path)

    def eval(self, dataloader, loss_fn=None, tim_loss_fn=None, iadds_loss_fn=None, class_weight=None):
        """reference :534-638: forward without dropout, same loss mix (ITM inputs re-sampled, :565-568), argmax."""
        m, lib = self.model, _lib.lib()
        if class_weight is None and loss_fn is not None and getattr(loss_fn, "weight", None) is not None:
            class_weight = loss_fn.weight
        m.eval()
        ids_all, preds, labels, losses = [], [], [], []
        w_cls, w_itc, w_itm = self.loss_weights()
        cw = None if class_weight is None else class_weight.to(self.device, torch.float32).contiguous()
        # data parallel: each rank evaluates its interleaved share of the data set and the predictions are gathered (the
        # reference evaluates everything in its one process; per-post predictions do not depend on the batch composition,
        # the reported loss is the mean of the per-batch losses of all ranks)
        shard = mmdist.shard_eval_loader(dataloader)
        with torch.no_grad():
            for batch in self._device_batches(shard if shard is not None else dataloader):
                ids, mask, px = self._unpack(batch)
                ids, mask = ids.to(self.device), mask.to(self.device)
                tim = self.prepare_itm_inputs(ids, mask) if self.use_tim_loss else None
                tim_ids, tim_mask, lbl_tim = tim if tim is not None else (None, None, None)
                keys = batch["data_id"].tolist() if (getattr(m, "_vcache", None) is not None and "data_id" in batch) else None
                out_cls, _, _, _ = m._engine_forward(ids, mask, px, tim_ids, tim_mask, vision_keys=keys)
                onehot = batch["labels"].to(self.device, torch.int64).contiguous()
                loss = torch.empty(4, device=self.device)
                _lib.check(lib.mmhip_loss(m._handle, _lib.ptr(onehot), _lib.ptr(cw), _lib.ptr(lbl_tim), w_cls, w_itc, w_itm,
                                          _lib.ptr(loss), None, _lib.stream_ptr()), "loss")
                losses.append(loss[0:1].clone())
                preds.append(out_cls.argmax(dim=1))
                labels.append(onehot.argmax(dim=1))
                if "data_id" in batch:
                    ids_all.append(batch["data_id"])
        batch_losses = torch.cat(losses).cpu().numpy().tolist() if losses else []
        res = {"data_id": torch.cat(ids_all).cpu().numpy() if ids_all else np.zeros(0, dtype=np.int64),
               "loss": float(np.mean(batch_losses)) if batch_losses else float("nan"),
               "predictions": torch.cat(preds).cpu().numpy() if preds else np.zeros(0, dtype=np.int64),
               "labels": torch.cat(labels).cpu().numpy() if labels else np.zeros(0, dtype=np.int64)}
        if shard is not None:
            res["batch_losses"] = batch_losses
            res = mmdist.gather_eval(res)
        self.check_indices()                   # (the .cpu() copies above synchronised already)
        return res
