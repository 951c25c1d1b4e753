#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code (this container only).

Imports `/root/reference/models/mm_late.py` untouched behind a shim for the packages this
container lacks (torchvision / torchmetrics stubs, `config.T`, the removed
`ViTFeatureExtractor` alias, local random-init model directories instead of hub names), loads
the deterministic weights of `oracle.mm_oracle.make_params` into the reference `MM_Model`,
runs it, and writes inputs + expected outputs as small `.npz` files next to this script.

Nothing from /root/reference is copied: the fixtures are data (inputs, outputs, seeds).
Run:  python tests/golden/make_golden.py           (needs /root/reference; CPU only)
"""
import os
import sys
import types
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/models"

from oracle import mm_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- shim
def install_shim():
    import importlib.machinery
    import transformers          # before the stubs, so its availability probes see the real environment

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _Any()

    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms", Compose=_Any, Resize=_Any, ToTensor=_Any, Normalize=_Any,
                         CenterCrop=_Any, RandomResizedCrop=_Any, RandomHorizontalFlip=_Any, Lambda=_Any)
    tv.models = stub("torchvision.models", resnet50=_Any, resnet152=_Any)
    tm = stub("torchmetrics")
    tm.classification = stub("torchmetrics.classification", MulticlassF1Score=_Any, MulticlassPrecision=_Any,
                             MulticlassRecall=_Any, MultilabelF1Score=_Any, MultilabelPrecision=_Any,
                             MultilabelRecall=_Any, MulticlassAccuracy=_Any)
    for n in ("F1Score", "Precision", "Recall", "Accuracy"):
        setattr(tm, n, _Any)
        setattr(tm.classification, n, _Any)
    vit_proc = transformers.ViTImageProcessor      # (touching a lazy attribute may swap sys.modules['transformers'])
    if not hasattr(sys.modules["transformers"], "ViTFeatureExtractor"):
        sys.modules["transformers"].ViTFeatureExtractor = vit_proc
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(os.path.dirname(REF), "preprocessing"))
    import config as ref_config
    ref_config.T = [[1, 0], [0, 1]]      # models/utils.py:16 imports a name config.py never defines
    return ref_config


def build_reference_model(ref_config, cfg: O.OracleConfig, txt_name: str, tmp: str, p_txt: float):
    """Random-init HF directories -> reference MM_Model(num_labels, txt, img, dropout, fusion)."""
    from transformers import ViTConfig, ViTModel, XLMRobertaConfig, XLMRobertaModel, BertConfig, BertModel
    vdir, tdir = os.path.join(tmp, "vit"), os.path.join(tmp, txt_name)
    ViTModel(ViTConfig(num_hidden_layers=cfg.layers_img)).save_pretrained(vdir)
    if cfg.txt_kind == "xlmr":
        tc = XLMRobertaConfig(vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos, type_vocab_size=1,
                              layer_norm_eps=cfg.ln_eps_txt, num_hidden_layers=cfg.layers_txt,
                              hidden_dropout_prob=p_txt, attention_probs_dropout_prob=p_txt,
                              pad_token_id=1, bos_token_id=0, eos_token_id=2)
        XLMRobertaModel(tc).save_pretrained(tdir)
    else:
        tc = BertConfig(vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos, type_vocab_size=2,
                        num_hidden_layers=cfg.layers_txt, hidden_dropout_prob=p_txt,
                        attention_probs_dropout_prob=p_txt)
        BertModel(tc).save_pretrained(tdir)
    ref_config.MODEL_DIR_DICT["vit"] = vdir
    ref_config.MODEL_DIR_DICT[txt_name] = tdir
    import mm_late as ref_mm_late
    ref_mm_late.MODEL_DIR_DICT = ref_config.MODEL_DIR_DICT
    model = ref_mm_late.MM_Model(cfg.num_labels, txt_name, "vit", cfg.p_head, cfg.fusion)
    return ref_mm_late, model


def to_hf5_key(k: str) -> str:
    """4.25.1 checkpoint key -> key of the installed transformers (ViT renames only)."""
    if "vision_model.encoder.layer." in k:
        k = k.replace("vision_model.encoder.layer.", "vision_model.layers.")
        k = k.replace("attention.attention.query", "attention.q_proj")
        k = k.replace("attention.attention.key", "attention.k_proj")
        k = k.replace("attention.attention.value", "attention.v_proj")
        k = k.replace("attention.output.dense", "attention.o_proj")
        k = k.replace("intermediate.dense", "mlp.fc1")
        k = k.replace("output.dense", "mlp.fc2")
    return k


def load_params(model, P):
    sd = model.state_dict()
    mapped = {to_hf5_key(k): v for k, v in P.items()}
    missing = [k for k in sd if k not in mapped and not k.endswith("position_ids") and not k.endswith("token_type_ids")]
    extra = [k for k in mapped if k not in sd]
    assert not missing and not extra, (missing[:5], extra[:5])
    for k, v in mapped.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    model.load_state_dict(mapped, strict=False)


def grads_by_ref_key(model, names):
    inv = {to_hf5_key(k): k for k in names}
    out = {}
    for n, p in model.named_parameters():
        if n in inv:
            out[inv[n]] = None if p.grad is None else p.grad.detach().clone()
    return out


# ----------------------------------------------------------------------------- cases
def case_forward(tag, cfg, txt_name, B, T, seed_w, seed_x, pad, ref_config, with_layers):
    with tempfile.TemporaryDirectory() as tmp:
        ref_mm, model = build_reference_model(ref_config, cfg, txt_name, tmp, 0.1)
    P = O.make_params(cfg, seed_w)
    load_params(model, P)
    model.eval()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, seed_x, pad)
    np.random.seed(30)
    tim_ids, tim_mask, lbl_tim = O.prepare_itm_inputs(ids, mask)
    with torch.no_grad():
        out_cls, lpt, out_tim, _, feats = model(ids, mask, pixels, tim_inputs=(tim_ids, tim_mask))
        extra = {}
        if with_layers:
            de = model.dual_encoder
            vo = de.vision_model(pixel_values=pixels, output_hidden_states=True)
            to = de.text_model(input_ids=ids, attention_mask=mask, output_hidden_states=True)
            extra["vit_cls_per_layer"] = torch.stack([h[:, 0] for h in vo.hidden_states[1:]]).numpy()
            extra["txt_cls_per_layer"] = torch.stack([h[:, 0] for h in to.hidden_states[1:]]).numpy()
            extra["vit_last_hidden_post0"] = vo.last_hidden_state[0].numpy()
            extra["txt_last_hidden_post0"] = to.last_hidden_state[0].numpy()
    np.savez_compressed(
        os.path.join(HERE, f"{tag}.npz"),
        cfg=np.array(repr(O.asdict(cfg))), B=B, T=T, seed_w=seed_w, seed_x=seed_x, pad=pad,
        ids=ids.numpy(), mask=mask.numpy(), tim_ids=tim_ids.numpy(), tim_mask=tim_mask.numpy(),
        lbl_tim=lbl_tim.numpy(), onehot=onehot.numpy(),
        out_cls=out_cls.numpy(), logits_per_text=lpt.numpy(), out_tim=out_tim.numpy(), mm_features=feats.numpy(),
        **extra)
    print(tag, "out_cls", out_cls.flatten()[:4].tolist())


def case_train(tag, cfg, txt_name, B, T, seed_w, seed_x, ref_config):
    """train() mode, every dropout p = 0: three loss mixes + gradients of named parameters,
    using the reference's own utils.clip_loss and nn.CrossEntropyLoss as run_mm_late.py builds them."""
    import utils as ref_utils
    cfg0 = O.OracleConfig(**{**O.asdict(cfg), "p_hidden": 0.0, "p_attn": 0.0, "p_head": 0.0})
    with tempfile.TemporaryDirectory() as tmp:
        ref_mm, model = build_reference_model(ref_config, cfg0, txt_name, tmp, 0.0)
    P = O.make_params(cfg0, seed_w)
    load_params(model, P)
    model.train()
    ids, mask, pixels, onehot = O.synthetic_batch(cfg0, B, T, seed_x, True)
    np.random.seed(30)
    tim_ids, tim_mask, lbl_tim = O.prepare_itm_inputs(ids, mask)
    w = torch.tensor([0.7, 1.6, 0.9, 1.1][: cfg0.num_labels])
    loss_fn = torch.nn.CrossEntropyLoss(weight=w)            # run_mm_late.py:85
    tim_loss_fn = torch.nn.CrossEntropyLoss()                # run_mm_late.py:97
    watch = ["linear_cls.weight", "linear_cls.bias", "linear_tim.weight", "linear_fusion.bias", "fc_Q.bias",
             "fc_K.weight", "fc_V.weight", "fc_Q.weight", "dual_encoder.logit_scale",
             "dual_encoder.text_projection.weight", "dual_encoder.visual_projection.weight",
             "dual_encoder.text_model.pooler.dense.weight",
             "dual_encoder.text_model.encoder.layer.0.attention.self.query.weight",
             "dual_encoder.text_model.encoder.layer.0.attention.self.key.bias",
             "dual_encoder.text_model.encoder.layer.0.attention.self.value.weight",
             "dual_encoder.text_model.encoder.layer.0.attention.output.dense.weight",
             "dual_encoder.text_model.encoder.layer.0.attention.output.LayerNorm.weight",
             "dual_encoder.text_model.encoder.layer.0.intermediate.dense.weight",
             "dual_encoder.text_model.encoder.layer.0.intermediate.dense.bias",
             "dual_encoder.text_model.encoder.layer.1.output.dense.weight",
             "dual_encoder.text_model.encoder.layer.1.output.LayerNorm.bias",
             "dual_encoder.text_model.embeddings.LayerNorm.weight",
             "dual_encoder.text_model.embeddings.position_embeddings.weight",
             "dual_encoder.text_model.embeddings.token_type_embeddings.weight",
             "dual_encoder.text_model.embeddings.word_embeddings.weight"]
    out = dict(cfg=np.array(repr(O.asdict(cfg0))), B=B, T=T, seed_w=seed_w, seed_x=seed_x, ids=ids.numpy(),
               mask=mask.numpy(), tim_ids=tim_ids.numpy(), tim_mask=tim_mask.numpy(), lbl_tim=lbl_tim.numpy(),
               onehot=onehot.numpy(), class_weight=w.numpy(), watch=np.array(watch))
    for mix, (itc, itm) in {"plain": (False, False), "itc": (True, False), "itm": (False, True), "itcitm": (True, True)}.items():
        model.zero_grad(set_to_none=True)
        o, lpt, ot, _, _ = model(ids, mask, pixels, tim_inputs=(tim_ids, tim_mask) if itm else None)
        label = onehot.type_as(o)                            # mm_late.py:471
        if itc and itm:
            loss = (1 - 0.2) * loss_fn(o, label) + 0.1 * ref_utils.clip_loss(lpt) + 0.1 * tim_loss_fn(ot, lbl_tim)
        elif itc:
            loss = (1 - 0.1) * loss_fn(o, label) + 0.1 * ref_utils.clip_loss(lpt)
        elif itm:
            loss = (1 - 0.1) * loss_fn(o, label) + 0.1 * tim_loss_fn(ot, lbl_tim)
        else:
            loss = loss_fn(o, label)
        loss.backward()
        out[f"{mix}.loss"] = loss.item()
        out[f"{mix}.out_cls"] = o.detach().numpy()
        G = grads_by_ref_key(model, watch)
        none = sorted(n for n, q in model.named_parameters() if q.requires_grad and q.grad is None)
        out[f"{mix}.grad_none"] = np.array(none)
        for k in watch:
            g = G.get(k)
            if g is None:
                continue
            out[f"{mix}.gnorm.{k}"] = g.norm().item()
            if k.endswith("word_embeddings.weight"):
                touched = torch.unique(ids)[:6]
                out[f"{mix}.gslice.{k}"] = g[touched][:, :48].numpy()
                out[f"{mix}.gslice_rows.{k}"] = touched.numpy()
                out[f"{mix}.gpad_row_norm"] = g[cfg0.pad_id].norm().item()
            elif g.dim() == 2:
                out[f"{mix}.gslice.{k}"] = g[:8, :48].numpy()
            else:
                out[f"{mix}.gslice.{k}"] = g.flatten()[:64].numpy()
        print(tag, mix, "loss", loss.item())
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)


def case_itm(ref_config):
    """prepare_itm_inputs with the reference's own method (models/mm_late.py:389-414)."""
    import mm_late as ref_mm_late
    obj = ref_mm_late.MMLate_Model.__new__(ref_mm_late.MMLate_Model)
    res = {}
    for B in (1, 2, 8, 64):
        g = torch.Generator().manual_seed(B)
        ids = torch.randint(3, 1000, (B, 16), generator=g)
        mask = (torch.rand(B, 16, generator=g) > 0.3).long()
        np.random.seed(30)
        t_ids, t_mask, lbl = obj.prepare_itm_inputs(ids, mask)
        res[f"B{B}.ids"], res[f"B{B}.mask"] = ids.numpy(), mask.numpy()
        res[f"B{B}.tim_ids"], res[f"B{B}.tim_mask"], res[f"B{B}.lbl"] = t_ids.numpy(), t_mask.numpy(), lbl.numpy()
    np.savez_compressed(os.path.join(HERE, "itm_sampling.npz"), **res)
    print("itm B8 labels", res["B8.lbl"].tolist())


def case_adamw(ref_config):
    """three torch.optim.AdamW steps configured as models/utils.py:280-292 + models/mm_late.py:420-422."""
    import utils as ref_utils
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(1024, generator=g) * 0.05
    grads = [torch.randn(1024, generator=g) * (10.0 ** -(i + 1)) for i in range(3)]
    p = torch.nn.Parameter(p0.clone())
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
    groups = ref_utils.get_optimizer_params([("p", p), ("frozen", frozen)], 0.00025, 1e-5)
    opt = torch.optim.AdamW(groups, lr=1e-5)
    traj = []
    for gr in grads:
        p.grad = gr.clone()
        opt.step()
        traj.append(p.detach().clone().numpy())
    np.savez_compressed(os.path.join(HERE, "adamw.npz"), p0=p0.numpy(), grads=torch.stack(grads).numpy(),
                        traj=np.stack(traj), lr=1e-5, wd=0.00025)


def case_losses(ref_config):
    import utils as ref_utils
    g = torch.Generator().manual_seed(3)
    out = torch.randn(4, 3, generator=g)
    onehot = torch.nn.functional.one_hot(torch.tensor([0, 2, 1, 2]), 3)
    w = torch.tensor([0.5, 2.0, 1.25])
    l_cls = torch.nn.CrossEntropyLoss(weight=w)(out, onehot.type_as(out)).item()
    sim = torch.randn(6, 6, generator=g) * 3
    l_clip = ref_utils.clip_loss(sim).item()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), out=out.numpy(), onehot=onehot.numpy(), w=w.numpy(),
                        l_cls=l_cls, sim=sim.numpy(), l_clip=l_clip)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_config = install_shim()
    small = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, num_labels=3)
    small_bert = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=512, type_vocab=2, txt_kind="bert",
                                pad_id=0, ln_eps_txt=1e-12, num_labels=2)
    small_concat = O.OracleConfig(layers_txt=1, layers_img=1, vocab=1000, max_pos=130, num_labels=4, fusion="concat")
    full = O.OracleConfig(vocab=1000, max_pos=130, num_labels=2)
    case_losses(ref_config)
    case_adamw(ref_config)
    case_itm(ref_config)
    case_forward("fwd_small_xlmr", small, "bernice", 4, 64, 0, 11, True, ref_config, True)
    case_forward("fwd_small_bert", small_bert, "bert", 4, 64, 1, 12, True, ref_config, False)
    case_forward("fwd_small_concat", small_concat, "bernice", 3, 32, 2, 13, True, ref_config, False)
    case_forward("fwd_full_xlmr", full, "bernice", 2, 128, 0, 14, True, ref_config, False)
    case_train("train_small_xlmr", small, "bernice", 4, 64, 0, 21, ref_config)
    extra_forward_cases(ref_config)


def extra_forward_cases(ref_config):
    """Round 4 (VERDICT r3 #1c): nine more forward goldens -- other weight / input seeds, batch sizes, sequence lengths, padding on and
    off, BERT and XLM-R, concat and attention fusion, 2 / 6 / 12 layers -- so that a numerics policy is judged on thirteen cases, not four.
    `python tests/golden/make_golden.py --extra` writes only these."""
    full = O.OracleConfig(vocab=1000, max_pos=130, num_labels=2)
    full3 = O.OracleConfig(vocab=1000, max_pos=130, num_labels=3)
    full_bert = O.OracleConfig(vocab=1000, max_pos=512, type_vocab=2, txt_kind="bert", pad_id=0, ln_eps_txt=1e-12, num_labels=2)
    full_concat = O.OracleConfig(vocab=1000, max_pos=130, num_labels=4, fusion="concat")
    mid = O.OracleConfig(layers_txt=6, layers_img=6, vocab=1000, max_pos=130, num_labels=3)
    small = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, num_labels=3)
    small_bert = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=512, type_vocab=2, txt_kind="bert",
                                pad_id=0, ln_eps_txt=1e-12, num_labels=2)
    case_forward("fwd_x_full_xlmr_a", full, "bernice", 2, 128, 3, 31, True, ref_config, False)
    case_forward("fwd_x_full_xlmr_b", full3, "bernice", 3, 96, 4, 32, False, ref_config, False)
    case_forward("fwd_x_full_xlmr_c", full3, "bernice", 4, 128, 11, 39, True, ref_config, False)
    case_forward("fwd_x_full_bert_a", full_bert, "bert", 2, 128, 5, 33, True, ref_config, False)
    case_forward("fwd_x_full_bert_b", full_bert, "bert", 4, 64, 6, 34, False, ref_config, False)
    case_forward("fwd_x_full_concat", full_concat, "bernice", 3, 64, 7, 35, True, ref_config, False)
    case_forward("fwd_x_mid_xlmr", mid, "bernice", 4, 64, 10, 38, True, ref_config, False)
    case_forward("fwd_x_small_xlmr", small, "bernice", 8, 32, 8, 36, False, ref_config, False)
    case_forward("fwd_x_small_bert", small_bert, "bert", 5, 48, 9, 37, True, ref_config, False)


if __name__ == "__main__":
    if "--extra" in sys.argv:
        torch.manual_seed(0)
        torch.set_num_threads(8)
        extra_forward_cases(install_shim())
    else:
        main()
