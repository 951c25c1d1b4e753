#!/usr/bin/env python3
"""Golden vectors for BASELINE config 4's vision tower (CLIP-ViT-L/14 shape), this container only.

The reference cannot run this configuration as shipped (models/config.py:82-84 and mm_late.py:74-75,81 hard-wire 768-wide image
features), so the pinned part is the third-party arithmetic the path depends on: HuggingFace's `VisionTextDualEncoderModel` with a
`CLIPVisionModel` tower (transformers, as installed) is run on the deterministic weights of `oracle.mm_oracle.make_params`, and its
vision last_hidden_state / pooler_output, text pooler_output and logits_per_text are stored with the inputs.  The fusion head on
top (concat with a widened linear_fusion) is the reference's own formula (models/mm_late.py:92-96) restated in the oracle.
Run:  python tests/golden/make_clip_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import mm_oracle as O  # noqa: E402


def main():
    from transformers import CLIPVisionConfig, XLMRobertaConfig, VisionTextDualEncoderConfig, VisionTextDualEncoderModel
    for name, image in (("clip_small_224", 224), ("clip_small_336", 336)):
        cfg = O.OracleConfig(layers_txt=1, layers_img=2, vocab=400, max_pos=130, num_labels=3, fusion="concat", img_kind="clip",
                             hidden_img=256, heads_img=4, inter_img=512, patch=14, image=image, ln_eps_img=1e-5, p_hidden=0.0, p_attn=0.0, p_head=0.0)
        B, T, seed_w, seed_x = 3, 32, 11, 23
        vc = CLIPVisionConfig(hidden_size=cfg.Hv, intermediate_size=cfg.Iv, num_hidden_layers=cfg.layers_img, num_attention_heads=cfg.heads_v,
                              image_size=image, patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu", attention_dropout=0.0)
        tc = XLMRobertaConfig(vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos, type_vocab_size=1, layer_norm_eps=cfg.ln_eps_txt,
                              num_hidden_layers=cfg.layers_txt, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=1)
        hf = VisionTextDualEncoderModel(VisionTextDualEncoderConfig.from_vision_text_configs(vc, tc, projection_dim=cfg.proj_dim)).eval()
        P = O.make_params(cfg, seed_w)
        sd = {}
        for k, v in P.items():
            if k.startswith("dual_encoder.vision_model.vision_model."):
                sd["vision_model." + k[len("dual_encoder.vision_model.vision_model."):]] = v
            elif k.startswith("dual_encoder."):
                sd[k[len("dual_encoder."):]] = v
        missing, unexpected = hf.load_state_dict(sd, strict=False)
        assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
        ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, seed_x, True)
        with torch.no_grad():
            out = hf(input_ids=ids, attention_mask=mask, pixel_values=pixels, return_dict=True)
        vo, to = out.vision_model_output, out.text_model_output
        np.savez_compressed(os.path.join(HERE, name + ".npz"), cfg=str(O.asdict(cfg)), B=B, T=T, seed_w=seed_w, seed_x=seed_x,
                            ids=ids.numpy(), mask=mask.numpy(),
                            v_cls=vo.last_hidden_state[:, 0].numpy(), v_tok7=vo.last_hidden_state[:, 7].numpy(),
                            v_last=vo.last_hidden_state[:, -1].numpy(), v_pool=vo.pooler_output.numpy(), t_pool=to.pooler_output.numpy(),
                            logits_per_text=out.logits_per_text.numpy())
        print(name, "tokens", vo.last_hidden_state.shape[1], "logits", out.logits_per_text.abs().max().item())


if __name__ == "__main__":
    main()
