#!/usr/bin/env python3
"""Create a dummy task tree laid out as the reference expects (cwd = <root>/work/models/run):
    <root>/BERNICE/            tokenizer (tokenizers WordLevel, XLM-R special-token ids) + config.json (random-init shapes)
    <root>/ViT/                config.json
    <root>/work/models/data/   data_key_imgtxt_random.csv (task 2, TIR columns) + text-image/T{id}.jpg
    <root>/work/models/results/mm_late/testing/
so that  `cd <root>/work/models/run && python -m smtc_amd.run_mm_late --task 2 --testing ...`  exercises the real pipeline
(data key -> normalize_tweet -> tokenizer -> PIL decode/resize/normalise -> model).  The shipped reference data keys are
header-only, and no tokenizer / weights exist offline, hence this generator.
With --roi also the early-fusion inputs (reference models/datasets.py:291-294, config.py:148):
    <root>/LXMERT-base/        BERT-style tokenizer ([PAD] 0, [UNK] 100, [CLS] 101, [SEP] 102; token types)
    <root>/work/models/data/<task>_img_feats/{features/feat_<id>, boxes/nbox_<id>}    torch.save'd [1,36,2048] / [1,36,4] per post
    <root>/work/models/results/mm_early/testing/
usage: make_dummy_task.py ROOT [n_rows] [layers] [--roi]"""
import json
import os
import sys

import numpy as np


def main(root, n=240, layers=2):
    from PIL import Image
    from tokenizers import Tokenizer, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    rng = np.random.RandomState(0)
    words = [f"w{i}" for i in range(300)] + ["@USER", "HTTPURL", "sarcasm", "image", "text", "adds", "nothing"]
    texts = []
    for i in range(n):
        k = rng.randint(3, 40)
        t = " ".join(rng.choice(words, k))
        if i % 5 == 0:
            t += " https://t.co/abc" + str(i)
        if i % 7 == 0:
            t = "@someone " + t
        texts.append(t)
    run = os.path.join(root, "work", "models", "run")
    data = os.path.join(root, "work", "models", "data")
    os.makedirs(run, exist_ok=True)
    os.makedirs(os.path.join(data, "text-image"), exist_ok=True)
    os.makedirs(os.path.join(root, "work", "models", "results", "mm_late", "testing"), exist_ok=True)
    # ---- data key, task 2 columns (models/config.py:18-26)
    labels = rng.randint(0, 4, n)
    cols = ["image_adds_text_repr", "image_adds_text_notrepr", "image_notadds_text_repr", "image_notadds_text_notrepr"]
    split = np.array(["train"] * (n * 2 // 3) + ["val"] * (n // 6) + ["test"] * (n - n * 2 // 3 - n // 6))
    with open(os.path.join(data, "data_key_imgtxt_random.csv"), "w") as f:
        f.write("tweet_id,text," + ",".join(cols) + ",split\n")
        for i in range(n):
            onehot = ["1" if labels[i] == c else "0" for c in range(4)]
            f.write(f"{1000 + i},\"{texts[i]}\"," + ",".join(onehot) + f",{split[i]}\n")
            img = Image.fromarray(rng.randint(0, 256, (rng.randint(60, 300), rng.randint(60, 300), 3), dtype=np.uint8))
            img.save(os.path.join(data, "text-image", f"T{1000 + i}.jpg"), quality=80)
    # ---- tokenizer with the XLM-R special-token ids (<s> 0, <pad> 1, </s> 2, <unk> 3)
    tok = Tokenizer(models.WordLevel(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    trainer = trainers.WordLevelTrainer(special_tokens=["<s>", "<pad>", "</s>", "<unk>"])
    from smtc_amd.datasets import normalize_tweet
    tok.train_from_iterator([normalize_tweet(t) for t in texts], trainer)
    tok.post_processor = processors.TemplateProcessing(single="<s> $A </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    bdir, vdir = os.path.join(root, "BERNICE"), os.path.join(root, "ViT")
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(vdir, exist_ok=True)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>",
                                   cls_token="<s>", sep_token="</s>")
    fast.save_pretrained(bdir)
    with open(os.path.join(bdir, "config.json"), "w") as f:
        json.dump({"model_type": "xlm-roberta", "vocab_size": max(64, tok.get_vocab_size()), "max_position_embeddings": 130, "type_vocab_size": 1,
                   "num_hidden_layers": layers, "layer_norm_eps": 1e-5, "pad_token_id": 1, "hidden_dropout_prob": 0.1,
                   "attention_probs_dropout_prob": 0.1, "hidden_size": 768, "num_attention_heads": 12, "intermediate_size": 3072}, f)
    with open(os.path.join(vdir, "config.json"), "w") as f:
        json.dump({"model_type": "vit", "num_hidden_layers": layers, "image_size": 224, "patch_size": 16, "layer_norm_eps": 1e-12,
                   "hidden_size": 768, "num_attention_heads": 12, "intermediate_size": 3072}, f)
    return run


def add_roi(root, n=240, task_name="tir"):
    """ROI-feature files + a BERT-style tokenizer for run_mm_early.py's data-key path; call after main(root, n)"""
    import torch
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    data = os.path.join(root, "work", "models", "data")
    fdir, bdir = os.path.join(data, task_name + "_img_feats", "features"), os.path.join(data, task_name + "_img_feats", "boxes")
    os.makedirs(fdir, exist_ok=True)
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.join(root, "work", "models", "results", "mm_early", "testing"), exist_ok=True)
    g = torch.Generator().manual_seed(7)
    for i in range(n):
        torch.save(torch.rand(1, 36, 2048, generator=g) * 2, os.path.join(fdir, f"feat_{1000 + i}"))
        xy = torch.rand(1, 36, 2, generator=g) * 0.6
        torch.save(torch.cat([xy, xy + 0.05 + torch.rand(1, 36, 2, generator=g) * 0.35], -1), os.path.join(bdir, f"nbox_{1000 + i}"))
    vocab = {"[PAD]": 0}
    for i in range(1, 100):
        vocab[f"[unused{i}]"] = i
    vocab.update({"[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103})
    for wd in [f"w{i}" for i in range(300)] + ["@user", "httpurl", "sarcasm", "image", "text", "adds", "nothing", "@USER", "HTTPURL"]:
        vocab.setdefault(wd, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", pair="[CLS] $A [SEP] $B:1 [SEP]:1", special_tokens=[("[CLS]", 101), ("[SEP]", 102)])
    ldir = os.path.join(root, "LXMERT-base")
    os.makedirs(ldir, exist_ok=True)
    PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]", sep_token="[SEP]", mask_token="[MASK]").save_pretrained(ldir)
    return ldir


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smtc_amd  # noqa: F401
    argv = [x for x in sys.argv[1:] if x != "--roi"]
    print(main(argv[0], int(argv[1]) if len(argv) > 1 else 240, int(argv[2]) if len(argv) > 2 else 2))
    if "--roi" in sys.argv:
        print(add_roi(argv[0], int(argv[1]) if len(argv) > 1 else 240))
