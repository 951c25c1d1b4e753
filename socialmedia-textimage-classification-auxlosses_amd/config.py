"""Task / path / model-directory configuration -- mirrors reference models/config.py (same names, same defaults).

Additive to the reference: `batch_size` override (`--batch_size`, BASELINE config 2 needs 64), architecture presets
for random-init runs when the local HuggingFace directories of MODEL_DIR_DICT do not exist (no network here)."""
import os

# constants, reference models/config.py:82-85
txt_feat_size = 768
fixed_feat_size = 768
img_feat_size = 768
img_feat_size_cnn = 2048

TASKS = {0: "text_is_represented", 1: "image_adds", 2: "tir", 3: "mvsa", 4: "mhp", 5: "mic", 6: "msd"}   # :87-95
DATA_PATH = "../data/"
PATH = {0: DATA_PATH + "data_key_imgtxt_random.csv", 1: DATA_PATH + "data_key_imgtxt_random.csv",
        2: DATA_PATH + "data_key_imgtxt_random.csv", 3: DATA_PATH + "data_key_mvsa.csv", 4: DATA_PATH + "data_key_mhp.csv",
        5: DATA_PATH + "data_key_mic.csv", 6: DATA_PATH + "data_key_msd.csv"}
IMG_FMT = {0: DATA_PATH + "text-image/T{}.jpg", 1: DATA_PATH + "text-image/T{}.jpg", 2: DATA_PATH + "text-image/T{}.jpg",
           3: DATA_PATH + "MVSA-Single/data/{}.jpg", 4: DATA_PATH + "MHP/Data/Images/{}.jpg",
           5: DATA_PATH + "MIC/spc_imgs_twitter/{}_1.jpg", 6: DATA_PATH + "MSD/dataset_image/{}.jpg"}
CLASSES = {2: ["image adds and text is represented", "image adds and text is not represented",
               "image does not add and text is represented", "image does not adds and text is not represented"],
           3: ["neutral", "positive", "negative"], 6: ["not sarcastic", "sarcastic"]}
metric_names = ["f1_weighted", "f1_macro", "precision_weighted", "precision_macro", "recall_weighted", "recall_macro", "loss"]
RES_PATH = "../results/"
results_dir_mm_late = RES_PATH + "mm_late/"
MODEL_DIR_DICT = {"bert": "../../../BERT-base/", "bertweet": "../../../BERTWEET-base/", "roberta": "../../../RoBERTa-base/",
                  "bernice": "../../../BERNICE/", "vit": "../../../ViT/", "beit": "../../../BEiT/", "deit": "../../../DEiT/",
                  "clip": "../../../CLIP-ViT-L-14/", "clip336": "../../../CLIP-ViT-L-14-336/",
                  "lxmert": "../../../LXMERT-base/"}      # the reference names the hub id "unc-nlp/lxmert-base-uncased" (config.py:148): no network here

# task -> (num_labels, batch_size), reference models/config.py:13-48
_TASK_SHAPE = {0: (2, 8), 1: (2, 8), 2: (4, 8), 3: (3, 16), 4: (4, 8), 5: (2, 16), 6: (2, 16)}

# architecture presets used when the model directory is absent (random init at the true shapes)
TEXT_ARCH = {
    "bernice": dict(txt_kind="xlmr", vocab=250002, max_pos=130, type_vocab=1, pad_id=1, ln_eps_txt=1e-5),
    "roberta": dict(txt_kind="xlmr", vocab=50265, max_pos=514, type_vocab=1, pad_id=1, ln_eps_txt=1e-5),
    "bertweet": dict(txt_kind="xlmr", vocab=64001, max_pos=130, type_vocab=1, pad_id=1, ln_eps_txt=1e-5),
    "bert": dict(txt_kind="bert", vocab=30522, max_pos=512, type_vocab=2, pad_id=0, ln_eps_txt=1e-12),
}
# "clip" / "clip336" (additive; BASELINE config 4): CLIP-ViT-L/14 vision tower, HF CLIPVisionModel -- 1024 wide, 16 heads, 4096 MLP,
# 24 pre-LN layers, quick-GELU, 14 x 14 patches (257 tokens at 224, 577 at 336: openai/clip-vit-large-patch14[-336])
IMAGE_ARCH = {"vit": dict(image=224, patch=16, ln_eps_img=1e-12, img_kind="vit"),
              "clip": dict(image=224, patch=14, ln_eps_img=1e-5, img_kind="clip", hidden_img=1024, heads_img=16, inter_img=4096, layers_img=24),
              "clip336": dict(image=336, patch=14, ln_eps_img=1e-5, img_kind="clip", hidden_img=1024, heads_img=16, inter_img=4096, layers_img=24)}


class Config(object):
    """reference models/config.py:1-77.  `data` is filled from the data key when it exists (pandas), else left None
    (synthetic runs)."""

    def __init__(self, args, model_name=None, multimodal=True, txt=False):
        self.multilabel = args.task in {10}
        self.num_labels, self.batch_size = _TASK_SHAPE[args.task]
        if getattr(args, "batch_size", None):
            self.batch_size = args.batch_size
        self.data = None
        path = PATH[args.task]
        if os.path.exists(path) and not getattr(args, "synthetic", False):
            import numpy as np
            import pandas as pd
            key = pd.read_csv(path)
            if args.task < 2:
                self.data = key[["tweet_id", "text", TASKS[args.task], "split"]].rename(columns={TASKS[args.task]: "label"})
            elif args.task == 2:
                data = key[["tweet_id", "text", "split"]].copy()
                cols = ["image_adds_text_repr", "image_adds_text_notrepr", "image_notadds_text_repr", "image_notadds_text_notrepr"]
                data["label"] = np.argmax(key[cols].to_numpy(), axis=1)
                self.data = data[["tweet_id", "text", "label", "split"]]
            elif args.task == 5:
                self.data = key[["id", "text", "label", "split"]].rename(columns={"id": "tweet_id"})
            else:
                self.data = key[["tweet_id", "text", "label", "split"]]
        self.img_fmt = IMG_FMT[args.task]
        self.task_name = TASKS[args.task]
        self.classes = CLASSES.get(args.task)
        self.dropout, self.weight_decay, self.lr = args.dropout, args.weight_decay, args.lr
        self.max_length = 40 if model_name == "vilt" else 128
        if multimodal:
            self.use_clip_loss, self.use_tim_loss, self.use_iadds_loss = args.use_clip_loss, args.use_tim_loss, False
            self.beta_itc = args.beta_itc if self.use_clip_loss else None
            self.beta_itm = args.beta_itm if self.use_tim_loss else None
            self.beta_iadds = None
            self.loss_str = ""
            if args.use_clip_loss:
                self.loss_str += "itc{}".format(str(self.beta_itc))
            if args.use_tim_loss:
                self.loss_str += "itm{}".format(str(self.beta_itm))
        self.use_loss_correction = False
