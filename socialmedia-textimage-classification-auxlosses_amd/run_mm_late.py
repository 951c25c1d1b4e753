"""Command line of the late-fusion runs -- same flags, defaults, file names and CSV layouts as the reference's
models/run_mm_late.py:20-191.  Additive flags: --batch_size, --synthetic/--n_synthetic (no dataset on disk),
--dtype, --results_dir, --cpu_preprocess, --num_workers, --cache_vision.  Data parallel: launch with `python -m torch.distributed.run --nproc-per-node N ...`.

    python -m smtc_amd.run_mm_late --txt_model_name bernice --img_model_name vit --fusion_name attention --task 2 --testing
"""
import argparse
import logging
import os
import sys

import numpy as np
import pandas as pd
import torch

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import smtc_amd  # noqa: F401
    __package__ = "smtc_amd"

from . import dist as mmdist
from .config import Config, results_dir_mm_late, TEXT_ARCH
from .mm_late import MMLate_Model
from .synthetic import SyntheticPosts
from .utils import compute_metrics, balanced_class_weights

logging.basicConfig(format="%(asctime)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S", level=logging.INFO)
logger = logging.getLogger(__name__)


def build_parser():
    p = argparse.ArgumentParser(description="run late fusion models")
    # reference flags, models/run_mm_late.py:21-43 (names, types, choices, defaults unchanged)
    p.add_argument("--txt_model_name", type=str, choices=["bert", "bernice", "bertweet", "roberta"], help="model name")
    p.add_argument("--img_model_name", type=str, choices=["vit", "beit", "deit", "resnet50", "resnet152", "clip", "clip336"],
                   help="model name (clip / clip336: CLIP-ViT-L/14 vision tower at 224 / 336, additive: BASELINE config 4; use --fusion_name concat)")
    p.add_argument("--fusion_name", type=str, choices=["xatt", "concat", "attention", "concat_cnn", "aspect-att", "gmu"], help="fusion method")
    p.add_argument("--use_clip_loss", action="store_true", help="use contrastive Loss")
    p.add_argument("--use_tim_loss", action="store_true", help="use TIM Loss")
    p.add_argument("--use_iadds_loss", action="store_true", help="use image-adds loss")
    p.add_argument("--beta_iadds", type=float, default=0.1, help="hyperparameter for iadds loss")
    p.add_argument("--beta_itc", type=float, default=0.1, help="hyperparameter for itc loss")
    p.add_argument("--beta_itm", type=float, default=0.1, help="hyperparameter for itm loss")
    p.add_argument("--use_loss_correction", action="store_true", help="use Loss correction (only for binary cases)")
    p.add_argument("--task", type=int, choices=[0, 1, 2, 3, 4, 5, 6], help="task to run")
    p.add_argument("--epochs", type=int, default=2, help="number of epochs")
    p.add_argument("--weight_decay", type=float, default=0.00025, help="weight decay param")
    p.add_argument("--lr", type=float, default=1e-5, help="learning rate param")
    p.add_argument("--dropout", type=float, default=0.05, help="dropout param")
    p.add_argument("--seed", type=int, default=30, help="manual seed")
    p.add_argument("--nsamples", type=int, default=-1, help="number of training samples")
    p.add_argument("--testing", action="store_true", help="testing sample")
    p.add_argument("--eval_txt_test", action="store_true", help="eval txt test")
    p.add_argument("--save_model", action="store_true", help="save model")
    p.add_argument("--load_saved_model", action="store_true", help="load saved model")
    p.add_argument("--save_preds", action="store_true", help="eval test")
    p.add_argument("--use_saved_features", action="store_true", help="use preprocessed features")
    # additive
    p.add_argument("--batch_size", type=int, default=None, help="per-GPU batch size (default: the reference's per-task value)")
    p.add_argument("--synthetic", action="store_true", help="synthetic posts instead of the data key / images")
    p.add_argument("--n_synthetic", type=int, default=256, help="synthetic training posts per rank")
    p.add_argument("--dtype", choices=["bf16", "f16", "bf16x3"], default="bf16", help="bf16x3 = strict-parity mode (fp32 activations, 3 bf16 MFMA products per Linear)")
    p.add_argument("--results_dir", type=str, default=None, help="default ../results/mm_late/ as in the reference")
    p.add_argument("--arch_layers", type=int, default=None, help="(testing) override encoder depth")
    p.add_argument("--cpu_preprocess", action="store_true", help="resize / normalize images on the host (PIL) instead of the GPU kernels")
    p.add_argument("--num_workers", type=int, default=0, help="DataLoader workers (reference: 0); 8 or more keep the image decode ahead of the GPU step (tools/loader_bench.py)")
    p.add_argument("--item_tokenize", action="store_true", help="tokenise per item like the reference instead of once per batch in the collate")
    p.add_argument("--cache_vision", type=int, default=0, metavar="POSTS",
                   help="keep the frozen image tower's outputs of up to POSTS posts in HBM (306 KB each): epochs after the first skip the tower")
    return p


def file_names(args, results_dir, loss_str):
    """reference models/run_mm_late.py:88-96,124-127"""
    nsamples_str = "" if args.nsamples == -1 else "N" + str(args.nsamples) + "_"
    stem = results_dir + "{}-{}-{}_task{}_seed{}_{}_{}".format(args.txt_model_name, args.img_model_name, args.fusion_name, args.task, args.seed,
                                                               loss_str, nsamples_str)
    return {"model": stem + "net.pth", "val": stem + "metrics_val.csv", "test": stem + "metrics_test.csv", "preds": stem + "preds.csv"}


def make_loaders(args, cfg, trainer):
    """synthetic posts (this environment ships header-only data keys: SURVEY.md 4); real data keys go through
    smtc_amd.datasets when the tokenizer / images exist"""
    a = trainer.model.arch
    if cfg.data is not None and not args.synthetic:
        from .datasets import loaders_from_data_key
        if int(getattr(args, "num_workers", 0) or 0) > 0 and trainer.device.type == "cuda":
            trainer.warm_start()                 # the first step before the DataLoader forks its workers (MMLate_Model.warm_start)
        return loaders_from_data_key(cfg, args, trainer)
    n = 200 if args.testing else args.n_synthetic                        # --testing subsamples 200 rows (models/utils.py:135-138)
    rank = mmdist.rank()
    mk = lambda cnt, seed: SyntheticPosts(cnt, a["vocab"], cfg.num_labels, cfg.max_length, seed, a["txt_kind"], a["pad_id"], a["image"])
    tr, va, te = mk(n, 11 + rank), mk(max(cfg.batch_size, n // 4), 1011), mk(max(cfg.batch_size, n // 4), 2011)
    dl = lambda ds, shuffle: torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=shuffle, drop_last=False)
    labels = [int(tr[i]["labels"].argmax()) for i in range(min(len(tr), 512))]
    return dl(tr, True), dl(va, False), dl(te, False), balanced_class_weights(labels, cfg.num_labels)


def main(argv=None):
    args = build_parser().parse_args(argv)
    for flag in ("eval_txt_test", "use_saved_features"):
        if getattr(args, flag):
            # text-only test split (models/mm_late.py:372-376) and pre-extracted processor outputs (models/datasets.py:155-158):
            # outside the hot path (DESIGN.md "Out of scope"); refused rather than silently ignored
            raise NotImplementedError(f"--{flag} is not part of this build")
    mmdist.init_from_env()
    # the host side of a step is a handful of tiny tensor ops; torch's intra-op pool defaults to one thread per host core (256) and those
    # threads spin after every parallel region -- measured on the 16-core CPU share of a GPU box: the training process burned 15 cores
    # spinning and left 1.4 to the eight DataLoader workers (tools/loader_bench.py, profiles/r03_loader_bench.txt)
    torch.set_num_threads(max(1, int(os.environ.get("MMHIP_HOST_THREADS", "4"))))
    torch.manual_seed(args.seed)                      # models/run_mm_late.py:48-49 (same on every rank: identical initial weights)
    np.random.seed(args.seed + mmdist.rank())         # ITM negative sampling draws from numpy: its own stream per rank
    results_dir = args.results_dir or results_dir_mm_late
    if args.testing:
        results_dir += "testing/"
    logger.info("Model: {}-{}, Task: {}, Fusion: {}, Testing: {}, ITC Loss: {}, TIM Loss: {}, beta_itc: {}, beta_itm: {}, NSamples: {}, seed: {}".format(
        args.txt_model_name, args.img_model_name, args.task, args.fusion_name, args.testing, args.use_clip_loss, args.use_tim_loss,
        args.beta_itc, args.beta_itm, args.nsamples, args.seed))
    cfg = Config(args)
    kw = dict(dtype=args.dtype, seed=args.seed)
    if args.arch_layers:
        kw["arch"] = dict(layers_txt=args.arch_layers, layers_img=args.arch_layers)
    trainer = MMLate_Model(cfg, args.txt_model_name, args.img_model_name, args.fusion_name, multilabel=cfg.multilabel, **kw)
    if args.cache_vision > 0:
        trainer.model.enable_vision_cache(args.cache_vision)
    train_loader, val_loader, test_loader, weight = make_loaders(args, cfg, trainer)
    names = file_names(args, results_dir, cfg.loss_str)
    model_path = names["model"] if (args.save_model or args.load_saved_model) else None
    if mmdist.rank() == 0:
        os.makedirs(results_dir, exist_ok=True)       # the reference requires the directory to pre-exist
    if not args.load_saved_model:
        logger.info("Training")
        trainer.train(train_loader, val_loader, args.epochs, None, cfg.lr, cfg.weight_decay, te_dataloader=test_loader, model_path=model_path,
                      val_filename=names["val"], te_filename=names["test"], class_weight=weight)
        vc = getattr(trainer.model, "_vcache", None)
        if vc is not None:
            logger.info("image-tower output cache: %d posts served from HBM, %d computed, %d cached", vc["hits"], vc["misses"], len(vc["slots"]))
        if args.save_preds and mmdist.rank() == 0:
            pred = trainer.eval(test_loader, class_weight=weight)
            pd.DataFrame({"data_id": pred["data_id"].tolist(), "label": pred["labels"].tolist(),
                          "prediction": pred["predictions"].tolist()}).to_csv(names["preds"], index=False)
            logger.info("%s saved", names["preds"])
    else:
        logger.info("Loading %s", model_path)
        trainer.load_saved_model(model_path)
        pred = trainer.eval(test_loader, class_weight=weight)
        if mmdist.rank() == 0:
            stem = names["preds"][: -len("preds.csv")]
            pd.DataFrame({"data_id": pred["data_id"].tolist(), "label": pred["labels"].tolist(),
                          "prediction": pred["predictions"].tolist()}).to_csv(stem + "preds_lm.csv", index=False)
            pd.DataFrame(compute_metrics(pred, cfg.num_labels)).to_csv(stem + "metrics_lm.csv", index=False)      # columns metric, result
    if mmdist.world_size() > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
