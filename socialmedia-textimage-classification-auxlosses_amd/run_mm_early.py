"""Command line of the early-fusion runs -- flags, defaults and file names of the reference's models/run_mm_early.py:17-118, LXMERT branch.
The reference reads 36 x 2048 ROI features and boxes per post from `<DATA>/<task>_img_feats/{features,boxes}/...` (models/datasets.py:291-294),
files produced by its offline Faster-RCNN extraction.  The data-key path reads exactly those files (datasets.Lxmert_Dataset) with the
tokenizer of config.MODEL_DIR_DICT["lxmert"]; the extraction itself is not part of this build.  --synthetic (additive flag, as in
run_mm_late.py) serves random posts of the same shapes when no data exist (tools/make_dummy_task.py --roi writes a small real tree).
Additive flags: --synthetic / --n_synthetic, --batch_size, --dtype, --results_dir, --arch_layers, --num_workers.  Data parallel: `python -m torch.distributed.run --nproc-per-node N -m smtc_amd.run_mm_early ...`.

    python -m smtc_amd.run_mm_early --model lxmert --task 3 --epochs 1 --use_clip_loss --use_tim_loss --synthetic
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
from .config import Config, RES_PATH
from .mm_early import MMEarly_Model
from .utils import compute_metrics, balanced_class_weights

logging.basicConfig(format="%(asctime)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S", level=logging.INFO)
logger = logging.getLogger(__name__)
results_dir_mm_early = RES_PATH + "mm_early/"


class SyntheticLxmertPosts(torch.utils.data.Dataset):
    """items shaped like the reference's Lxmert_Dataset (models/datasets.py:262-301): [1,T] ids / mask / token types, 36 x 2048 features,
    36 x 4 normalised boxes, one-hot labels, data id"""

    def __init__(self, n, vocab, num_labels, T, seed):
        g = torch.Generator().manual_seed(seed)
        self.ids = torch.randint(1, vocab, (n, T), generator=g)
        lens = torch.randint(4, T + 1, (n,), generator=g)
        self.mask = (torch.arange(T)[None, :] < lens[:, None]).long()
        self.ids = self.ids * self.mask
        self.ids[:, 0] = 101
        self.labels = torch.nn.functional.one_hot(torch.randint(0, num_labels, (n,), generator=g), num_labels)
        self.seed, self.n = seed, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return {"input_ids": self.ids[i][None], "attention_mask": self.mask[i][None], "token_type_ids": torch.zeros_like(self.ids[i])[None],
                "features": torch.rand(36, 2048, generator=g) * 2, "normalized_boxes": torch.rand(36, 4, generator=g),
                "labels": self.labels[i], "data_id": torch.tensor(1000 + i)}


def build_parser():
    p = argparse.ArgumentParser(description="run early fusion models")
    # reference flags, models/run_mm_early.py:17-37
    p.add_argument("--model", type=str, choices=["vilt", "lxmert"], help="model name")
    p.add_argument("--use_clip_loss", action="store_true", help="use CLIP Loss")
    p.add_argument("--beta_itc", type=float, default=0.1, help="hyperparameter for itc loss")
    p.add_argument("--beta_itm", type=float, default=0.1, help="hyperparameter for itm loss")
    p.add_argument("--use_tim_loss", action="store_true", help="use TIM Loss")
    p.add_argument("--use_loss_correction", action="store_true", help="use Loss correction (only for binary cases)")
    p.add_argument("--task", type=int, choices=[0, 1, 2, 3, 4, 5, 6], help="task to run")
    p.add_argument("--epochs", type=int, default=2, help="number of epochs")
    p.add_argument("--weight_decay", type=float, default=0.00025, help="weight decay param")
    p.add_argument("--lr", type=float, default=1e-5, help="learning rate param")
    p.add_argument("--dropout", type=float, default=0.05, help="dropout param")
    p.add_argument("--seed", type=int, default=30, help="manual seed")
    p.add_argument("--testing", action="store_true", help="testing sample")
    p.add_argument("--evaltest", action="store_true", help="eval test")
    p.add_argument("--save_model", action="store_true", help="eval test")
    p.add_argument("--use_saved_features", action="store_true", help="use preprocessed features")
    # additive
    p.add_argument("--synthetic", action="store_true", help="synthetic posts instead of the data key + ROI-feature files")
    p.add_argument("--num_workers", type=int, default=0, help="DataLoader worker processes of the data-key path")
    p.add_argument("--n_synthetic", type=int, default=128, help="synthetic training posts per rank")
    p.add_argument("--batch_size", type=int, default=None)
    p.add_argument("--dtype", choices=["bf16", "f16", "bf16x3"], default="bf16")
    p.add_argument("--results_dir", type=str, default=None, help="default ../results/mm_early/ as in the reference")
    p.add_argument("--arch_layers", type=int, default=None, help="(testing) override the three encoder depths")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.model != "lxmert":
        raise NotImplementedError("early fusion: only --model lxmert (BASELINE config 5); ViLT is out of scope")
    if args.use_loss_correction or args.use_saved_features:
        raise NotImplementedError("--use_loss_correction / --use_saved_features are not part of this build")
    mmdist.init_from_env()
    torch.manual_seed(args.seed)                      # models/run_mm_early.py:40-41
    np.random.seed(args.seed + mmdist.rank())
    results_dir = args.results_dir or results_dir_mm_early
    if args.testing:
        results_dir += "testing/"
    logger.info("Model: {}, Task: {}, Epochs: {}, ITC loss: {}, TIM loss: {}, beta_itc: {}, beta_itm: {}, seed: {}".format(
        args.model, args.task, args.epochs, args.use_clip_loss, args.use_tim_loss, args.beta_itc, args.beta_itm, args.seed))
    cfg = Config(args, model_name=args.model)
    kw = dict(dtype=args.dtype, seed=args.seed)
    if args.arch_layers:
        kw["arch"] = dict(l_layers=args.arch_layers, r_layers=args.arch_layers, x_layers=args.arch_layers)
    trainer = MMEarly_Model(cfg, args.model, multilabel=cfg.multilabel, **kw)
    a = trainer.model.arch
    if args.synthetic:
        n = 200 if args.testing else args.n_synthetic
        mk = lambda cnt, seed: SyntheticLxmertPosts(cnt, a["vocab"], cfg.num_labels, cfg.max_length, seed)
        tr, va, te = mk(n, 11 + mmdist.rank()), mk(max(cfg.batch_size, n // 4), 1011), mk(max(cfg.batch_size, n // 4), 2011)
        dl = lambda ds, sh: torch.utils.data.DataLoader(ds, batch_size=cfg.batch_size, shuffle=sh)
        weight = balanced_class_weights([int(tr.labels[i].argmax()) for i in range(len(tr))], cfg.num_labels)
        train_loader, val_loader, test_loader = dl(tr, True), dl(va, False), dl(te, False)
    else:
        # the reference's load_data (models/mm_early.py:228-258): data key -> prepare_data -> Lxmert_Dataset over the pre-extracted ROI files
        from transformers import AutoTokenizer
        from .config import MODEL_DIR_DICT
        from .datasets import lxmert_loaders_from_data_key
        if cfg.data is None:
            raise FileNotFoundError("data key of task {} not found (config.PATH): run from <tree>/models/run, or use --synthetic".format(args.task))
        tdir = MODEL_DIR_DICT["lxmert"]
        if not os.path.isdir(tdir):
            raise FileNotFoundError(f"tokenizer directory {tdir!r} (config.MODEL_DIR_DICT['lxmert']) not found: place the model there, or use --synthetic")
        tok = AutoTokenizer.from_pretrained(tdir)
        if len(tok) > a["vocab"]:
            raise ValueError(f"tokenizer has {len(tok)} entries, the model's word table {a['vocab']}")
        train_loader, val_loader, test_loader, weight = lxmert_loaders_from_data_key(cfg, args, tok)
    stem = results_dir + "{}_task{}_seed{}_{}_".format(args.model, args.task, args.seed, cfg.loss_str)      # :66-74
    if mmdist.rank() == 0:
        os.makedirs(results_dir, exist_ok=True)
    logger.info("Training...")
    trainer.train(train_loader, val_loader, args.epochs, None, cfg.lr, cfg.weight_decay, te_dataloader=test_loader,
                  model_path=stem + "net.pth" if args.save_model else None, val_filename=stem + "metrics_val.csv", te_filename=stem + "metrics_test.csv",
                  class_weight=weight)
    if args.evaltest and mmdist.rank() == 0:          # :88-115
        pred = trainer.eval(test_loader, class_weight=weight)
        metrics = compute_metrics(pred, cfg.num_labels)
        print(metrics)
        if not args.testing:
            pd.DataFrame({"data_id": pred["data_id"].tolist(), "label": pred["labels"].tolist(), "prediction": pred["predictions"].tolist()}).to_csv(
                stem + "preds.csv", index=False)
            pd.DataFrame(metrics).to_csv(stem + "metrics.csv", index=False)
    if mmdist.world_size() > 1:
        torch.distributed.destroy_process_group()
    logger.info("Done!")


if __name__ == "__main__":
    main()
