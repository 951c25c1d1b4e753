#!/usr/bin/env python3
"""Golden vectors for the evaluation loop, produced by the REFERENCE's own
`MMLate_Model.eval` (models/mm_late.py:534-638; this container only, behind the shim of make_golden.py): a
three-batch loader (4 + 4 + 3 posts, items shaped like MM_Dataset's: [B,1,T] ids / mask, [B,1,3,224,224] pixels, one-hot int64
labels, data ids) through the reference MM_Model with the oracle's deterministic weights, for the plain loss and for the
ITC + ITM mix (ITM negatives re-sampled per batch from numpy's global stream, seeded 30 as run_mm_late.py:49 does).
Stored: the returned dict (data_id, loss, predictions, labels) plus the per-batch class logits; the inputs are regenerated
from the seeds by the tests (oracle.synthetic_batch), like the forward goldens.
Run:  python tests/golden/make_eval_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import install_shim, build_reference_model, load_params  # noqa: E402
from oracle import mm_oracle as O  # noqa: E402

BATCHES = ((4, 31), (4, 32), (3, 33))          # (posts, seed of oracle.synthetic_batch)
T = 64
CLASS_W = [0.7, 1.6, 0.9]


def loader(cfg):
    out, first = [], 5000
    for B, seed in BATCHES:
        ids, mask, pixels, onehot = O.synthetic_batch(cfg, B, T, seed, True)
        out.append({"input_ids": ids.unsqueeze(1), "attention_mask": mask.unsqueeze(1), "pixel_values": pixels.unsqueeze(1),
                    "labels": onehot, "data_id": torch.arange(first, first + B)})
        first += B
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_config = install_shim()
    cfg = O.OracleConfig(layers_txt=2, layers_img=2, vocab=1000, max_pos=130, num_labels=3)
    with tempfile.TemporaryDirectory() as tmp:
        ref_mm, model = build_reference_model(ref_config, cfg, "bernice", tmp, 0.1)
    load_params(model, O.make_params(cfg, 0))
    ref_mm.device = torch.device("cpu")
    loss_fn = torch.nn.CrossEntropyLoss(weight=torch.tensor(CLASS_W))
    tim_loss_fn = torch.nn.CrossEntropyLoss()
    out = dict(cfg=np.array(repr(O.asdict(cfg))), T=T, seed_w=0, batches=np.array(BATCHES), class_w=np.array(CLASS_W, dtype=np.float32))
    for tag, itc, itm in (("plain", False, False), ("itcitm", True, True)):
        tr = ref_mm.MMLate_Model.__new__(ref_mm.MMLate_Model)
        tr.model, tr.cnn, tr.multilabel = model, False, False
        tr.use_clip_loss, tr.use_tim_loss, tr.use_iadds_loss, tr.use_loss_correction = itc, itm, False, False
        tr.beta_itc, tr.beta_itm, tr.beta_iadds = 0.1, 0.1, 0.0
        tr.softmax = torch.nn.Softmax(dim=1)
        np.random.seed(30)
        res = tr.eval(loader(cfg), loss_fn, tim_loss_fn=tim_loss_fn if itm else None)
        assert not model.training
        out[tag + ".data_id"] = res["data_id"].numpy()
        out[tag + ".loss"] = np.float64(res["loss"])
        out[tag + ".predictions"] = res["predictions"].numpy()
        out[tag + ".labels"] = res["labels"].numpy()
        print(tag, "loss", res["loss"], "pred", res["predictions"].tolist(), "labels", res["labels"].tolist())
    # class logits per batch (what the argmax was taken of), so a test can tell a near-tie from a wrong prediction
    with torch.no_grad():
        logits = [model(torch.squeeze(b["input_ids"]), torch.squeeze(b["attention_mask"]), torch.squeeze(b["pixel_values"]))[0] for b in loader(cfg)]
    out["out_cls"] = torch.cat(logits).numpy()
    np.savez_compressed(os.path.join(HERE, "eval_small_xlmr.npz"), **out)


if __name__ == "__main__":
    main()
