"""Helpers with the reference's names (models/utils.py): clip_loss / contrastive_loss (:225-231),
get_optimizer_params (:280-292), metrics (:294-335, re-stated without torchmetrics)."""
import numpy as np
import torch
import torch.nn.functional as F


def contrastive_loss(logits: torch.Tensor) -> torch.Tensor:
    return F.cross_entropy(logits, torch.arange(len(logits), device=logits.device))


def clip_loss(similarity: torch.Tensor) -> torch.Tensor:
    return (contrastive_loss(similarity) + contrastive_loss(similarity.t())) / 2.0


def get_optimizer_params(named_parameters, weight_decay, lr, verbose=False):
    params = {"lr": lr, "weight_decay": weight_decay, "params": []}
    for name, param in named_parameters:
        if verbose:
            print(name)
        if param.requires_grad:
            params["params"].append(param)
    return [params]


def _prf(pred, target, num_classes):
    """per-class precision / recall / f1 and supports (torchmetrics Multiclass* semantics, zero_division -> 0)"""
    p, r, f, sup = (np.zeros(num_classes) for _ in range(4))
    for c in range(num_classes):
        tp = float(np.sum((pred == c) & (target == c)))
        fp = float(np.sum((pred == c) & (target != c)))
        fn = float(np.sum((pred != c) & (target == c)))
        p[c] = tp / (tp + fp) if tp + fp > 0 else 0.0
        r[c] = tp / (tp + fn) if tp + fn > 0 else 0.0
        f[c] = 2 * p[c] * r[c] / (p[c] + r[c]) if p[c] + r[c] > 0 else 0.0
        sup[c] = tp + fn
    return p, r, f, sup


def compute_metrics(res, num_classes, multi_label=False):
    """reference models/utils.py:294-325: weighted / macro F1, precision, recall + loss, in metric_names order."""
    pred, target = np.asarray(res["predictions"]), np.asarray(res["labels"])
    p, r, f, sup = _prf(pred, target, num_classes)
    w = sup / max(sup.sum(), 1.0)
    return {"f1_weighted": float((f * w).sum()), "f1_macro": float(f.mean()), "precision_weighted": float((p * w).sum()),
            "precision_macro": float(p.mean()), "recall_weighted": float((r * w).sum()), "recall_macro": float(r.mean()),
            "loss": float(res["loss"])}


def agg_metrics_val(res_list, metric_names, num_classes):
    """reference models/utils.py:327-335: column `metric` + one column `epoch-N` per evaluated epoch."""
    out = {"metric": list(metric_names)}
    for res in res_list:
        m = compute_metrics(res, num_classes)
        out["epoch-{}".format(res["epoch"])] = [m[k] for k in metric_names]
    return out


def balanced_class_weights(labels, num_classes):
    """sklearn compute_class_weight('balanced'): n / (k * count_c)   (reference models/utils.py:173-178)"""
    labels = np.asarray(labels)
    counts = np.array([max(1, int((labels == c).sum())) for c in range(num_classes)], dtype=np.float64)
    return torch.tensor(len(labels) / (num_classes * counts), dtype=torch.float32)
