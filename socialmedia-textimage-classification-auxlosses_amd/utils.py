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


def _stats(pred, target, num_classes):
    tp, fp, fn = (np.zeros(num_classes) for _ in range(3))
    for c in range(num_classes):
        tp[c] = np.sum((pred == c) & (target == c))
        fp[c] = np.sum((pred == c) & (target != c))
        fn[c] = np.sum((pred != c) & (target == c))
    return tp, fp, fn


def _safe_div(a, b):
    return np.divide(a, b, out=np.zeros_like(a, dtype=np.float64), where=b != 0)


def _reduce(score, average, tp, fp, fn):
    """torchmetrics 0.11 (`timrel-env.yml:120`) `_adjust_weights_safe_divide`, multiclass: 'weighted' weighs by support
    tp + fn; 'macro' averages over the classes that occur in predictions or labels (a class absent from both is left out)"""
    w = tp + fn if average == "weighted" else ((tp + fp + fn) != 0).astype(np.float64)
    return float(np.sum(_safe_div(w * score, np.full_like(score, w.sum()))))


def compute_metrics(res, num_classes, multi_label=False, multilabel=None):
    """reference models/utils.py:294-325 without torchmetrics: {"metric": [...], "result": [...]} with weighted / macro F1,
    precision, recall (torchmetrics 0.11 multiclass definitions, zero division -> 0) and the loss, in that order.
    (`multilabel=` is what the reference's load branch passes, run_mm_late.py:177 -- a TypeError there; both are taken.)"""
    if multi_label or multilabel:
        raise NotImplementedError("no task of the reference enables the multilabel branch (config.py:10)")
    pred, target = np.asarray(res["predictions"]).reshape(-1), np.asarray(res["labels"]).reshape(-1)
    tp, fp, fn = _stats(pred, target, num_classes)
    prec, rec, f1 = _safe_div(tp, tp + fp), _safe_div(tp, tp + fn), _safe_div(2 * tp, 2 * tp + fp + fn)
    results = {}
    for name, score in (("f1", f1), ("precision", prec), ("recall", rec)):
        for avg in ("weighted", "macro"):
            results[f"{name}_{avg}"] = _reduce(score, avg, tp, fp, fn)
    results["loss"] = res["loss"]
    return {"metric": list(results), "result": list(results.values())}


def agg_metrics_val(res_val, metric_names, num_labels):
    """reference models/utils.py:327-335: column `metric` + one column `epoch-N` (1-based) per evaluated epoch"""
    out = {"metric": metric_names}
    for predictions in res_val:
        m = compute_metrics(predictions, num_labels)
        d = dict(zip(m["metric"], m["result"]))
        out["epoch-" + str(predictions["epoch"] + 1)] = [d[k] for k in metric_names]
    return out


def balanced_class_weights(labels, num_classes):
    """sklearn compute_class_weight('balanced'): n / (k * count_c)   (reference models/utils.py:173-178)"""
    labels = np.asarray(labels)
    counts = np.array([max(1, int((labels == c).sum())) for c in range(num_classes)], dtype=np.float64)
    return torch.tensor(len(labels) / (num_classes * counts), dtype=torch.float32)
