"""Oracle (test infrastructure): confusion-matrix meter.

Follows reference ``nr4seg/utils/metrics.py:13-65`` (``SemanticsMeter``):
rows = ground truth, ``truth == -1`` dropped, mIoU averaged over classes that
have at least one ground-truth pixel, total accuracy, class-average accuracy
ignoring absent classes.
"""
from __future__ import annotations

import numpy as np


def confusion(preds, truths, n_classes: int) -> np.ndarray:
    preds = np.asarray(preds).reshape(-1)
    truths = np.asarray(truths).reshape(-1)
    keep = truths != -1
    preds, truths = preds[keep], truths[keep]
    # sklearn's confusion_matrix(labels=range(C)) drops pairs outside labels
    ok = (preds >= 0) & (preds < n_classes) & (truths >= 0) & (truths <
                                                               n_classes)
    cm = np.zeros((n_classes, n_classes), dtype=np.int64)
    np.add.at(cm, (truths[ok].astype(np.int64), preds[ok].astype(np.int64)), 1)
    return cm


def measure(cm: np.ndarray):
    cm = cm.astype(np.int64)
    rows = cm.sum(axis=1).astype(np.float64)
    cols = cm.sum(axis=0).astype(np.float64)
    diag = np.diagonal(cm).astype(np.float64)
    present = rows > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        per_class_acc = diag / rows
        ious = diag / (rows + cols - diag)
    class_avg_acc = float(np.mean(per_class_acc[present]))
    total_acc = float(diag.sum() / cm.sum())
    miou = float(np.mean(ious[present]))
    return miou, total_acc, class_avg_acc
