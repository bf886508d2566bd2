"""Mirror of reference ``nr4seg/utils/metrics.py`` (``SemanticsMeter``).

GPU label maps are counted by the HIP kernel ``ucsa_confusion_matrix``
(integer atomics: exact); CPU inputs (numpy / CPU tensors, as the reference's
callers sometimes pass) are counted with numpy.  ``measure`` follows :48-65:
mIoU over classes with at least one ground-truth pixel, total accuracy,
class-average accuracy ignoring absent classes.  The 40x40 matrix can be
summed across ranks (``dist.allreduce_confusion_``)."""
from __future__ import annotations

import numpy as np
import torch


class SemanticsMeter:
    """``conf_mat`` (numpy int64 [C,C] or None) as in the reference; counts of
    GPU label maps accumulate in a device tensor and are folded into it only
    when it is read (``conf_mat`` / ``measure``): an ``update`` per evaluated
    frame no longer costs a device synchronisation (3.7 s of a 63 s continual-loop
    profile, round 5)."""

    def __init__(self, number_classes):
        self._host = None
        self._dev = None
        self.number_classes = number_classes

    def clear(self):
        self._host = None
        self._dev = None

    @property
    def conf_mat(self):
        if self._dev is not None:
            cm = self._dev.cpu().numpy().astype(np.int64)
            self._dev = None
            self._host = cm if self._host is None else self._host + cm
        return self._host

    @conf_mat.setter
    def conf_mat(self, value):
        self._dev = None
        self._host = value

    def update(self, preds, truths):
        C = self.number_classes
        if torch.is_tensor(preds) and preds.is_cuda:
            from .. import ops
            cm = ops.confusion_matrix(preds, truths.to(preds.device), C)
            if self._dev is not None and self._dev.device != cm.device:
                self.conf_mat       # fold the other device's counts first
            self._dev = cm if self._dev is None else self._dev + cm
            return
        p = preds.detach().cpu().numpy() if torch.is_tensor(preds) else np.asarray(preds)
        t = truths.detach().cpu().numpy() if torch.is_tensor(truths) else np.asarray(truths)
        p, t = p.reshape(-1), t.reshape(-1)
        ok = (t >= 0) & (t < C) & (p >= 0) & (p < C)
        cm = np.zeros((C, C), dtype=np.int64)
        np.add.at(cm, (t[ok].astype(np.int64), p[ok].astype(np.int64)), 1)
        self._host = cm if self._host is None else self._host + cm

    def measure(self):
        cm = self.conf_mat.astype(np.int64)
        rows = cm.sum(axis=1).astype(np.float64)
        cols = cm.sum(axis=0).astype(np.float64)
        diag = np.diagonal(cm).astype(np.float64)
        present = rows > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            acc_c = diag / rows
            ious = diag / (rows + cols - diag)
        class_average_accuracy = float(np.mean(acc_c[present]))
        total_accuracy = float(diag.sum() / cm.sum())
        miou_valid_class = float(np.mean(ious[present]))
        return miou_valid_class, total_accuracy, class_average_accuracy
