"""Oracle (test infrastructure): NeRF / segmentation losses, Adam, post-proc.

* ``nerf_losses`` follows reference
  ``nr4seg/lightning/joint_train_lightning_net.py:180-223`` and the weighting
  at :44-45, :503-507.
* ``seg_loss`` follows :37-38 and :456-458 (CrossEntropy applied to the
  *softmax output*, i.e. a double softmax -- reproduced, not fixed).
* ``semantic_postproc`` follows :246-251.
* ``adam_step`` follows torch.optim.Adam as configured at :897-919
  (betas (0.9, 0.99), eps 1e-15, L2 weight decay folded into the gradient for
  the "net" group).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

WEIGHT_DEPTH = 0.1
WEIGHT_SEMANTICS = 0.04


def nerf_losses(pred_rgb, pred_sem, pred_depth, gt_rgb, labels, gt_depth,
                one_m_to_scene_uom: float):
    """All [B,N,...].  Returns (loss_color, loss_semantics|None, loss_depth).
    ``labels`` is modified like the reference does (invalid rows -> -1)."""
    sem = pred_sem.clone()
    labels = labels.clone()
    invalid = torch.sum(sem, dim=-1) == 0
    sem[invalid] = 1
    sem = sem / torch.sum(sem, dim=-1, keepdim=True)
    labels[invalid] = -1
    loss_color = ((pred_rgb - gt_rgb)**2).mean()
    if int(invalid.sum()) == sem.shape[1]:
        loss_sem = None
    else:
        logp = torch.log(sem + 1e-15).permute(0, 2, 1)
        loss_sem = F.nll_loss(logp, labels, ignore_index=-1,
                              reduction="none").mean()
    valid = gt_depth != 0
    loss_depth = torch.abs(pred_depth[valid] / one_m_to_scene_uom -
                           gt_depth[valid]).mean(-1)
    return loss_color, loss_sem, loss_depth


def nerf_total_loss(lc, ls, ld):
    total = lc
    if ls is not None:
        total = total + ls * WEIGHT_SEMANTICS
    if ld is not None:
        total = total + ld * WEIGHT_DEPTH
    return total


def seg_loss(logits, labels):
    """logits [B,C,H,W], labels [B,H,W] int64 (-1 ignored) -> scalar."""
    pred = F.softmax(logits, dim=1)
    per_px = F.cross_entropy(pred, labels, ignore_index=-1, reduction="none")
    return per_px.mean(), pred


def semantic_postproc(sem):
    """[..., C] composited probabilities -> (normalised, argmax)."""
    sem = sem.clone()
    invalid = torch.sum(sem, dim=-1) == 0
    sem[invalid] = 1
    sem = sem / torch.sum(sem, dim=-1, keepdim=True)
    return sem, torch.argmax(sem, dim=-1)


def adam_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.99,
              eps=1e-15, weight_decay=0.0):
    """One torch.optim.Adam update (non-AMSGrad, L2 decay), returns new
    (p, m, v); ``step`` is 1-based."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = (v.sqrt() / (bc2**0.5)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def psnr(pred, gt):
    """Not in the reference: -10 log10 MSE on [0,1] RGB."""
    mse = torch.mean((pred - gt)**2)
    return float(-10.0 * torch.log10(mse))
