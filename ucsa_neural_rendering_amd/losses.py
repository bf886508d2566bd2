"""Autograd wrappers of the fused loss kernels.

``nerf_losses`` returns what reference ``forward_nerf_train`` returns
(joint_train_lightning_net.py:208-223): ``(loss_color, loss_semantics | None,
loss_depth)``; the three scalars share one fused forward+backward kernel
(``ucsa_nerf_loss``): the gradient of ``color + 0.04 sem + 0.1 depth`` wrt the
rendered outputs is produced in the same pass, and the returned scalars carry
an autograd node that hands those gradients back, so
``total = lc + ls*0.04 + ld*0.1; total.backward()`` works exactly like in the
reference (:497-513), including under ``GradScaler``.
"""
from __future__ import annotations

import torch

from . import ops

WEIGHT_DEPTH = 0.1  # reference :44
WEIGHT_SEMANTICS = 0.04  # reference :45


class _NerfLossFn(torch.autograd.Function):
    """Outputs the three loss terms, the semantics term with the reference's
    ``None`` as a zero, and their reference-weighted TOTAL -- all views of the
    kernel's ``stats`` vector (one fused forward: ucsa_nerf_loss also writes
    the gradient of the total wrt the rendered outputs).  The backward is ONE
    launch whatever the caller did with the outputs (ucsa_nerf_loss_apply:
    stored gradients x the cotangents, which stay on the device): the ~20
    elementwise launches autograd used to spend per step on ``lc + 0.04 ls +
    0.1 ld`` and back (190 us of a 4.9 ms step) are gone."""

    @staticmethod
    def forward(ctx, rgb, sem, depth, gt_rgb, labels, gt_depth, uom):
        stats, g = ops.nerf_loss(rgb, sem, depth, gt_rgb, labels, gt_depth, uom,
                                 w_sem=WEIGHT_SEMANTICS, w_depth=WEIGHT_DEPTH,
                                 grad_scale=1.0, want_grad=True)
        ctx.save_for_backward(*g)
        ctx.shapes = (rgb.shape, sem.shape, depth.shape)
        ctx.set_materialize_grads(False)
        return stats[0], stats[1], stats[2], stats[6], stats[5]

    @staticmethod
    def backward(ctx, g_c, g_s, g_d, g_s0, g_t):
        grads = ctx.saved_tensors
        sr, ss, sd = ctx.shapes
        if g_s is not None and g_s0 is not None:
            g_s = g_s + g_s0
        elif g_s0 is not None:
            g_s = g_s0
        if g_c is None and g_s is None and g_d is None and g_t is None:
            return (None,) * 7
        d_rgb, d_sem, d_depth = ops.nerf_loss_apply(grads, g_t, g_c, g_s, g_d,
                                                    WEIGHT_SEMANTICS, WEIGHT_DEPTH)
        return d_rgb.view(sr), d_sem.view(ss), d_depth.view(sd), None, None, None, None


def nerf_losses(pred_rgb, pred_sem, pred_depth, gt_rgb, labels, gt_depth,
                one_m_to_scene_uom, none_if_invalid=False):
    """-> (loss_color, loss_semantics, loss_depth) like the reference.

    When every ray has invalid semantics the reference returns
    ``loss_semantics = None`` ("no gradient flow", :212-213) and the caller
    skips the term.  Finding that out on the host costs a device
    synchronisation per training step (measured: the step becomes host-bound,
    2.0 instead of 1.5 ms through the marcher), so by default the term comes
    back as a zero with zero gradient instead -- the same total loss and the
    same gradients.  ``none_if_invalid=True`` gives the reference's ``None``
    (one read-back).

    The three terms remember the kernel's own total: ``nerf_total_loss`` of
    exactly these three returns it (same value: ``lc + ls*0.04`` then
    ``+ ld*0.1`` in fp32) instead of re-deriving it with elementwise ops."""
    lc, ls_raw, ld, ls0, total = _NerfLossFn.apply(
        pred_rgb, pred_sem, pred_depth, gt_rgb, labels, gt_depth,
        float(one_m_to_scene_uom))
    if none_if_invalid:
        ls = None if bool(torch.isnan(ls_raw.detach())) else ls_raw
    else:
        ls = ls0
    lc._ucsa_total = (total, ls, ld)
    return lc, ls, ld


def nerf_total_loss(lc, ls, ld):
    fused = getattr(lc, "_ucsa_total", None)
    if fused is not None and fused[1] is ls and fused[2] is ld and ls is not None:
        return fused[0]
    total = lc
    if ls is not None:
        total = total + ls * WEIGHT_SEMANTICS
    if ld is not None:
        total = total + ld * WEIGHT_DEPTH
    return total


class _SegLossFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, labels):
        r = ops.seg_tail(logits, labels, want_prob=False, want_grad=True)
        ctx.save_for_backward(r["d_logits"])
        return r["loss"].view(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None


def seg_loss(logits, labels):
    """CrossEntropyLoss(ignore_index=-1, reduction="none")(softmax(out),
    label).mean()  -- reference :37-38, :456-458."""
    return _SegLossFn.apply(logits, labels)
