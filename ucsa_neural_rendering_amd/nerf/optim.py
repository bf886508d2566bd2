"""Adam for the NeRF parameter groups, backed by the HIP kernel
``ucsa_adam_step``.

Same constructor contract as ``torch.optim.Adam`` as the reference calls it
(nr4seg/lightning/joint_train_lightning_net.py:897-919): named param groups,
``lr``, ``betas``, ``eps``, per-group ``weight_decay`` (L2 folded into the
gradient, not AdamW).  Works under ``torch.cuda.amp.GradScaler``
(``scaler.step(optimizer)`` unscales the grads, then calls ``step``)."""
from __future__ import annotations

import torch

from .. import ops


class HipAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("HipAdam updates GPU parameters only "
                                       "(no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                ops.adam_step(p.data, p.grad.contiguous(), st["exp_avg"],
                              st["exp_avg_sq"], st["step"], group["lr"], b1, b2,
                              group["eps"], group["weight_decay"])
                # the kernel wrote through the raw pointer: tell autograd (and the
                # packed-weight cache keyed on ._version) that p changed
                torch.autograd.graph.increment_version(p)
        return loss
