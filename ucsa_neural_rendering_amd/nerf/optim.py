"""Adam for the NeRF parameter groups, backed by the HIP kernel
``ucsa_adam_step``.

Same constructor contract as ``torch.optim.Adam`` as the reference calls it
(nr4seg/lightning/joint_train_lightning_net.py:897-919): named param groups,
``lr``, ``betas``, ``eps``, per-group ``weight_decay`` (L2 folded into the
gradient, not AdamW).  Under ``torch.amp.GradScaler`` (the reference steps
its NeRF optimizer through one, :46,:509-513) it declares
``_step_supports_amp_scaling``: the scaler then hands over its scale and its
found-inf flag as device tensors instead of reading the flag back, and the
kernel (``ucsa_adam_step_scaled``) unscales / skips on the device -- no host
synchronisation per training step."""
from __future__ import annotations

import torch

from .. import ops


class HipAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._step_supports_amp_scaling = True
        self._skipped = {}  # device -> int32[1]: steps skipped by the scaler

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # set by GradScaler.step() around this call (device tensors)
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        scaled = found_inf is not None  # grad_scale None: already unscaled
        devices = set()
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
                skipped = self._skipped.get(p.device)
                if scaled or skipped is not None:
                    if skipped is None:
                        skipped = torch.zeros(1, dtype=torch.int32, device=p.device)
                        self._skipped[p.device] = skipped
                    if scaled:
                        gs = (torch.ones(1, device=p.device) if grad_scale is None
                              else grad_scale.to(p.device, torch.float32).reshape(1))
                        fi = found_inf.to(p.device, torch.float32).reshape(1)
                    else:  # plain step after scaled ones: keep the step count
                        gs = torch.ones(1, device=p.device)
                        fi = torch.zeros(1, device=p.device)
                    ops.adam_step_scaled(
                        p.data, p.grad.contiguous(), st["exp_avg"],
                        st["exp_avg_sq"], st["step"], group["lr"], b1, b2,
                        group["eps"], group["weight_decay"], gs, fi, skipped)
                    if scaled:
                        devices.add((p.device, fi))
                else:
                    ops.adam_step(p.data, p.grad.contiguous(), st["exp_avg"],
                                  st["exp_avg_sq"], st["step"], group["lr"], b1,
                                  b2, group["eps"], group["weight_decay"])
                # the kernel wrote through the raw pointer: tell autograd (and the
                # packed-weight cache keyed on ._version) that p changed
                torch.autograd.graph.increment_version(p)
        for dev, fi in {d: f for d, f in devices}.items():
            ops.adam_count_skipped(fi, self._skipped[dev])
        return loss
