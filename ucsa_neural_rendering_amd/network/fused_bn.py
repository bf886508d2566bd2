"""BatchNorm2d (+ residual add) (+ ReLU) as ONE fused HIP op on channels-last
activations (``ucsa_bn_act_fwd`` / ``ucsa_bn_act_bwd``, csrc/batchnorm.hip):
the memory-bound passes between DeepLabV3's convolutions (SURVEY 8a row a14;
reference ``nr4seg/network/deeplabv3.py:6-19`` -> torchvision's bottlenecks).

``FusedBatchNorm2d`` IS an ``nn.BatchNorm2d`` (same parameters, buffers and
state_dict keys -- reference checkpoints load unchanged); its ``forward`` takes
two optional extras, ``residual`` and ``relu``.  The fused kernels run when the
input is a channels_last CUDA tensor in fp32 or bf16 AND the parameters /
running statistics are contiguous fp32 on the same device (autocast keeps
them so; ``model.bfloat16()`` does not) AND C <= 4096; any other case (CPU,
NCHW, 1x1 maps whose strides are ambiguous, a half-precision module) goes through
``F.batch_norm`` + add + relu, the reference's own sequence, so the module is
usable everywhere and the two paths can be compared against each other
(tests/test_gpu_fused_bn.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class _BnActFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var,
                momentum, eps, relu, training):
        y, mean, invstd = ops.bn_act_fwd(x, residual, weight, bias, running_mean,
                                         running_var, momentum, eps, relu, training)
        ctx.relu = relu
        ctx.training = training
        ctx.has_res = residual is not None
        ctx.eps = eps
        if training:
            ctx.save_for_backward(x, y if relu else None, weight, mean, invstd)
        else:
            ctx.save_for_backward(x, y if relu else None, weight, running_mean,
                                  running_var)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, a, b = ctx.saved_tensors
        if ctx.training:
            mean, invstd = a, b
        else:
            # eval mode: the statistics are constants -- dx = gamma invstd g.
            # Same kernels with dgamma / dbeta terms removed is not worth a third
            # code path: plain torch (eval-mode backward is not on the hot path)
            g = dy
            if ctx.relu:
                g = dy * (y > 0).to(dy.dtype)
            invstd = torch.rsqrt(b + ctx.eps)
            w = weight if weight is not None else torch.ones_like(invstd)
            dx = g * (w * invstd).to(g.dtype).view(1, -1, 1, 1)
            xh = (x.float() - a.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
            dw = (g.float() * xh).sum((0, 2, 3)) if ctx.needs_input_grad[2] else None
            db = g.float().sum((0, 2, 3)) if ctx.needs_input_grad[3] else None
            return (dx, g if ctx.has_res else None, dw, db, None, None, None, None,
                    None, None)
        want_wb = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dx, dres, dw, db = ops.bn_act_bwd(dy, x, y, weight, mean, invstd, ctx.relu,
                                          ctx.has_res and ctx.needs_input_grad[1],
                                          want_wb)
        return (dx, dres, dw if ctx.needs_input_grad[2] else None,
                db if ctx.needs_input_grad[3] else None, None, None, None, None,
                None, None)


def _fusable(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
            and x.shape[1] % 4 == 0 and x.shape[2] * x.shape[3] > 1
            and x.is_contiguous(memory_format=torch.channels_last)
            and not x.is_contiguous())


class FusedBatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose forward can also add a residual and apply the
    ReLU: ``bn(x, residual=identity, relu=True)``."""

    def forward(self, x, residual=None, relu: bool = False):
        training = self.training or (self.running_mean is None)
        C = x.shape[1] if x.dim() == 4 else 0
        # the kernels read the per-channel operands as float*: a module cast
        # with .half() / .bfloat16() (2-byte parameters), parameters on another
        # device, or C beyond the kernels' tables take the F.batch_norm path
        vecs_ok = C <= ops.BN_MAX_CHANNELS and all(
            ops.bn_vec_ok(v, C, x.device)
            for v in (self.weight, self.bias, self.running_mean, self.running_var))
        if vecs_ok and _fusable(x) and (
                residual is None or
                (residual.shape == x.shape and residual.dtype == x.dtype
                 and residual.is_contiguous(memory_format=torch.channels_last))):
            momentum = self.momentum
            if self.training and self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
                if momentum is None:   # cumulative moving average
                    momentum = 1.0 / float(self.num_batches_tracked)
            rm = self.running_mean if (not self.training or self.track_running_stats) else None
            rv = self.running_var if (not self.training or self.track_running_stats) else None
            return _BnActFn.apply(x, residual, self.weight, self.bias, rm, rv,
                                  0.0 if momentum is None else float(momentum),
                                  float(self.eps), bool(relu), bool(training))
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
