"""Mirror of reference ``nr4seg/nerf/activation.py``: ``trunc_exp``.

Inside the HIP field the exponential and its clamped backward are fused into
the sigma-MLP kernels; this autograd Function exists for API parity and for
callers that apply it to their own tensors."""
import torch
from torch.autograd import Function


class _trunc_exp(Function):

    @staticmethod
    def forward(ctx, x):
        x = x.float()  # reference: custom_fwd(cast_inputs=torch.float)
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply
