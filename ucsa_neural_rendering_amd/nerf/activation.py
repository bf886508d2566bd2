"""``trunc_exp`` of reference ``nr4seg/nerf/activation.py:7-21``: the density
activation exp(x) whose derivative is evaluated at clamp(x, -15, 15).

Inside the HIP field both halves are fused into the sigma-MLP kernels (forward
``sigma = exp(h0)``, backward in ``k_weights_bwd`` / ``k_march_weights_bwd``);
this autograd Function is the API-parity entry for callers that apply it to
their own tensors."""
import torch

_LIMIT = 15.0


class TruncExp(torch.autograd.Function):
    """y = exp(float32(x)); dL/dx = dL/dy * exp(clamp(x, -15, 15))."""

    @staticmethod
    def forward(ctx, inp):
        as_f32 = inp.to(torch.float32)  # the reference casts with custom_fwd
        ctx.save_for_backward(as_f32)
        return as_f32.exp()

    @staticmethod
    def backward(ctx, grad_out):
        (as_f32,) = ctx.saved_tensors
        return grad_out * torch.clamp(as_f32, min=-_LIMIT, max=_LIMIT).exp()


def trunc_exp(x):
    return TruncExp.apply(x)
