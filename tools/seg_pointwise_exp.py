"""Experiment: 1x1 convolutions of DeepLabV3-R101 as one GEMM over the whole
batch (F.linear on the NHWC view / batched matmul on NCHW) instead of MIOpen's
per-image GEMM launches.  B=8, 3x240x320, fwd+bwd+Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from ucsa_neural_rendering_amd.network import DeepLabV3
from ucsa_neural_rendering_amd import losses as ul
dev = torch.device("cuda", 0)
B = 8
_orig = nn.Conv2d.forward
PW = {"mode": "off"}


def pw_forward(self, x):
    if PW["mode"] == "off" or self.kernel_size != (1, 1) or self.groups != 1 or x.shape[-1] * x.shape[-2] == 1:
        return _orig(self, x)
    if self.stride != (1, 1):
        x = x[:, :, ::self.stride[0], ::self.stride[1]]
    w = self.weight.view(self.out_channels, self.in_channels)
    if PW["mode"] == "nhwc":
        y = F.linear(x.permute(0, 2, 3, 1), w, self.bias)   # view when channels_last
        return y.permute(0, 3, 1, 2)
    Bn, C, H, W = x.shape
    y = torch.matmul(w, x.reshape(Bn, C, H * W))
    if self.bias is not None:
        y = y + self.bias.view(1, -1, 1)
    return y.view(Bn, -1, H, W)


nn.Conv2d.forward = pw_forward
for mode, pw in (("fp32_nchw", "off"), ("fp32_nchw", "bmm"), ("fp32_cl", "off"), ("fp32_cl", "nhwc"),
                 ("bf16_cl", "off"), ("bf16_cl", "nhwc")):
    PW["mode"] = pw
    torch.manual_seed(0)
    m = DeepLabV3({"pretrained": False, "pretrained_backbone": False, "num_classes": 40}).to(dev).train()
    cl = mode.endswith("cl")
    if cl:
        m = m.to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-5)
    x = torch.rand(B, 3, 240, 320, device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(-1, 40, (B, 240, 320), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode.startswith("bf16")):
            logits = m(x)["out"]
        loss = ul.seg_loss(logits.float().contiguous(), y)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    try:
        for _ in range(3):
            l0 = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            l = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{mode} pointwise={pw}: {dt*1e3:8.1f} ms/step  loss0 {float(l0):.5f} loss {float(l.detach()):.5f}", flush=True)
    except Exception as e:
        print(mode, pw, "FAILED", repr(e)[:300], flush=True)
    del m, opt
    torch.cuda.empty_cache()
