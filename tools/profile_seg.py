"""DeepLabV3-R101 fwd+bwd+Adam steps for a rocprofv3 kernel trace
(MODE=fp32_nchw|fp32_cl|bf16_cl, B, WARM, STEPS in the environment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucsa_neural_rendering_amd.network import DeepLabV3
from ucsa_neural_rendering_amd import losses as ul
dev = torch.device("cuda", 0)
mode = os.environ.get("MODE", "fp32_nchw")
B = int(os.environ.get("B", "8"))
torch.manual_seed(0)
torch.backends.cudnn.benchmark = os.environ.get("FIND", "0") == "1"
m = DeepLabV3({"pretrained": False, "pretrained_backbone": False, "num_classes": 40,
               "backbone": os.environ.get("BACKBONE", "resnet101")}).to(dev).train()
cl = mode.endswith("cl")
if cl:
    m = m.to(memory_format=torch.channels_last)
opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=cl)
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


for _ in range(int(os.environ.get("WARM", "3"))):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("STEPS", "10"))
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"{mode} B={B} find={torch.backends.cudnn.benchmark}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms/step", flush=True)
