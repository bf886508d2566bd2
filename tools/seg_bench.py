"""DeepLabV3 fwd+bwd+Adam timing (cfg3's segmentation half): B=8, 3x240x320."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucsa_neural_rendering_amd.network import DeepLabV3
from ucsa_neural_rendering_amd import losses as ul
dev = torch.device("cuda", 0)
B = 8
# SEG_BENCHMARK=1: torch.backends.cudnn.benchmark (MIOpen's exhaustive find)
torch.backends.cudnn.benchmark = os.environ.get("SEG_BENCHMARK", "0") == "1"
print("cudnn.benchmark =", torch.backends.cudnn.benchmark, " MIOPEN_FIND_MODE =",
      os.environ.get("MIOPEN_FIND_MODE"), flush=True)
for backbone in ("resnet101", "resnet50"):
    for mode in ("fp32_nchw", "fp32_cl", "bf16_cl"):
        torch.manual_seed(0)
        m = DeepLabV3({"pretrained": False, "pretrained_backbone": False, "num_classes": 40, "backbone": backbone}).to(dev).train()
        cl = mode.endswith("cl")
        if cl: m = m.to(memory_format=torch.channels_last)
        opt = torch.optim.Adam(m.parameters(), lr=1e-5)
        x = torch.rand(B, 3, 240, 320, device=dev)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        y = torch.randint(-1, 40, (B, 240, 320), device=dev)
        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode.startswith("bf16")):
                logits = m(x)["out"]
            loss = ul.seg_loss(logits.float().contiguous(), y)
            opt.zero_grad(); loss.backward(); opt.step()
            return loss
        try:
            for _ in range(4): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): l = step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            print(f"{backbone} {mode}: {dt*1e3:8.1f} ms/step  {B/dt:7.1f} img/s  loss {float(l.detach()):.4f}", flush=True)
        except Exception as e:
            print(backbone, mode, "FAILED", repr(e)[:200], flush=True)
        del m, opt
        torch.cuda.empty_cache()
