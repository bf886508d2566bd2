"""Second stage of the ASPP pooling-branch diagnosis (see aspp_probe.py): the
crash is train-mode BatchNorm2d on the [2,256,1,1] bf16 output of the 1x1
convolution.  Which property triggers it: strides, dtype, batch size, MIOpen
vs the native kernel?  Each case in its own subprocess."""
import subprocess
import sys

PRE = """
import torch, torch.nn.functional as F
torch.manual_seed(0)
bn = torch.nn.BatchNorm2d(256).cuda().train()
conv = torch.nn.Conv2d(2048, 256, 1, bias=False).cuda()
def mk(B, dt, strides=None, C=256):
    t = torch.randn(B, C, 1, 1, device='cuda').to(dt)
    return t if strides is None else torch.as_strided(t.clone(), (B, C, 1, 1), strides)
"""
CASES = {
    "bf16 B=2 strides (C,1,1,1) [plain contiguous]": "y = bn(mk(2, torch.bfloat16))",
    "bf16 B=2 strides (C,1,C,C) [NHWC-style]": "y = bn(mk(2, torch.bfloat16, (256,1,256,256)))",
    "bf16 B=3 strides (C,1,C,C)": "y = bn(mk(3, torch.bfloat16, (256,1,256,256)))",
    "bf16 B=4 strides (C,1,C,C)": "y = bn(mk(4, torch.bfloat16, (256,1,256,256)))",
    "fp16 B=2 strides (C,1,C,C)": "y = bn(mk(2, torch.float16, (256,1,256,256)))",
    "fp32 B=2 strides (C,1,C,C)": "y = bn(mk(2, torch.float32, (256,1,256,256)))",
    "bf16 B=2 conv(NHWC in) out -> strides printed": "x = mk(2, torch.bfloat16, C=2048).contiguous(memory_format=torch.channels_last); c = conv.to(torch.bfloat16).to(memory_format=torch.channels_last)(x); print('conv out strides', c.stride(), c.is_contiguous(), c.is_contiguous(memory_format=torch.channels_last)); y = bn(c)",
    "bf16 B=2 conv out .reshape(B,C,1,1).clone()": "x = mk(2, torch.bfloat16, C=2048).contiguous(memory_format=torch.channels_last); c = conv.to(torch.bfloat16).to(memory_format=torch.channels_last)(x); y = bn(c.reshape(2,256).clone().view(2,256,1,1))",
    "bf16 B=2 NHWC strides, MIOpen off (cudnn.enabled=False)": "torch.backends.cudnn.enabled = False; y = bn(mk(2, torch.bfloat16, (256,1,256,256)))",
    "bf16 B=2 NHWC strides, 2x2 map": "t = torch.randn(2,256,2,2,device='cuda').bfloat16().contiguous(memory_format=torch.channels_last); y = bn(t)",
    "bf16 B=2 NHWC strides, 1x2 map": "t = torch.randn(2,256,1,2,device='cuda').bfloat16().contiguous(memory_format=torch.channels_last); y = bn(t)",
    "bf16 B=2 NHWC strides, C=64": "bn = torch.nn.BatchNorm2d(64).cuda().train(); y = bn(mk(2, torch.bfloat16, (64,1,64,64), C=64))",
    "bf16 B=2 NHWC strides, bwd": "t = mk(2, torch.bfloat16).requires_grad_(); y = bn(t); y.float().sum().backward()",
}


def main():
    for name, body in CASES.items():
        src = PRE + body + "\ntorch.cuda.synchronize()\nprint('ok', tuple(y.shape), y.dtype, y.stride(), bool(torch.isfinite(y.float()).all()))\n"
        r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=300)
        out = " | ".join(r.stdout.strip().splitlines()[-2:])
        err = (r.stderr.strip().splitlines() or [""])[-1][:160]
        print(f"{name:62s} rc={r.returncode} {out} {err if r.returncode else ''}", flush=True)


if __name__ == "__main__":
    main()
