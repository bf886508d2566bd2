"""DeepLabV3 training-step leg (cfg3's segmentation half)."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403


def conv_flops(model, x):
    """Forward FLOP of the convolutions (2 x MACs) and linear layers of
    `model` on input `x`, counted with forward hooks on the build's own
    modules ('FLOPs from a counter on the build's own model')."""
    total = [0]
    hooks = []

    def conv_hook(m, inp, out):
        kh, kw = m.kernel_size
        total[0] += 2 * out.numel() * (m.in_channels // m.groups) * kh * kw

    def lin_hook(m, inp, out):
        total[0] += 2 * out.numel() * m.in_features

    for m in model.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, torch.nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    return total[0]


def seg_throughput(device, steps=5, B=8, find=False):
    """cfg3's segmentation half: DeepLabV3-ResNet-101 forward + backward +
    Adam on [8,3,240,320] uniform-random images / labels, with the
    reference's CE-on-softmax loss through ucsa_seg_tail.

    * ``fp32``: the module's default path -- fp32 like the reference (no
      autocast around seg), channels-last, every BatchNorm (+ add) (+ ReLU) one
      fused HIP op (csrc/batchnorm.hip), 1x1 convolutions as one GEMM over the
      batch; 3x3 / 7x7 convolutions are MIOpen.
    * ``fp32_nchw_unfused``: the same modules on NCHW inputs, i.e.
      F.batch_norm + add + relu kernels and MIOpen's per-image 1x1 GEMMs (what
      rounds 1-2 measured as "fp32").
    * ``bf16_channels_last``: bf16 autocast, fused BatchNorm in bf16.
    * ``*_graph``: forward and backward replayed as HIP graphs
      (torch.cuda.make_graphed_callables); the optimizer step stays eager."""
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.network import DeepLabV3
    out = {}
    # MIOpen exhaustive find, as scripts/train_joint.py sets it: minutes of
    # search on a fresh box, so the default bench run measures immediate mode
    before = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(find)
    out["miopen_find"] = bool(find)
    for mode in ("fp32", "fp32_nchw_unfused", "bf16_channels_last", "fp32_graph",
                 "bf16_graph"):
        torch.manual_seed(0)
        amp = mode.startswith("bf16")
        nchw = mode == "fp32_nchw_unfused"
        m = DeepLabV3({"pretrained": False, "pretrained_backbone": False,
                       "num_classes": N_CLASSES}).to(device).train()
        x = torch.rand(B, 3, 240, 320, device=device)
        if not nchw:
            m = m.to(memory_format=torch.channels_last)
            x = x.contiguous(memory_format=torch.channels_last)
        y = torch.randint(-1, N_CLASSES, (B, 240, 320), device=device)
        opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=not nchw)

        class _Net(torch.nn.Module):   # parameters visible to make_graphed_callables
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, inp):
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                    return self.inner(inp)["out"]

        net = _Net(m)
        fwd = net
        try:
            if mode.endswith("_graph"):
                fwd = torch.cuda.make_graphed_callables(net, (x.clone(),))
        except Exception as e:  # report, do not hide
            out[mode] = {"failed": repr(e)[:300]}
            del m, opt
            torch.cuda.empty_cache()
            continue

        def one():
            logits = fwd(x)
            loss = ul.seg_loss(logits.float().contiguous(), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss

        for _ in range(2):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if "fwd_flop_per_image" not in out:
            m.eval()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out["fwd_flop_per_image"] = conv_flops(m, x[:1])
            m.train()
        flop = 3.0 * B * out["fwd_flop_per_image"]
        peak = F16_MFMA_PEAK_TF if amp else F32_MFMA_PEAK_TF
        out[mode] = {"ms_per_step": dt * 1e3, "images_per_s": B / dt,
                     "loss": float(loss),
                     "roofline": {"bound": "mfma", "algorithmic_flop": flop,
                                  "achieved": flop / dt / 1e12, "peak": peak,
                                  "unit": "TFLOP/s", "frac": flop / dt / 1e12 / peak,
                                  "note": "3 x the forward convolution flop of the "
                                          "mirror (hook counter, conv_flops) x 8 images "
                                          "/ step time; peak = " +
                                          ("bf16 dense MFMA" if amp else
                                           "fp32-input MFMA (= fp32 vector) rate")}}
        # HBM traffic / MFMA-busy per step from the committed PMC passes
        # (tools/seg_pmc.sh; rocprofv3 cannot run inside this process)
        key = {"fp32": "fp32_cl", "bf16_channels_last": "bf16_cl"}.get(mode)
        if key:
            try:
                pj = json.load(open(os.path.join(ROOT, SEG_PMC_JSON))).get(key)
            except (OSError, ValueError):
                pj = None
            if not pj:
                out[mode]["roofline"]["traffic"] = None
                out[mode]["roofline"]["traffic_source"] = (
                    None, f"{SEG_PMC_JSON} absent: rocprofv3 --pmc of the DeepLab step "
                          "aborts on this pool (profiles/README.md)")
            if pj:
                r = out[mode]["roofline"]
                r["traffic"] = pj["hbm_bytes_per_step"]
                r["traffic_source"] = SEG_PMC_JSON
                r["hbm_utilisation"] = pj["hbm_bytes_per_step"] / dt / 1e9 / HBM_PEAK_GBS
                r["mfma_pipe_busy_frac"] = pj["mfma_busy_frac"]
                r["valu_issue_frac"] = pj["valu_issue_frac"]
        del m, opt, fwd
        torch.cuda.empty_cache()
    out["workload"] = ("DeepLabV3-ResNet-101 train step, batch 8 x 3x240x320, "
                       "CE-on-softmax loss, Adam")
    torch.backends.cudnn.benchmark = before
    return out
