"""Row a14 at BASELINE cfg3's size: the DeepLabV3 mirror
(``network/deeplabv3.py``; reference ``nr4seg/network/deeplabv3.py:6-19``,
called at ``joint_train_lightning_net.py:159-165`` and ``:456-461``) on the
MI355X against the SAME module run in fp32 on the host CPU -- logits,
CE-on-softmax loss (``ucsa_seg_tail`` vs the torch modules the reference
instantiates) and every parameter gradient, for ``[8, 3, 240, 320]`` inputs
and 40 classes, ResNet-50 (cfg3's wording) and ResNet-101 (the reference's
model).

Layouts / precisions covered: fp32 NCHW (the reference's), fp32
channels_last (``PointwiseConv2d`` = one GEMM over the NHWC view), bf16
autocast + channels_last (the optional fast path; its own, looser tolerance).

Stated tolerances (fp32): logits <= 1e-3 absolute, loss <= 1e-5 relative,
parameter gradients <= 1e-3 relative L2 over all parameters.  bf16: logits
<= 0.25 absolute / 3e-2 relative L2, gradients <= 0.15 relative L2.

Dropout(0.5) in the ASPP projection draws from the device RNG, which cannot
be replayed across devices: the train-mode pass runs with BatchNorm in batch-
statistics mode and Dropout disabled (p -> identity) on both sides.  That
pass runs with BatchNorm momentum 1, so the running statistics it leaves are
the batch's (compared too), and the eval-mode forward that follows (the
pseudo-label pass, reference :374-381) runs on realistic statistics.

If torchvision is importable on the box, the same state_dict is loaded
(strict) into ``torchvision.models.segmentation.deeplabv3_resnet*`` and the
logits compared -- that pins the mirror to the third-party model itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B, C, H, W = 8, 40, 240, 320


def _model(backbone, seed=0):
    from ucsa_neural_rendering_amd.network import DeepLabV3
    torch.manual_seed(seed)
    m = DeepLabV3({"pretrained": False, "pretrained_backbone": False,
                   "num_classes": C, "backbone": backbone})
    # non-trivial BatchNorm affine parameters; momentum 1 so that ONE
    # train-mode pass leaves the batch statistics in the running buffers (a
    # random-init ResNet evaluated with the default 0/1 statistics doubles its
    # variance at every residual block -- not a meaningful eval-mode input)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.momentum = 1.0
                mod.weight.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
                mod.bias.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    return m


def _inputs(seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, H, W, generator=g)            # images in [0,1], no mean/std
    y = torch.randint(-1, C, (B, H, W), generator=g)   # -1 = ignored
    return x, y


def _train_mode(m):
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()
    return m


def _cpu_reference(backbone):
    """fp32 CPU: eval logits; train-mode logits, loss, gradients."""
    m = _model(backbone)
    x, y = _inputs()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _train_mode(m)
    out = m(x)["out"]
    # the reference's loss: CE(ignore -1, reduction none) on softmax(out), mean
    loss = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction="none")(
        F.softmax(out, dim=1), y).mean()
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k}
    m.eval()   # running statistics = the batch's (momentum 1)
    with torch.no_grad():
        ev = m(x)["out"]
    return dict(sd=sd, eval_logits=ev, logits=out.detach(), loss=float(loss),
                grads=grads, stats=stats)


_REF = {}


def _ref(backbone):
    if backbone not in _REF:
        _REF[backbone] = _cpu_reference(backbone)
    return _REF[backbone]


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _gpu_run(backbone, mode):
    from ucsa_neural_rendering_amd import losses as ul
    ref = _ref(backbone)
    m = _model(backbone)
    m.load_state_dict(ref["sd"], strict=True)
    m = m.cuda()
    x, y = _inputs()
    x, y = x.cuda(), y.cuda()
    amp = mode == "bf16"
    if mode != "fp32_nchw":
        m = m.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    _train_mode(m)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out = m(x)["out"]
    loss = ul.seg_loss(out.float().contiguous(), y)      # ucsa_seg_tail fwd+bwd
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad for k, p in m.named_parameters()}
    stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k}
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        ev = m(x)["out"].float()
    return dict(eval_logits=ev, logits=out.detach().float(), loss=float(loss),
                grads=grads, stats=stats)


@pytest.mark.parametrize("backbone", ["resnet50", "resnet101"])
@pytest.mark.parametrize("mode", ["fp32_nchw", "channels_last"])
def test_deeplab_fp32_forward_backward_matches_cpu(backbone, mode):
    ref = _ref(backbone)
    got = _gpu_run(backbone, mode)
    scale = max(1.0, float(ref["logits"].abs().max()))
    print(f"[{backbone} {mode}] max|logits| {scale:.3f}  train dlogits "
          f"{float((got['logits'].cpu() - ref['logits']).abs().max()):.3e}  eval dlogits "
          f"{float((got['eval_logits'].cpu() - ref['eval_logits']).abs().max()):.3e}  "
          f"loss {got['loss']:.7f} vs {ref['loss']:.7f}")
    assert float((got["logits"].cpu() - ref["logits"]).abs().max()) <= 1e-3 * scale
    # BatchNorm running statistics after the pass (momentum 1: the batch's)
    for k, v in ref["stats"].items():
        assert float((got["stats"][k].cpu() - v).abs().max()) <= 1e-4 * max(
            1.0, float(v.abs().max())), k
    scale = max(1.0, float(ref["eval_logits"].abs().max()))
    assert float((got["eval_logits"].cpu() - ref["eval_logits"]).abs().max()) <= 1e-3 * scale
    assert abs(got["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    flat_g = torch.cat([got["grads"][k].reshape(-1).cpu() for k in ref["grads"]])
    flat_r = torch.cat([ref["grads"][k].reshape(-1) for k in ref["grads"]])
    print(f"[{backbone} {mode}] grad rel L2 {_rel_l2(flat_g, flat_r):.3e}")
    assert _rel_l2(flat_g, flat_r) <= 1e-3
    # no parameter is left without a gradient, none is wildly off on its own
    worst = max((_rel_l2(got["grads"][k], ref["grads"][k]), k)
                for k in ref["grads"] if float(ref["grads"][k].norm()) > 1e-6 * float(flat_r.norm()))
    print(f"[{backbone} {mode}] worst single parameter {worst}")
    assert worst[0] <= 2e-2, worst


@pytest.mark.parametrize("backbone", ["resnet50", "resnet101"])
def test_deeplab_bf16_channels_last_within_its_tolerance(backbone):
    """`model: {amp: bf16}` (optional fast path, never the parity path)."""
    ref = _ref(backbone)
    got = _gpu_run(backbone, "bf16")
    flat_g = torch.cat([got["grads"][k].reshape(-1).float().cpu() for k in ref["grads"]])
    flat_r = torch.cat([ref["grads"][k].reshape(-1) for k in ref["grads"]])
    print(f"[{backbone} bf16] logits rel L2 {_rel_l2(got['logits'], ref['logits']):.3e} max "
          f"{float((got['logits'].cpu() - ref['logits']).abs().max()):.3e}  eval rel L2 "
          f"{_rel_l2(got['eval_logits'], ref['eval_logits']):.3e}  loss {got['loss']:.6f} vs "
          f"{ref['loss']:.6f}  grad rel L2 {_rel_l2(flat_g, flat_r):.3e}")
    assert _rel_l2(got["logits"], ref["logits"]) <= 3e-2
    assert float((got["logits"].cpu() - ref["logits"]).abs().max()) <= 0.25 * max(
        1.0, float(ref["logits"].abs().max()))
    assert _rel_l2(got["eval_logits"], ref["eval_logits"]) <= 3e-2
    assert abs(got["loss"] - ref["loss"]) <= 2e-3 * abs(ref["loss"])
    flat_g = torch.cat([got["grads"][k].reshape(-1).float().cpu() for k in ref["grads"]])
    flat_r = torch.cat([ref["grads"][k].reshape(-1) for k in ref["grads"]])
    assert _rel_l2(flat_g, flat_r) <= 0.15


@pytest.mark.parametrize("backbone", ["resnet50", "resnet101"])
def test_deeplab_matches_torchvision_when_present(backbone):
    tv = pytest.importorskip("torchvision")
    ctor = getattr(tv.models.segmentation, f"deeplabv3_{backbone}")
    try:
        t = ctor(weights=None, weights_backbone=None, num_classes=C, aux_loss=None)
    except TypeError:  # torchvision 0.12 signature (the reference's pin)
        t = ctor(pretrained=False, pretrained_backbone=False, num_classes=C, aux_loss=None)
    ref = _ref(backbone)
    sd = {k[len("_model."):]: v for k, v in ref["sd"].items()}
    t.load_state_dict(sd, strict=True)
    x, _ = _inputs()
    t.load_state_dict({k[len("_model."):]: v for k, v in ref["stats"].items()},
                      strict=False)
    t.eval()
    with torch.no_grad():
        want = t(x)["out"]
        got = t.cuda()(x.cuda())["out"]
    scale = max(1.0, float(ref["eval_logits"].abs().max()))
    assert float((want - ref["eval_logits"]).abs().max()) <= 1e-4 * scale
    assert float((got.cpu() - ref["eval_logits"]).abs().max()) <= 1e-3 * scale
