"""Trained-quality parity (north_star: "matched mIoU/PSNR +-0.5"): the CPU
oracle and the HIP path TRAINED side by side -- same initialisation (tcnn
style, seed 123), same frames, same ray indices, same stratified-sampling
(``rng_t``) and inverse-CDF (``rng_u``) tensors at every step, the reference's
loss weights and Adam settings (reference
``nr4seg/lightning/joint_train_lightning_net.py:473-513`` training_step_nerf,
``:897-919`` optimizer) -- for 150 steps of 512 rays x (32+32) samples on the
synthetic room; then four HELD-OUT 64x48 views are rendered by each with ITS
OWN trained parameters and compared with the analytic ground truth:

    |PSNR_hip - PSNR_oracle| <= 0.5 dB,  |mIoU_hip - mIoU_oracle| <= 0.5 pt

for ``train_precision`` fp32, bf16x3 (against the fp32 oracle) and tcnn
(against the oracle emulating tiny-cuda-nn's fp16 roundings: fp16 table and
features, fp16 weights / layer inputs).  Per-step gradient parity is
tests/test_gpu_backward.py / test_gpu_configs.py; this is the trajectory.
``-m gpu``."""
import copy

import numpy as np
import pytest
import torch

from oracle import field as ofield
from oracle import losses as olosses
from oracle import metrics as ometrics
from oracle import renderer as oren
from tests.util import AABB4, hip_network_from_oracle

pytestmark = pytest.mark.gpu

STEPS, N, T, t, C = 150, 512, 32, 32, 40
H, W, VIEWS, HELD = 48, 64, 12, 4   # frames 0..7 train, frames 8..11 held out
LR, WD = 1e-2, 1e-6


@pytest.fixture(scope="module")
def scene():
    """Frames of the synthetic room (GT by analytic ray casting on the GPU),
    and the per-step random tensors, all as CPU tensors."""
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    ds = SyntheticSceneDataset(3, n_views=VIEWS, H=H, W=W, n_classes=C, device="cuda")
    frames = []
    for i in range(VIEWS):
        it = ds[i]
        frames.append(dict(o=it["rays_o"].cpu(), d=it["rays_d"].cpu(),
                           nrm=it["direction_norms"].cpu(),
                           rgb=it["img"].reshape(3, -1).t().contiguous().cpu(),
                           label=it["label"].reshape(-1).cpu(),
                           depth=it["depth"].float().reshape(-1).cpu()))
    g = torch.Generator().manual_seed(2024)
    draws = [dict(frame=int(torch.randint(0, VIEWS - HELD, (1,), generator=g)),
                  inds=torch.randint(0, H * W, (N,), generator=g),
                  rt=torch.rand(N, T, generator=g), ru=torch.rand(N, t, generator=g))
             for _ in range(STEPS)]
    u_eval = torch.rand(HELD * H * W, t, generator=g)
    return frames, draws, u_eval


def _batch(frames, dr):
    f, i = frames[dr["frame"]], dr["inds"]
    return (f["o"][i][None], f["d"][i][None], f["nrm"][i][None], f["rgb"][i][None],
            f["label"][i][None], f["depth"][i][None])


def _held(frames, key):
    return torch.cat([frames[i][key] for i in range(VIEWS - HELD, VIEWS)], 0)


def _quality(img, sem, frames):
    """PSNR (mean over the held-out views of the per-view PSNR, as the
    module's test loop does) and mIoU (one confusion matrix over all of them)."""
    img, gt = img.cpu().view(HELD, H * W, 3), _held(frames, "rgb").view(HELD, H * W, 3)
    psnr = float((-10 * torch.log10(((img - gt) ** 2).mean((1, 2)))).mean())
    _, lab = olosses.semantic_postproc(sem.cpu())
    miou = ometrics.measure(ometrics.confusion(lab.numpy(), _held(frames, "label").numpy(), C))[0]
    return psnr, 100.0 * miou


def _train_oracle(frames, draws, u_eval, emulate_tcnn):
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)
    if emulate_tcnn:
        fld.emulate_fp16 = True
        fld.fp16_table = True
    fld.requires_grad_(True)
    st = [dict(m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in fld.parameters()]
    losses = []
    for k, dr in enumerate(draws):
        o, d, nrm, rgb, lab, dep = _batch(frames, dr)
        out = oren.run(fld, o, d, nrm, AABB4, num_steps=T, upsample_steps=t,
                       t_rand=dr["rt"], u=dr["ru"])
        lc, ls, ld = olosses.nerf_losses(out["image"], out["semantics"], out["depth"],
                                         rgb, lab, dep, 1.0)
        loss = olosses.nerf_total_loss(lc, ls, ld)
        for p in fld.parameters():
            p.grad = None
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            for i, (p, s) in enumerate(zip(fld.parameters(), st)):
                pn, s["m"], s["v"] = olosses.adam_step(
                    p, p.grad, s["m"], s["v"], k + 1, LR, weight_decay=0.0 if i == 0 else WD)
                p.copy_(pn)
    fld.requires_grad_(False)
    with torch.no_grad():
        out = oren.run(fld, _held(frames, "o")[None], _held(frames, "d")[None],
                       _held(frames, "nrm")[None], AABB4, num_steps=T, upsample_steps=t, u=u_eval)
    return _quality(out["image"][0], out["semantics"][0], frames) + (losses,)


def _train_hip(frames, draws, u_eval, precision):
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    net = hip_network_from_oracle(ofield.OracleField(bound=4.0, num_semantic_classes=C,
                                                     seed=123)).train()
    net.train_precision = precision
    opt = HipAdam([{"name": "encoding", "params": list(net.encoder.parameters())},
                   {"name": "net", "params": list(net.sigma_net.parameters()) +
                    list(net.color_net.parameters()) + list(net.semantics_net.parameters()),
                    "weight_decay": WD}], lr=LR, betas=(0.9, 0.99), eps=1e-15)
    # the reference steps the NeRF optimizer through a GradScaler (:46, :509-513);
    # the tcnn arithmetic (fp16 gradients between layers) needs its scale
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    losses = []
    for dr in draws:
        o, d, nrm, rgb, lab, dep = [x.cuda() for x in _batch(frames, dr)]
        out = net.render(o, d, nrm, perturb=True, num_steps=T, upsample_steps=t,
                         rng_t=dr["rt"].cuda(), rng_u=dr["ru"].cuda())
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], rgb, lab,
                                    dep, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.detach())
    net.eval()
    # inference arithmetic of the same family as the training one
    net.precision = {"fp32": "fp32", "bf16x3": "bf16x3", "tcnn": "fp16"}[precision]
    net.fp16_table = precision == "tcnn"
    with torch.no_grad():
        out = net.render(_held(frames, "o")[None].cuda(), _held(frames, "d")[None].cuda(),
                         _held(frames, "nrm")[None].cuda(), staged=True, num_steps=T,
                         upsample_steps=t, rng_u=u_eval.cuda())
    return _quality(out["image"][0], out["semantics"][0], frames) + \
        ([float(x) for x in torch.stack(losses).cpu()],)


@pytest.fixture(scope="module")
def oracle_fp32(scene):
    return _train_oracle(*scene, emulate_tcnn=False)


@pytest.fixture(scope="module")
def oracle_tcnn(scene):
    return _train_oracle(*scene, emulate_tcnn=True)


def _compare(tag, hip, ora):
    (ph, mh, lh), (po, mo, lo) = hip, ora
    lh, lo = np.array(lh), np.array(lo)
    print(f"{tag}: held-out PSNR hip {ph:.3f} dB / oracle {po:.3f} dB (d {ph - po:+.3f}); "
          f"mIoU hip {mh:.2f} / oracle {mo:.2f} pt (d {mh - mo:+.2f}); "
          f"loss step 1 {lh[0]:.5f} / {lo[0]:.5f}, mean of last 10 {lh[-10:].mean():.5f} / "
          f"{lo[-10:].mean():.5f}; max |loss difference| first 20 steps "
          f"{np.abs(lh[:20] - lo[:20]).max():.2e}")
    assert lo[-10:].mean() < 0.6 * lo[0] and lh[-10:].mean() < 0.6 * lh[0]   # both learned
    assert po > 14.0                                      # the run means something
    assert abs(ph - po) <= 0.5, (tag, ph, po)
    assert abs(mh - mo) <= 0.5, (tag, mh, mo)
    return lh, lo


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_trajectory_quality_matches_the_fp32_oracle(scene, oracle_fp32, precision):
    lh, lo = _compare(precision, _train_hip(*scene, precision), oracle_fp32)
    # before round-off has had time to grow the two runs are the same run
    assert np.abs(lh[:5] - lo[:5]).max() <= 2e-5 * max(1.0, lo[0])


def test_trajectory_quality_tcnn_numerics_matches_the_fp16_emulating_oracle(scene, oracle_tcnn):
    _compare("tcnn", _train_hip(*scene, "tcnn"), oracle_tcnn)
