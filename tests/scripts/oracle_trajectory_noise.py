"""How far apart are two trainings of the SAME implementation?  The oracle of
tests/test_gpu_trajectory.py (150 steps x 2048 rays x (16+16), the reference's Adam:
lr 1e-2, eps 1e-15) trained three times on the CPU, NO GPU:
  a  as the test trains it (8 threads)
  b  with 3 threads (another partition of the BLAS sums = another round-off)
  c  with 8 threads and every density multiplied by 1 + 1e-6 * randn
and the test's quality figures (mean over its 9 checkpoints of the train-view /
held-out PSNR and mIoU) side by side: the spread between a, b and c is what no
HIP-vs-oracle bound can go below.   python tests/scripts/oracle_trajectory_noise.py
(frames by SyntheticRoom.cast on oracle.rays pixels: the test's scene, CPU-made)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import field as ofield, losses as olosses, metrics as ometrics, renderer as oren
from oracle.rays import pixel_rays
from tests.util import AABB4
from ucsa_neural_rendering_amd.dataset.synthetic_scene import SyntheticRoom, _slerp_loop_poses

STEPS, N, T, t, C = 150, 2048, 16, 16, 40
CHECKPOINTS = (69, 79, 89, 99, 109, 119, 129, 139, 149)
H, W, VIEWS, HELD = 48, 64, 12, 4
LR, WD = 1e-2, 1e-6


def scene():
    room = SyntheticRoom(3, n_classes=C)
    poses = _slerp_loop_poses(VIEWS, seed=123 + 3)
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    frames = []
    for i in range(VIEWS):
        o, d, n = pixel_rays(poses[i:i + 1], intr, H, W)
        t_hit, rgb, label = room.cast(o[0], d[0])
        frames.append(dict(o=o[0], d=d[0], nrm=n[0].reshape(-1), rgb=rgb, label=label,
                           depth=(t_hit / n[0].reshape(-1)).half().float()))
    g = torch.Generator().manual_seed(2024)
    draws = [dict(frame=int(torch.randint(0, VIEWS - HELD, (1,), generator=g)),
                  inds=torch.randint(0, H * W, (N,), generator=g),
                  rt=torch.rand(N, T, generator=g), ru=torch.rand(N, t, generator=g))
             for _ in range(STEPS)]
    u_eval = torch.rand(VIEWS * H * W, t, generator=g)
    return frames, draws, u_eval


def cat(frames, key):
    return torch.cat([f[key] for f in frames], 0)


def quality(img, sem, frames):
    img, gt = img.view(VIEWS, H * W, 3), cat(frames, "rgb").view(VIEWS, H * W, 3)
    per_view = -10 * torch.log10(((img - gt) ** 2).mean((1, 2)))
    _, lab = olosses.semantic_postproc(sem)
    lab, gt_lab = lab.view(VIEWS, -1).numpy(), cat(frames, "label").view(VIEWS, -1).numpy()
    k = VIEWS - HELD
    miou = lambda a, b: 100.0 * ometrics.measure(ometrics.confusion(a, b, C))[0]   # noqa: E731
    return (float(per_view[:k].mean()), miou(lab[:k], gt_lab[:k]), float(per_view[k:].mean()), miou(lab[k:], gt_lab[k:]))


def train(frames, draws, u_eval, threads, noise):
    torch.set_num_threads(threads)
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=C, seed=123)
    fld.requires_grad_(True)
    if noise:
        dens = fld.density
        gen = torch.Generator().manual_seed(5)

        def noisy(x, _d=dens):
            out = dict(_d(x))
            out["sigma"] = out["sigma"] * (1.0 + noise * torch.randn(out["sigma"].shape, generator=gen))
            return out
        fld.density = noisy
    st = [dict(m=torch.zeros_like(p), v=torch.zeros_like(p)) for p in fld.parameters()]
    quals, losses = [], []
    for k, dr in enumerate(draws):
        f, i = frames[dr["frame"]], dr["inds"]
        out = oren.run(fld, f["o"][i][None], f["d"][i][None], f["nrm"][i][None], AABB4, num_steps=T,
                       upsample_steps=t, t_rand=dr["rt"], u=dr["ru"])
        lc, ls, ld = olosses.nerf_losses(out["image"], out["semantics"], out["depth"], f["rgb"][i][None],
                                         f["label"][i][None], f["depth"][i][None], 1.0)
        loss = olosses.nerf_total_loss(lc, ls, ld)
        for p in fld.parameters():
            p.grad = None
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            for j, (p, s) in enumerate(zip(fld.parameters(), st)):
                pn, s["m"], s["v"] = olosses.adam_step(p, p.grad, s["m"], s["v"], k + 1, LR,
                                                       weight_decay=0.0 if j == 0 else WD)
                p.copy_(pn)
            if k in CHECKPOINTS:
                o = oren.render(fld, cat(frames, "o")[None], cat(frames, "d")[None], cat(frames, "nrm")[None],
                                AABB4, staged=True, max_ray_batch=6144, num_steps=T, upsample_steps=t,
                                u=u_eval[None])
                quals.append(quality(o["image"][0], o["semantics"][0], frames))
    q = np.array(quals)
    return q.mean(0), q, losses


def main():
    sc = scene()
    rows = {}
    for name, threads, noise in (("a: 8 threads", 8, 0.0), ("b: 3 threads", 3, 0.0), ("c: 8 threads, density noise 1e-6", 8, 1e-6)):
        t0 = time.time()
        mean, per, losses = train(*sc, threads, noise)
        rows[name] = (mean, per)
        print(f"{name}: train views {mean[0]:.3f} dB / {mean[1]:.2f} pt, held out {mean[2]:.3f} dB / {mean[3]:.2f} pt;"
              f" loss {losses[0]:.4f} -> {np.mean(losses[-10:]):.4f}; {time.time() - t0:.0f} s", flush=True)
        print("   train-view mIoU per checkpoint: " + " ".join(f"{x:.1f}" for x in per[:, 1]), flush=True)
        print("   train-view PSNR per checkpoint: " + " ".join(f"{x:.2f}" for x in per[:, 0]), flush=True)
    names = list(rows)
    base = rows[names[0]][0]
    for n in names[1:]:
        dlt = rows[n][0] - base
        print(f"{n} minus a: train {dlt[0]:+.3f} dB {dlt[1]:+.2f} pt, held out {dlt[2]:+.3f} dB {dlt[3]:+.2f} pt")


if __name__ == "__main__":
    main()
