"""Train the synthetic room THROUGH the marcher (cuda_ray=True,
march_training=True) and through the live path, same budget of steps; report
ms/step, samples per ray, and novel-view PSNR / mIoU of each field rendered by
its own renderer.   python tools/march_train.py [steps] [dt_gamma_inv]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ucsa_neural_rendering_amd import losses as ul, ops  # noqa: E402
from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset  # noqa: E402
from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import SemanticNeRFNetwork  # noqa: E402
from ucsa_neural_rendering_amd.nerf.optim import HipAdam  # noqa: E402
from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter  # noqa: E402


def evaluate(net, ds, fn):
    meter = SemanticsMeter(bench.N_CLASSES)
    ps, ms = [], []
    for v in (3, 7, 12):
        it = ds[v]
        o, d, n = it["rays_o"][None], it["rays_d"][None], it["direction_norms"][None]
        with torch.no_grad():
            fn(o, d, n)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(o, d, n)
            torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        gt = it["img"].reshape(3, -1).t()
        ps.append(float(-10 * torch.log10(((out["image"][0] - gt) ** 2).mean())))
        meter.update(out["semantics"][0].argmax(-1).cpu(), it["label"].reshape(-1).cpu())
    return {"psnr": sum(ps) / 3, "miou": meter.measure()[0], "ms_per_view": sum(ms) / 3,
            "rays_per_s": 76800 / (sum(ms) / 3) * 1e3}


def train(march, steps, dt_gamma, dev, seed=123):
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True, density_scale=1,
                              num_semantic_classes=bench.N_CLASSES, seed=seed).to(dev).train()
    net.march_training = march
    ds = SyntheticSceneDataset(0, n_views=16, H=240, W=320, n_classes=bench.N_CLASSES, device=dev)
    opt = HipAdam(
        [{"name": "encoding", "params": list(net.encoder.parameters())},
         {"name": "net", "params": list(net.sigma_net.parameters()) + list(net.color_net.parameters()) +
          list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
        lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device=dev).manual_seed(seed)
    pts = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        if march and ((it % 16 == 0 or (it < EARLY_REFRESH_STEPS and it % REFRESH_EARLY == 0))
                      if REFRESH_EARLY else net.refresh_due(it)):
            net.update_extra_state(decay=(DECAY_EARLY if it < EARLY_STEPS else 0.95) if DECAY_EARLY else None)
        item = ds[it % len(ds)]
        inds = torch.randint(0, 240 * 320, (4096,), device=dev, generator=g)
        if TILE_ORDER if TILE_ORDER >= 0 else not march:   # default: live path only
            inds = ops.tile_order(inds, 320)
        o, d, nrm = item["rays_o"][inds], item["rays_d"][inds], item["direction_norms"][inds]
        gt_rgb = item["img"].reshape(3, -1).t()[inds][None]
        labels = item["label"].reshape(-1)[inds][None]
        gt_depth = item["depth"].float().reshape(-1)[inds][None]
        if march:
            g_now = dt_gamma
            if COARSE_START and it < COARSE_START:
                g_now = max(dt_gamma, 1 / 64)   # coarse steps while the air is still full
            out = net.render(o[None], d[None], nrm[None], perturb=True, dt_gamma=g_now)
        else:
            out = net.render(o[None], d[None], nrm[None], perturb=True, num_steps=256, upsample_steps=256)
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], gt_rgb, labels, gt_depth, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if march and it % 50 == 49:
            pts.append(int(net.step_counter[(net.local_step - 1) % 16, 0]) / 4096)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    net.eval()
    res = {"ms_per_step": dt / steps * 1e3, "rays_per_s_train": 4096 * steps / dt,
           "final_loss": float(loss.detach())}
    if march:
        res["points_per_ray_every_50_steps"] = [round(p, 1) for p in pts]
    return net, ds, res


TILE_ORDER = int(os.environ.get("TILE_ORDER", "-1"))
COARSE_START = int(os.environ.get("COARSE_START", "0"))
REFRESH_EARLY = int(os.environ.get("REFRESH_EARLY", "0"))   # 0: the renderer's schedule (refresh_due); n: every n steps at first
EARLY_REFRESH_STEPS = int(os.environ.get("EARLY_REFRESH_STEPS", "128"))
DECAY_EARLY = float(os.environ.get("DECAY_EARLY", "0"))  # 0: built-in schedule
EARLY_STEPS = int(os.environ.get("EARLY_STEPS", "256"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    dtg = 1.0 / float(sys.argv[2]) if len(sys.argv) > 2 else 1 / 128
    dev = torch.device("cuda:0")
    out = {"steps": steps, "dt_gamma": dtg}
    net, ds, res = train(True, steps, dtg, dev)
    net.update_extra_state()
    g = net.density_grid
    thr = min(0.01, net.mean_density)
    res["occupied_frac"] = [round(float((g[c] > thr).float().mean()), 4) for c in range(3)]
    res["eval_march"] = evaluate(net, ds, lambda o, d, n: net.run_cuda(o, d, n, dt_gamma=dtg, far_closure=False))
    res["eval_march"]["points_per_ray"] = net.last_march_points / 76800
    res["eval_march_closure"] = evaluate(net, ds, lambda o, d, n: net.run_cuda(o, d, n, dt_gamma=dtg))
    res["eval_live_renderer"] = evaluate(net, ds, lambda o, d, n: net.run(o, d, n, num_steps=96, upsample_steps=96))
    out["trained_through_marcher"] = res
    if not os.environ.get("SKIP_LIVE"):
        net, ds, res = train(False, steps, dtg, dev)
        res["eval_live_renderer"] = evaluate(net, ds, lambda o, d, n: net.run(o, d, n, num_steps=96, upsample_steps=96))
        out["trained_through_live_path"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
