"""Is the NeRF training step CPU-bound (Python enqueue time) or GPU-bound?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, copy
import bench
from ucsa_neural_rendering_amd import losses as ul
from ucsa_neural_rendering_amd.nerf.optim import HipAdam
dev = torch.device("cuda", 0)
net, ds = bench.build_field(dev, train_steps=20)
net = net.train()
opt = HipAdam([{"params": list(net.encoder.parameters())}, {"params": list(net.sigma_net.parameters()) + list(net.color_net.parameters()) + list(net.semantics_net.parameters()), "weight_decay": 1e-6}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
g = torch.Generator(device=dev).manual_seed(7)
item = ds[0]; n_rays, T, t = 4096, 256, 256
inds = torch.randint(0, 240 * 320, (n_rays,), device=dev, generator=g)
o, d, nrm = item["rays_o"][inds][None], item["rays_d"][inds][None], item["direction_norms"][inds][None]
gt_rgb = item["img"].reshape(3, -1).t()[inds][None]; labels = item["label"].reshape(-1)[inds][None]; gt_depth = item["depth"].float().reshape(-1)[inds][None]
rt = torch.rand(n_rays, T, device=dev, generator=g); ru = torch.rand(n_rays, t, device=dev, generator=g)
def one(sync_loss=True):
    out = net.render(o, d, nrm, perturb=True, num_steps=T, upsample_steps=t, rng_t=rt, rng_u=ru)
    if sync_loss:
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], gt_rgb, labels, gt_depth, 1.0)
    else:
        lc, ls, ld = ul._NerfLossFn.apply(out["image"], out["semantics"], out["depth"], gt_rgb, labels, gt_depth, 1.0)
    loss = ul.nerf_total_loss(lc, ls, ld)
    opt.zero_grad(); loss.backward(); opt.step()
for mode in (True, False):
    for _ in range(3): one(mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): one(mode)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"sync_in_loss={mode}: enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
