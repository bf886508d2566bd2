"""Where the HOST time of a marched training step goes: cProfile over 300
steady-state steps (after 500 warm-up steps), no GPU sync inside the loop."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucsa_neural_rendering_amd import losses as ul, ops
TILE = int(os.environ.get('TILE_ORDER', '0'))
from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import SemanticNeRFNetwork
from ucsa_neural_rendering_amd.nerf.optim import HipAdam
dev = torch.device("cuda:0")
net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True, density_scale=1,
                          num_semantic_classes=bench.N_CLASSES, seed=123).to(dev).train()
net.march_training = True
ds = SyntheticSceneDataset(0, n_views=16, H=240, W=320, n_classes=bench.N_CLASSES, device=dev)
opt = HipAdam(
    [{"name": "encoding", "params": list(net.encoder.parameters())},
     {"name": "net", "params": list(net.sigma_net.parameters()) + list(net.color_net.parameters()) +
      list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
    lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
g = torch.Generator(device=dev).manual_seed(123)


def step(it):
    if net.refresh_due(it):
        net.update_extra_state()
    item = ds[it % len(ds)]
    inds = torch.randint(0, 240 * 320, (4096,), device=dev, generator=g)
    if TILE:
        inds = ops.tile_order(inds, 320, H=240)
    o, d, nrm = item["rays_o"][inds], item["rays_d"][inds], item["direction_norms"][inds]
    gt_rgb = item["img"].reshape(3, -1).t()[inds][None]
    labels = item["label"].reshape(-1)[inds][None]
    gt_depth = item["depth"].float().reshape(-1)[inds][None]
    out = net.render(o[None], d[None], nrm[None], perturb=True, dt_gamma=1 / 256)
    lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], gt_rgb, labels, gt_depth, 1.0)
    loss = ul.nerf_total_loss(lc, ls, ld)
    opt.zero_grad()
    loss.backward()
    opt.step()


for it in range(500):
    step(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(500, 800):
    step(it)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"300 steps: host loop {t_host/300*1e3:.2f} ms/step, with final sync {t_all/300*1e3:.2f} ms/step")
if os.environ.get("NO_PROFILE"):
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for it in range(800, 1100):
    step(it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
