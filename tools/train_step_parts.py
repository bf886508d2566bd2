"""Where a data-parallel NeRF training step (tools/bench_legs/train.py dp_train_leg.one, world 1) spends its
time beyond the pure step of train_throughput: batch assembly pieces timed separately (each with a sync)
and the whole step asynchronously."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tools.bench_legs.common import nerf_optimizer
from ucsa_neural_rendering_amd import losses as ul, ops
dev = torch.device("cuda:0")
net, ds = bench.build_field(dev, train_steps=50)
net = net.train()
opt = nerf_optimizer(net, 1)
g = torch.Generator(device=dev).manual_seed(7)
n_rays, T, t = 4096, 256, 256

def sync_time(fn, k=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, r

it = [0]
def get_item():
    it[0] += 1
    return ds[it[0] % len(ds)]
ms_item, item = sync_time(get_item)
def draw():
    inds = torch.randint(0, 240 * 320, (n_rays,), device=dev, generator=g)
    return ops.tile_order(inds, 320, H=240)
ms_draw, inds = sync_time(draw)
def gather():
    return (item["rays_o"][inds][None], item["rays_d"][inds][None], item["direction_norms"][inds][None],
            item["img"].reshape(3, -1).t()[inds][None], item["label"].reshape(-1)[inds][None],
            item["depth"].float().reshape(-1)[inds][None])
ms_gather, (o, d, nrm, rgb, lab, dep) = sync_time(gather)
def rands():
    return torch.rand(n_rays, T, device=dev, generator=g), torch.rand(n_rays, t, device=dev, generator=g)
ms_rand, (rt, ru) = sync_time(rands)
def core():
    out = net.render(o, d, nrm, perturb=True, num_steps=T, upsample_steps=t, rng_t=rt, rng_u=ru)
    lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], rgb, lab, dep, 1.0)
    loss = ul.nerf_total_loss(lc, ls, ld)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
ms_core, _ = sync_time(core)
def whole():
    global item, inds, o, d, nrm, rgb, lab, dep, rt, ru
    item = get_item(); inds = draw(); o, d, nrm, rgb, lab, dep = gather(); rt, ru = rands()
    return core()
ms_whole, _ = sync_time(whole)
# host time of enqueueing the whole step (no sync inside): how long Python needs
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    whole()
host = (time.perf_counter() - t0) / 50 * 1e3
torch.cuda.synchronize()
print(f"dataset item {ms_item:.3f} ms | draw+tile_order {ms_draw:.3f} | 6 gathers {ms_gather:.3f} | 2 rand {ms_rand:.3f} | "
      f"render+loss+bwd+Adam {ms_core:.3f} | whole step {ms_whole:.3f} (host enqueue time {host:.3f})")
print("item keys:", {k: (tuple(v.shape), str(v.device), str(v.dtype)) for k, v in item.items() if torch.is_tensor(v)})
