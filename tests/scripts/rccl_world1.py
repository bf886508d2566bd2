"""Worker of tests/test_gpu_rccl_world1.py: ONE process, UCSA_FORCE_DIST=1, a
world-size-1 `nccl` (RCCL) process group on cuda:0.  Every distributed branch
of ucsa_neural_rendering_amd.dist / nerf.optim / bench's cfg4 job is executed
on device tensors over RCCL and compared BIT FOR BIT with the plain
single-process path on the same inputs.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["UCSA_FORCE_DIST"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    os.environ.pop(k, None)

import copy  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ucsa_neural_rendering_amd import dist as udist  # noqa: E402
from ucsa_neural_rendering_amd import losses as ul  # noqa: E402
from ucsa_neural_rendering_amd.nerf.optim import (CollectiveGradScaler, HipAdam,  # noqa: E402
                                                  ShardedHipAdam)
from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays  # noqa: E402

res = {}
rank, local_rank, world = udist.init_from_env()          # backend None -> nccl on a GPU
assert (rank, world) == (0, 1)
res["backend"] = dist.get_backend()
res["active"] = udist.active()
dev = torch.device("cuda", 0)


def groups(net):
    return [{"name": "encoding", "params": list(net.encoder.parameters())},
            {"name": "net", "params": list(net.sigma_net.parameters()) +
             list(net.color_net.parameters()) + list(net.semantics_net.parameters()),
             "weight_decay": 1e-6}]


def equal(a, b):
    return all(torch.equal(p.detach(), q.detach()) for p, q in zip(a.parameters(), b.parameters()))


base = hip_network_from_oracle(lively_oracle_field()).train()
N, T, t = 1024, 32, 32
o, d, n = make_rays(N, 4)
g = torch.Generator().manual_seed(4)
gt_rgb = torch.rand(1, N, 3, generator=g).to(dev)
gt_depth = (torch.rand(1, N, generator=g) * 3 + 0.5).to(dev)
labels = torch.randint(0, 40, (1, N), generator=g).to(dev)
u = torch.rand(N, t, generator=g).to(dev)
tr = torch.rand(N, T, generator=g).to(dev)
o, d, n = o[None].to(dev), d[None].to(dev), n[None].to(dev)

# --- 3 training steps: HipAdam (no collective) vs ShardedHipAdam over RCCL ----
# the backward has float atomics, so ONE backward per step feeds every optimizer
variants = {"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}
nets = {k: copy.deepcopy(base) for k in variants}
ref = {k: copy.deepcopy(base) for k in variants}
opts = {k: ShardedHipAdam(groups(nets[k]), lr=1e-2, betas=(0.9, 0.99), eps=1e-15,
                          comm_dtype=variants[k]) for k in variants}
ropts = {k: HipAdam(groups(ref[k]), lr=1e-2, betas=(0.9, 0.99), eps=1e-15) for k in variants}
scalers = {k: CollectiveGradScaler("cuda", init_scale=2.0 ** 10) for k in variants}
rscalers = {k: torch.amp.GradScaler("cuda", init_scale=2.0 ** 10) for k in variants}


def payload(gr, dt):
    """What one rank's reduce-scatter delivers for payload dtype `dt` at world 1
    (ShardedHipAdam._reduce_scatter_low_precision, restated)."""
    if dt is None:
        return gr
    if dt == torch.bfloat16:
        return gr.to(dt).float()
    amax = gr.abs().max()
    k = torch.exp2(torch.floor(torch.log2(16384.0 / amax)))
    return (gr * k).to(torch.float16).float() * (1.0 / k)


steps_equal = {k: [] for k in variants}
drive = copy.deepcopy(base)
for it in range(3):
    for p, q in zip(drive.parameters(), nets["fp32"].parameters()):
        p.data.copy_(q.data)
        torch.autograd.graph.increment_version(p)
    out = drive.render(o, d, n, perturb=True, num_steps=T, upsample_steps=t, rng_t=tr, rng_u=u)
    lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"], gt_rgb, labels,
                                gt_depth, 1.0)
    for p in drive.parameters():
        p.grad = None
    (ul.nerf_total_loss(lc, ls, ld) * 1024.0).backward()       # the scalers' scale
    grads = [p.grad.detach().clone() for p in drive.parameters()]
    for k, dt in variants.items():
        for p, gr in zip(nets[k].parameters(), grads):
            p.grad = gr.clone()
        for p, gr in zip(ref[k].parameters(), grads):
            big = p.numel() >= (1 << 20)
            p.grad = payload(gr.view(-1), dt).view_as(gr) if big else gr.clone()
        # GradScaler.step/update without a scale(): mark the optimizer state by hand
        for sc, op in ((scalers[k], opts[k]), (rscalers[k], ropts[k])):
            sc.scale(torch.zeros(1, device=dev))     # lazy init of the scale tensor
            sc.step(op)
            sc.update()
        steps_equal[k].append(equal(nets[k], ref[k]))
res["sharded_equals_plain"] = steps_equal
res["last_comm_bytes"] = opts["fp32"].last_comm_bytes

# --- an overflow step: the collective found-inf flag skips the step ----------
k = "fp32"
before = [p.detach().clone() for p in nets[k].parameters()]
for p in nets[k].parameters():
    p.grad = torch.full_like(p, float("inf")) if p.numel() < 4096 else torch.zeros_like(p)
scalers[k].scale(torch.zeros(1, device=dev))
scalers[k].step(opts[k])
scalers[k].update()
res["overflow_step_skipped"] = all(torch.equal(a, p.detach())
                                   for a, p in zip(before, nets[k].parameters()))
res["scale_after_overflow"] = float(scalers[k].get_scale())

# --- the small collectives, on device tensors over RCCL ----------------------
a = torch.arange(1000, dtype=torch.float32, device=dev)
b = torch.arange(300000, dtype=torch.float32, device=dev)
aa, bb = a.clone(), b.clone()
udist.allreduce_sum_([aa, bb])                       # coalesced small + own large
res["allreduce_sum"] = bool(torch.equal(aa, a) and torch.equal(bb, b))
lin = torch.nn.Linear(8, 8).to(dev)
lin.weight.grad = torch.ones_like(lin.weight)
lin.bias.grad = torch.ones_like(lin.bias)
udist.average_grads_(lin.parameters())
res["average_grads"] = bool((lin.weight.grad == 1).all())
bn = torch.nn.BatchNorm2d(4).to(dev)
udist.broadcast_parameters_(bn)
udist.broadcast_buffers_(bn)
res["broadcast"] = True
cm = torch.arange(1600, dtype=torch.int64, device=dev).view(40, 40)
res["confusion"] = bool(torch.equal(udist.allreduce_confusion_(cm.clone()), cm))
res["global_count"] = float(udist.global_count(torch.tensor(7.0, device=dev)))
sm, sd = udist.global_mean_scale(100, torch.tensor(40.0, device=dev))
res["global_mean_scale"] = [float(sm), float(sd)]
res["all_gather_ints"] = udist.all_gather_ints(5, dev)
res["allreduce_max"] = float(udist.allreduce_max_(torch.tensor([3.0], device=dev)))
rows = torch.rand(5, 3, device=dev)
res["gather_rows"] = bool(torch.equal(udist.gather_rows(rows, [5]), rows))
objs = [None]
dist.all_gather_object(objs, ("rank0", str(dev)))
res["all_gather_object"] = objs[0][0]
shard = torch.empty(b.numel(), device=dev)
udist.reduce_scatter_sum_(shard, b)
back = torch.empty_like(b)
udist.all_gather_into_(back, shard)
res["rs_ag_roundtrip"] = bool(torch.equal(back, b))

# --- one cfg4 view with the gather, under RCCL vs without a group ------------
import bench  # noqa: E402
field = copy.deepcopy(base).eval()
field.hip_ray_chunk = 65536
_, _, kept_d = bench.cfg4_job(field, 1, 0, 1, dev, dist=dist, backend="nccl", warmup=0,
                              gather=True, keep=(0,))
_, _, kept_p = bench.cfg4_job(field, 1, 0, 1, dev, dist=None, warmup=0, keep=(0,))
res["cfg4_view_equal"] = all(torch.equal(kept_d[0][k2], kept_p[0][k2])
                             for k2 in ("image", "depth", "semantics"))
dist.barrier()
dist.destroy_process_group()
print(json.dumps(res))
