"""Two ranks (gloo rendezvous, both on cuda:0 -- the dev box has one GPU)
drive the HIP training path on different ray shards: after the SUM all-reduce
of the four parameter gradients every rank must hold bit-identical parameters,
and sharded rendering must reproduce the single-process image.  The driver's
multi-GPU runs use the same code over RCCL.  ``-m gpu``."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
    torch.cuda.set_device(0)
    udist.init_from_env("gloo")
    try:
        net = hip_network_from_oracle(lively_oracle_field()).train()
        opt = HipAdam([{"params": list(net.encoder.parameters())},
                       {"params": list(net.sigma_net.parameters()) +
                        list(net.color_net.parameters()) +
                        list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
                      lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        N, T, t = 512, 32, 32
        o, d, n = make_rays(N, 4)
        g = torch.Generator().manual_seed(4)
        gt_rgb = torch.rand(1, N, 3, generator=g)
        gt_depth = torch.rand(1, N, generator=g) * 3 + 0.5
        labels = torch.randint(0, 40, (1, N), generator=g)
        u = torch.rand(N, t, generator=g)
        tr = torch.rand(N, T, generator=g)
        b, e = udist.shard_range(N, rank, world)
        sl = slice(b, e)
        losses = []
        for it in range(3):
            out = net.render(o[None, sl].cuda(), d[None, sl].cuda(), n[None, sl].cuda(),
                             perturb=True, num_steps=T, upsample_steps=t,
                             rng_t=tr[sl].cuda(), rng_u=u[sl].cuda())
            lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                        gt_rgb[:, sl].cuda(), labels[:, sl].cuda(),
                                        gt_depth[:, sl].cuda(), 1.0)
            loss = ul.nerf_total_loss(lc, ls, ld)
            opt.zero_grad()
            loss.backward()
            ps = list(net.parameters())
            udist.allreduce_grads_(ps)
            for p in ps:
                p.grad.div_(world)
            opt.step()
            losses.append(float(loss.detach()))
        # sharded inference render of all rays, gathered on rank 0
        net.eval()
        with torch.no_grad():
            part = net.render(o[None, sl].cuda(), d[None, sl].cuda(), n[None, sl].cuda(),
                              num_steps=T, upsample_steps=t, rng_u=u[sl].cuda())
            full = net.render(o[None].cuda(), d[None].cuda(), n[None].cuda(),
                              num_steps=T, upsample_steps=t, rng_u=u.cuda())
        img = udist.gather_rows(part["image"][0].cpu(), [udist.shard_range(N, r, world)[1] -
                                                         udist.shard_range(N, r, world)[0]
                                                         for r in range(world)])
        ret[rank] = dict(
            sums=[float(p.detach().double().sum()) for p in net.parameters()],
            head=net.sigma_net.params.detach().cpu()[:16].clone(),
            losses=losses,
            gathered_equal=None if img is None else bool(torch.equal(img, full["image"][0].cpu())))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_keeps_replicas_identical_and_sharded_render_matches():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["sums"] == r1["sums"]
    assert torch.equal(r0["head"], r1["head"])
    assert r0["gathered_equal"] is True
    assert all(l == l for l in r0["losses"] + r1["losses"])  # finite
