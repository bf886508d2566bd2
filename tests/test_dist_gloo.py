"""world_size-2 gloo tests (CPU) of the ray-sharding logic: shard ranges,
coalesced SUM all-reduce, ragged gather, confusion-matrix reduce, and the
end-to-end claim that ray-sharded training with SUM-reduced gradients and
globally normalised losses equals single-process training.  The arithmetic
inside each rank is the CPU oracle here (test infrastructure); on the GPU box
the same helpers run over RCCL with the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ucsa_neural_rendering_amd import dist as udist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    udist.init_from_env("gloo")
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 307200):
        for w in (1, 2, 3, 8):
            spans = [udist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
            rr = sorted(sum((udist.shard_round_robin(n, r, w) for r in range(w)), []))
            assert rr == list(range(n))


def _collectives(rank, world):
    big = torch.full((300000,), float(rank + 1))
    smalls = [torch.full((5,), float(rank + 1)), torch.arange(7.0) * (rank + 1)]
    udist.allreduce_sum_([big] + smalls)
    rows = torch.arange(3 + rank).float().view(-1, 1) + 10 * rank
    g = udist.gather_rows(rows, [3, 4])
    cm = torch.eye(4, dtype=torch.int64) * (rank + 1)
    udist.allreduce_confusion_(cm)
    cnt = udist.global_count(torch.tensor(5 + rank))
    return (float(big[0]), [s.tolist() for s in smalls],
            None if g is None else g.view(-1).tolist(), cm.diagonal().tolist(),
            float(cnt))


def test_collectives_world2():
    r0, r1 = spawn(_collectives)
    assert r0[0] == r1[0] == 3.0
    assert r0[1][0] == [3.0] * 5 and r0[1][1] == [0, 3, 6, 9, 12, 15, 18]
    assert r0[2] == [0, 1, 2, 10, 11, 12, 13] and r1[2] is None
    assert r0[3] == r1[3] == [3, 3, 3, 3]
    assert r0[4] == r1[4] == 11.0


def _small_field():
    from oracle import field as ofield
    spec = ofield.make_grid_spec(bound=4.0, log2_hashmap_size=10)
    f = ofield.OracleField(bound=4.0, num_semantic_classes=6, seed=3, grid_spec=spec)
    g = torch.Generator().manual_seed(9)
    f.grid_params = (torch.rand(spec.n_params, generator=g) * 2 - 1) * 2.0
    return f


def _loss_terms(f, o, d, n, u, gt_rgb, labels, gt_depth, n_total, n_valid_total):
    """Local SUMS divided by GLOBAL counts (what each rank back-propagates)."""
    from oracle import renderer as oren
    aabb = torch.tensor([-4.0, -4, -4, 4, 4, 4])
    r = oren.run(f, o[None], d[None], n[None], aabb, num_steps=8,
                 upsample_steps=8, u=u)
    sem = r["semantics"] / r["semantics"].sum(-1, keepdim=True)
    lc = ((r["image"] - gt_rgb) ** 2).sum() / (3 * n_total)
    ls = torch.nn.functional.nll_loss(torch.log(sem + 1e-15).permute(0, 2, 1),
                                      labels, reduction="sum") / n_total
    valid = gt_depth != 0
    ld = (r["depth"][valid] - gt_depth[valid]).abs().sum() / n_valid_total
    return lc + 0.04 * ls + 0.1 * ld


def _data():
    g = torch.Generator().manual_seed(1)
    N = 24
    o = (torch.rand(N, 3, generator=g) * 2 - 1) * 2
    d = torch.randn(N, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    n = torch.ones(N, 1)
    u = torch.rand(N, 8, generator=g)
    gt_rgb = torch.rand(1, N, 3, generator=g)
    labels = torch.randint(0, 6, (1, N), generator=g)
    gt_depth = torch.rand(1, N, generator=g) * 3
    gt_depth[0, ::5] = 0
    return N, o, d, n, u, gt_rgb, labels, gt_depth


def _sharded_step(rank, world):
    N, o, d, n, u, gt_rgb, labels, gt_depth = _data()
    f = _small_field().requires_grad_(True)
    b, e = udist.shard_range(N, rank, world)
    n_valid = udist.global_count((gt_depth[:, b:e] != 0).sum())
    loss = _loss_terms(f, o[b:e], d[b:e], n[b:e], u[b:e], gt_rgb[:, b:e],
                       labels[:, b:e], gt_depth[:, b:e], N, float(n_valid))
    loss.backward()
    grads = [p.grad for p in f.parameters()]
    udist.allreduce_sum_(grads)
    return [g.clone() for g in grads]


def test_ray_sharded_gradients_equal_single_process():
    N, o, d, n, u, gt_rgb, labels, gt_depth = _data()
    f = _small_field().requires_grad_(True)
    loss = _loss_terms(f, o, d, n, u, gt_rgb, labels, gt_depth, N,
                       float((gt_depth != 0).sum()))
    loss.backward()
    ref = [p.grad for p in f.parameters()]
    r0, r1 = spawn(_sharded_step)
    for a, b, c in zip(r0, r1, ref):
        assert torch.equal(a, b)  # identical on every rank -> identical Adam
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-8)
