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


# ---- sharded NeRF optimizer ------------------------------
def _cpu_adam_ops(monkeypatch_target):
    """The HIP Adam entry points replaced by the oracle's Adam (test
    infrastructure): the CPU tests exercise the collectives, slicing, state
    layout and scaler protocol of ShardedHipAdam, not the kernel (that is
    tests/test_gpu_backward.py::test_adam_kernel_matches_torch_adam)."""
    from oracle import losses as olosses

    def adam_step(p, g, m, v, step, lr, b1, b2, eps, wd, inv_grad_scale=1.0):
        np_, nm, nv = olosses.adam_step(p, g * inv_grad_scale, m, v, step, lr, b1,
                                        b2, eps, wd)
        p.copy_(np_), m.copy_(nm), v.copy_(nv)

    def adam_step_scaled(p, g, m, v, step, lr, b1, b2, eps, wd, gs, fi, skipped):
        if float(fi) != 0.0:
            return
        adam_step(p, g / gs, m, v, step - int(skipped), lr, b1, b2, eps, wd)

    def adam_count_skipped(fi, skipped):
        if float(fi) != 0.0:
            skipped += 1

    monkeypatch_target.adam_step = adam_step
    monkeypatch_target.adam_step_scaled = adam_step_scaled
    monkeypatch_target.adam_count_skipped = adam_count_skipped


def _opt_problem():
    g = torch.Generator().manual_seed(11)
    big0 = torch.randn(4099, generator=g)          # not a multiple of the world
    s1 = torch.randn(48, generator=g)
    s2 = torch.randn(7, generator=g)
    tgt = [torch.randn(4099, generator=g), torch.randn(48, generator=g),
           torch.randn(7, generator=g)]
    return [big0, s1, s2], tgt


def _sharded_trajectory(rank, world, sharded, scaler_on, comm_dtype=None,
                        init_scale=64.0, loss_gain=1.0):
    from ucsa_neural_rendering_amd import ops
    from ucsa_neural_rendering_amd.nerf import optim as uoptim
    _cpu_adam_ops(ops)
    uoptim.HipAdam._require_gpu = staticmethod(lambda p: None)  # CPU tensors in this test
    init, tgt = _opt_problem()
    ps = [torch.nn.Parameter(t.clone()) for t in init]
    groups = [{"name": "encoding", "params": ps[:1]},
              {"name": "net", "params": ps[1:], "weight_decay": 1e-6}]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    if sharded:
        opt = uoptim.ShardedHipAdam(groups, shard_min_numel=1024,
                                    comm_dtype=comm_dtype, **kw)
    else:
        opt = uoptim.HipAdam(groups, **kw)
    scaler = uoptim.CollectiveGradScaler("cpu", enabled=scaler_on,
                                         init_scale=init_scale)
    gen = torch.Generator().manual_seed(100 + rank)           # per-rank data
    hist = []
    for it in range(6):
        opt.zero_grad()
        noise = [torch.randn(t.shape, generator=gen) * 0.1 for t in tgt]
        loss = sum((((p - t - n) ** 2).mean() for p, t, n in zip(ps, tgt, noise)))
        loss = loss * loss_gain
        if it == 2 and rank == 1 and scaler_on:
            loss = loss * float("inf")                         # one rank overflows
        scaler.scale(loss).backward()
        if not sharded:
            udist.average_grads_(ps)
        scaler.step(opt)
        scaler.update()
        hist.append(float(scaler.get_scale()) if scaler_on else 0.0)
    state_elems = sum(v.numel() for st in opt.state.values() for k, v in st.items()
                      if torch.is_tensor(v))
    return ([p.detach().clone() for p in ps], hist, state_elems,
            getattr(opt, "last_comm_bytes", None))


def _traj_sharded_scaled(rank, world):
    return _sharded_trajectory(rank, world, True, True)


def _traj_replicated_scaled(rank, world):
    return _sharded_trajectory(rank, world, False, True)


def _traj_sharded_plain(rank, world):
    return _sharded_trajectory(rank, world, True, False)


def _traj_replicated_plain(rank, world):
    return _sharded_trajectory(rank, world, False, False)


def _traj_sharded_fp16(rank, world):
    return _sharded_trajectory(rank, world, True, True, torch.float16)


@pytest.mark.parametrize("scaled", [False, True])
def test_sharded_adam_equals_replicated_adam(scaled):
    """reduce-scatter + Adam on 1/N + all-gather == all-reduce + full Adam,
    bit for bit at world 2 (a+b is commutative, /2 is exact); the replicas
    stay identical; an overflow on ONE rank skips the step on BOTH and backs
    the scale off on both (collective found-inf); the moment buffers hold
    ~1/N of the big tensor."""
    a = spawn(_traj_sharded_scaled if scaled else _traj_sharded_plain)
    b = spawn(_traj_replicated_scaled if scaled else _traj_replicated_plain)
    for r in range(2):
        for x, y in zip(a[r][0], b[r][0]):
            assert torch.equal(x, y)
    for x, y in zip(a[0][0], a[1][0]):
        assert torch.equal(x, y)
    assert a[0][1] == a[1][1] == b[0][1] == b[1][1]
    if scaled:
        assert a[0][1][1] == 64.0 and a[0][1][2] == 32.0      # backed off once
    # 2 moments x (4099 // 2 rounded to 4 + tail) + the small tensors, vs 2 x all
    assert a[0][2] < b[0][2] * 0.6
    assert a[0][3] is not None and a[0][3] > 0


def test_sharded_adam_fp16_gradient_payload_stays_close():
    a = spawn(_traj_sharded_fp16)
    b = spawn(_traj_sharded_scaled)
    for x, y in zip(a[0][0], a[1][0]):
        assert torch.equal(x, y)                               # replicas identical
    for x, y in zip(a[0][0], b[0][0]):
        assert float((x - y).abs().max()) <= 5e-3
    assert a[0][3] < b[0][3]                                   # fewer bytes on the links


def _traj_fp16_big(rank, world):
    # gradients of O(1..10) under the reference's scale 2^16: 6.5e4..6.5e5
    # as fp32, beyond fp16's 65504
    return _sharded_trajectory(rank, world, True, True, torch.float16,
                               init_scale=65536.0, loss_gain=4099.0)


def _traj_fp32_big(rank, world):
    return _sharded_trajectory(rank, world, True, True, None,
                               init_scale=65536.0, loss_gain=4099.0)


def _traj_fp16_plain_tiny(rank, world):
    # no scaler, gradients of ~1e-9: would flush to zero in a bare fp16 cast
    return _sharded_trajectory(rank, world, True, False, torch.float16,
                               loss_gain=1e-6)


def _traj_fp32_plain_tiny(rank, world):
    return _sharded_trajectory(rank, world, True, False, None, loss_gain=1e-6)


def test_fp16_payload_cannot_overflow_after_the_inf_check():
    """the scaler's inf check runs on the fp32, still-scaled
    gradients; a bare cast to fp16 of a value above 65504 (or a sum over the
    ranks passing it) would put inf into Adam AFTER found_inf was computed.
    The payload is range-normalised by the collective max|g|: parameters stay
    finite, equal on the ranks, and close to the fp32-payload trajectory; the
    only skipped step is the deliberately overflowed one."""
    a = spawn(_traj_fp16_big)
    b = spawn(_traj_fp32_big)
    for x, y in zip(a[0][0], a[1][0]):
        assert torch.isfinite(x).all() and torch.equal(x, y)
    for x, y in zip(a[0][0], b[0][0]):
        assert float((x - y).abs().max()) <= 5e-3
    assert a[0][1] == b[0][1]                                  # same scale history
    assert a[0][1][1] == 65536.0 and a[0][1][2] == 32768.0


def test_fp16_payload_without_scaler_keeps_tiny_gradients():
    a = spawn(_traj_fp16_plain_tiny)
    b = spawn(_traj_fp32_plain_tiny)
    init, _ = _opt_problem()
    for x, y, x0 in zip(a[0][0], b[0][0], init):
        assert float((y - x0).abs().max()) > 1e-3              # Adam did move
        assert float((x - y).abs().max()) <= 5e-3


def _gms(rank, world):
    n_loc = 10 + 4 * rank
    valid = torch.tensor(3.0 + rank)
    a, b = udist.global_mean_scale(n_loc, valid)
    return float(a), float(b)


def test_global_mean_scale_world2():
    r0, r1 = spawn(_gms)
    assert abs(r0[0] - 10 / 24) < 1e-6 and abs(r1[0] - 14 / 24) < 1e-6
    assert abs(r0[1] - 3 / 7) < 1e-6 and abs(r1[1] - 4 / 7) < 1e-6
