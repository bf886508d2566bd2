"""GPU parity of the occupancy-grid marching functions
against the C oracle (``oracle/raymarch.c``), through the C ABI.

Bars: the marchers and the compaction are index / position work evaluated in
the same fp32 order -> BIT-EXACT (offsets, counts, points, deltas).  The
composites sum in a different order (wave scans instead of a sequential loop)
and use the hardware exp2-based ``__expf`` -> 2e-6 absolute on weights / colours
(values in [0,1]), 1e-5 relative on gradients."""
import numpy as np
import pytest
import torch

from oracle import raymarch as orm
from tests.util import march_scene, slab_near_far

pytestmark = pytest.mark.gpu


def _rm():
    from ucsa_neural_rendering_amd.nerf.raymarching import raymarching
    return raymarching


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("N,bound,H,dt_gamma,perturb,outside", [
    (1000, 2.0, 32, 0.0, False, True),
    (257, 1.0, 16, 0.0, True, True),
    (4096, 4.0, 32, 1 / 128, False, False),
    (3001, 2.0, 64, 1 / 128, True, True),
    (64, 4.0, 128, 0.0, False, True),
    (40000, 2.0, 32, 1 / 128, True, True),   # > 32768 rays: lane-per-ray kernels
    (5000, 4.0, 128, 1 / 256, True, False),  # wave-per-ray, long orbits
    (777, 1.0, 64, 0.0, False, False),
    (9000, 2.0, 32, 1 / 128, False, True),   # wave-per-ray without staging rows
    (600, 0.25, 256, 1 / 128, True, True),   # 2*bound/H < MIN_STEPSIZE: dt is dt_max
    (33000, 0.25, 256, 0.0, False, True),    # same, lane-per-ray kernels
])
def test_march_rays_train_bit_exact(N, bound, H, dt_gamma, perturb, outside):
    rm = _rm()
    o, d, grid, C = march_scene(N, N + H, bound=bound, H=H, outside=outside)
    near, far = slab_near_far(o, d, bound)
    ref = orm.march_rays_train(o, d, bound, grid, 0.1, near, far,
                               perturb=perturb, align=128, force_all_rays=True,
                               dt_gamma=dt_gamma)
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    got = rm.march_rays_train(_t(o), _t(d), bound, _t(grid), 0.1, _t(near),
                              _t(far), cnt, -1, perturb, 128, True, dt_gamma)
    assert cnt.tolist() == ref[4].tolist()
    assert ref[4][0] > 0
    np.testing.assert_array_equal(got[3].cpu().numpy(), ref[3])
    for k in range(3):
        assert got[k].shape == ref[k].shape
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])


def test_march_rays_train_capacity_counter_and_empty():
    rm = _rm()
    o, d, grid, C = march_scene(500, 11)
    near, far = slab_near_far(o, d, 2.0)
    *_, cnt_all = orm.march_rays_train(o, d, 2.0, grid, 0.1, near, far,
                                       force_all_rays=True)
    cap = int(cnt_all[0]) // 3
    ref = orm.march_rays_train(o, d, 2.0, grid, 0.1, near, far, mean_count=cap,
                               align=128)
    cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
    got = rm.march_rays_train(_t(o), _t(d), 2.0, _t(grid), 0.1, _t(near),
                              _t(far), cnt, cap, False, 128, False, 0)
    assert cnt.tolist() == ref[4].tolist()
    for k in range(4):
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    # a counter that was not zeroed is the base of the spans
    base = np.array([37, 0], np.int32)
    ref = orm.march_rays_train(o, d, 2.0, grid, 0.1, near, far,
                               step_counter=base.copy(), mean_count=cap)
    cnt = _t(base.copy())
    got = rm.march_rays_train(_t(o), _t(d), 2.0, _t(grid), 0.1, _t(near),
                              _t(far), cnt, cap)
    assert cnt.tolist() == ref[4].tolist()
    for k in range(4):
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    # empty grid, rays that miss the box, no step counter given
    o2 = o.copy()
    o2[:50] = 50.0
    d2 = d.copy()
    d2[:50] = np.array([0, 0, 1.0], np.float32)
    near, far = slab_near_far(o2, d2, 2.0)
    ref = orm.march_rays_train(o2, d2, 2.0, grid, 0.1, near, far,
                               force_all_rays=True, align=128)
    got = rm.march_rays_train(_t(o2), _t(d2), 2.0, _t(grid), 0.1, _t(near),
                              _t(far), None, -1, False, 128, True)
    assert np.all(ref[3][:50, 2] == 0)
    for k in range(4):
        np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
    got = rm.march_rays_train(_t(o), _t(d), 2.0, _t(grid * 0), 0.1, _t(near),
                              _t(far), None, -1, False, -1, True)
    assert got[0].shape[0] == 0 and int(got[3][:, 2].sum()) == 0


def _marched(N=700, seed=4, n_sem=6):
    o, d, grid, C = march_scene(N, seed)
    near, far = slab_near_far(o, d, 2.0)
    xyzs, dirs, deltas, rays, _ = orm.march_rays_train(
        o, d, 2.0, grid, 0.1, near, far, force_all_rays=True, align=128,
        dt_gamma=1 / 128)
    rs = np.random.RandomState(seed)
    M = xyzs.shape[0]
    sig = (rs.rand(M) ** 3 * 60).astype(np.float32)
    rgb = rs.rand(M, 3).astype(np.float32)
    ls = rs.rand(M, n_sem).astype(np.float32)
    ls /= ls.sum(-1, keepdims=True)
    rays = rays[rs.permutation(N)]       # row order != ray id
    return sig, rgb, ls, deltas, rays


@pytest.mark.parametrize("n_sem", [0, 6, 40, 70])
def test_composite_rays_train_forward_backward(n_sem):
    rm = _rm()
    sig, rgb, ls, dl, rays = _marched(n_sem=max(n_sem, 1))
    N = rays.shape[0]
    assert rays[:, 2].max() > 64          # multi-trip rays are covered
    ts = _t(sig).requires_grad_(True)
    tr = _t(rgb).requires_grad_(True)
    rs = np.random.RandomState(1)
    g_ws = rs.randn(N).astype(np.float32)
    g_img = rs.randn(N, 3).astype(np.float32)
    if n_sem:
        tl = _t(ls).requires_grad_(True)
        ref = orm.composite_rays_train(sig, rgb, dl, rays, ls)
        got = rm.composite_rays_train_semantics(ts, tr, tl, _t(dl), _t(rays),
                                                n_sem)
        np.testing.assert_allclose(got[3].detach().cpu().numpy(), ref[3],
                                   atol=2e-6)
    else:
        ref = orm.composite_rays_train(sig, rgb, dl, rays)
        got = rm.composite_rays_train(ts, tr, _t(dl), _t(rays))
    np.testing.assert_allclose(got[0].detach().cpu().numpy(), ref[0], atol=2e-6)
    np.testing.assert_allclose(got[1].detach().cpu().numpy(), ref[1], atol=1e-5)
    np.testing.assert_allclose(got[2].detach().cpu().numpy(), ref[2], atol=2e-6)

    loss = (got[0] * _t(g_ws)).sum() + (got[2] * _t(g_img)).sum()
    g_sem = None
    if n_sem:
        g_sem = rs.randn(N, n_sem).astype(np.float32)
        loss = loss + (got[3] * _t(g_sem)).sum()
    loss.backward()
    refg = orm.composite_rays_train_backward(g_ws, g_img, sig, rgb, dl, rays,
                                             ref[0], ref[2], g_sem)
    scale = np.abs(refg[0]).max()
    np.testing.assert_allclose(ts.grad.cpu().numpy(), refg[0],
                               atol=1e-5 * scale, rtol=1e-4)
    np.testing.assert_allclose(tr.grad.cpu().numpy(), refg[1], atol=5e-6)
    if n_sem:
        np.testing.assert_allclose(tl.grad.cpu().numpy(), refg[2], atol=5e-6)


def test_composite_rays_train_dropped_and_empty_rays():
    rm = _rm()
    sig, rgb, ls, dl, rays = _marched(N=300, seed=8)
    M = sig.shape[0]
    cut = M // 2                      # rays reaching past `cut` are dropped
    rays = rays.copy()
    rays[::17, 2] = 0                 # and some empty rays
    ref = orm.composite_rays_train(sig[:cut], rgb[:cut], dl[:cut], rays,
                                   ls[:cut])
    got = _rm().composite_rays_train_semantics(
        _t(sig[:cut]), _t(rgb[:cut]), _t(ls[:cut]), _t(dl[:cut]), _t(rays), 6)
    dropped = rays[:, 1] + rays[:, 2] >= cut
    assert dropped.any() and (rays[:, 2] == 0).any()
    for k in range(4):
        np.testing.assert_allclose(got[k].cpu().numpy(), ref[k], atol=1e-5)
    assert np.all(got[0].cpu().numpy()[rays[dropped][:, 0]] == 0)
    assert np.all(got[3].cpu().numpy()[rays[dropped][:, 0]] == 0)


def _field(x):
    s = (12.0 * (1 + np.sin(7 * x[:, 0]) * np.cos(5 * x[:, 1]))).astype(np.float32)
    c = (0.5 + 0.5 * np.sin(x * 3)).astype(np.float32)
    l = np.abs(np.cos(x[:, :1] * np.arange(1, 8)[None])).astype(np.float32)
    return s, c, l


@pytest.mark.parametrize("perturb,dt_gamma", [(0, 0.0), (3, 1 / 128)])
def test_inference_loop_march_composite_compact(perturb, dt_gamma):
    """The reference's alive-ray loop (march n_step -> shade -> composite ->
    compact), GPU functions against the oracle at every iteration.  The
    analytic field is evaluated on the host from the ORACLE's points, which
    the GPU points must equal bit for bit."""
    rm = _rm()
    N, bound = 1500, 2.0
    o, d, grid, C = march_scene(N, 21)
    near, far = slab_near_far(o, d, bound)
    go, gd, gg, gn, gf = _t(o), _t(d), _t(grid), _t(near), _t(far)

    r_out = [np.zeros(N, np.float32), np.zeros(N, np.float32),
             np.zeros((N, 3), np.float32), np.zeros((N, 7), np.float32)]
    g_out = [_t(a) for a in r_out]
    r_alive = [np.arange(N, dtype=np.int32), np.zeros(N, np.int32)]
    r_t = [near.astype(np.float32).copy(), np.zeros(N, np.float32)]
    g_alive = [_t(a) for a in r_alive]
    g_t = [_t(a) for a in r_t]
    n_alive, i, step, flips = N, 0, 0, 0
    while step < 1024 and n_alive > 0:
        a, b = i % 2, (i + 1) % 2
        n_step = max(min(N // n_alive, 8), 1)
        ref = orm.march_rays(n_alive, n_step, r_alive[a], r_t[a], o, d, bound,
                             grid, 0.1, near, far, 128, perturb, dt_gamma)
        got = rm.march_rays(n_alive, n_step, g_alive[a], g_t[a], go, gd, bound,
                            gg, 0.1, gn, gf, 128, perturb, dt_gamma)
        for k in range(3):
            np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
        s, c, l = _field(ref[0])
        orm.composite_rays(n_alive, n_step, r_alive[a], r_t[a], s, c, ref[2],
                           r_out[0], r_out[1], r_out[2], l, r_out[3])
        rm.composite_rays_semantics(n_alive, n_step, g_alive[a], g_t[a], _t(s),
                                    _t(c), _t(l), got[2], g_out[0], g_out[1],
                                    g_out[2], g_out[3])
        gt_now = g_t[a].cpu().numpy()
        # the stop test T < 1e-4 may fall either way when exp differs by an ulp
        flip = (gt_now[:n_alive] < 0) != (r_t[a][:n_alive] < 0)
        flips += int(flip.sum())
        if flip.any():       # keep both loops on the same set of alive rays
            g_t[a] = _t(r_t[a])
            for k in range(4):
                g_out[k] = _t(r_out[k])
        else:
            np.testing.assert_allclose(gt_now[:n_alive], r_t[a][:n_alive],
                                       rtol=1e-6)
        rc = np.zeros(1, np.int32)
        gc = torch.zeros(1, dtype=torch.int32, device="cuda")
        orm.compact_rays(n_alive, r_alive[b], r_alive[a], r_t[b], r_t[a], rc)
        rm.compact_rays(n_alive, g_alive[b], g_alive[a], g_t[b], g_t[a], gc)
        assert int(gc.item()) == int(rc[0])
        n_alive = int(rc[0])
        np.testing.assert_array_equal(g_alive[b].cpu().numpy()[:n_alive],
                                      r_alive[b][:n_alive])
        np.testing.assert_allclose(g_t[b].cpu().numpy()[:n_alive],
                                   r_t[b][:n_alive], rtol=1e-6)
        step += n_step
        i += 1
    assert i > 3 and n_alive == 0
    assert flips <= 2
    for k, tol in ((0, 2e-6), (1, 2e-5), (2, 2e-6), (3, 2e-6)):
        np.testing.assert_allclose(g_out[k].cpu().numpy(), r_out[k], atol=tol)
    assert r_out[0].max() > 0.99


def test_composite_rays_without_semantics_and_compact_large():
    rm = _rm()
    rs = np.random.RandomState(2)
    n_alive, n_step, N = 5000, 4, 6000
    alive = rs.permutation(N)[:n_alive].astype(np.int32)
    t0 = rs.rand(n_alive).astype(np.float32)
    sig = (rs.rand(n_alive * n_step) * 30).astype(np.float32)
    rgb = rs.rand(n_alive * n_step, 3).astype(np.float32)
    dl = (rs.rand(n_alive * n_step, 2) * 0.05 + 0.004).astype(np.float32)
    dl[rs.rand(n_alive * n_step) < 0.1] = 0         # exhausted rays
    ws = (rs.rand(N) * 0.9).astype(np.float32)
    ws[alive[:200]] = 0.99995                       # T < 1e-4 on entry
    dep = rs.rand(N).astype(np.float32)
    img = rs.rand(N, 3).astype(np.float32)
    r = [t0.copy(), ws.copy(), dep.copy(), img.copy()]
    g = [_t(a.copy()) for a in (t0, ws, dep, img)]
    orm.composite_rays(n_alive, n_step, alive, r[0], sig, rgb, dl, r[1], r[2],
                       r[3])
    rm.composite_rays(n_alive, n_step, _t(alive), g[0], _t(sig), _t(rgb),
                      _t(dl), g[1], g[2], g[3])
    gt = g[0].cpu().numpy()
    np.testing.assert_array_equal(gt < 0, r[0] < 0)
    assert (r[0] < 0).sum() > 300 and (r[0] >= 0).sum() > 300
    np.testing.assert_allclose(gt, r[0], rtol=1e-6)
    for k in (1, 2, 3):
        np.testing.assert_allclose(g[k].cpu().numpy(), r[k], atol=2e-6)
    # stable compaction, counter used as base
    ra = np.zeros(N, np.int32)
    rt = np.zeros(N, np.float32)
    rc = np.array([5], np.int32)
    orm.compact_rays(n_alive, ra, alive, rt, r[0], rc)
    ga = torch.zeros(N, dtype=torch.int32, device="cuda")
    gtt = torch.zeros(N, device="cuda")
    gc = _t(np.array([5], np.int32))
    rm.compact_rays(n_alive, ga, _t(alive), gtt, g[0], gc)
    assert gc.tolist() == rc.tolist()
    np.testing.assert_array_equal(ga.cpu().numpy()[:rc[0]], ra[:rc[0]])


def test_marching_refuses_cpu_buffers():
    from ucsa_neural_rendering_amd._lib import UcsaError
    rm = _rm()
    o, d, grid, C = march_scene(8, 0)
    near, far = slab_near_far(o, d, 2.0)
    with pytest.raises(UcsaError):
        rm.march_rays_train(_t(o), _t(d), 2.0, _t(grid), 0.1,
                            torch.from_numpy(near), _t(far))
    with pytest.raises(UcsaError):
        rm.compact_rays(4, torch.zeros(8, dtype=torch.int32),
                        torch.zeros(8, dtype=torch.int32), torch.zeros(8),
                        torch.zeros(8), torch.zeros(1, dtype=torch.int32))


def _oracle_loop(o, d, grid, bound, near, far, dt_gamma, n_sem=7):
    """The reference-API inference loop on the oracle (no jitter)."""
    N = o.shape[0]
    out = [np.zeros(N, np.float32), np.zeros(N, np.float32),
           np.zeros((N, 3), np.float32), np.zeros((N, n_sem), np.float32)]
    alive = [np.arange(N, dtype=np.int32), np.zeros(N, np.int32)]
    rt = [near.astype(np.float32).copy(), np.zeros(N, np.float32)]
    n_alive, i, step, pts = N, 0, 0, 0
    while step < 1024 and n_alive > 0:
        a, b = i % 2, (i + 1) % 2
        n_step = max(min(N // n_alive, 8), 1)
        x, _, dl = orm.march_rays(n_alive, n_step, alive[a], rt[a], o, d, bound,
                                  grid, 0.1, near, far, -1, 0, dt_gamma)
        s, c, l = _field(x)
        orm.composite_rays(n_alive, n_step, alive[a], rt[a], s, c, dl, out[0],
                           out[1], out[2], l, out[3])
        cnt = np.zeros(1, np.int32)
        orm.compact_rays(n_alive, alive[b], alive[a], rt[b], rt[a], cnt)
        pts += int((dl[:n_alive * n_step, 0] > 0).sum())
        n_alive = int(cnt[0])
        step += n_step
        i += 1
    return out, pts


@pytest.mark.parametrize("staged", [True, False])
@pytest.mark.parametrize("caps,dt_gamma", [((1024,), 0.0), ((8, 40, 1024), 0.0),
                                           ((32, 96, 1024), 1 / 128),
                                           ((1, 2, 3, 1024), 1 / 128)])
def test_segmented_marcher_equals_reference_loop(caps, dt_gamma, staged):
    """ucsa_march_segment_* (exact-size spans, device-side alive count, wave
    per ray composite with early stop) against the oracle driven through the
    reference-API loop: same samples, so same sums up to re-association."""
    from ucsa_neural_rendering_amd import ops
    N, bound = 1300, 2.0
    o, d, grid, C = march_scene(N, 33)
    near, far = slab_near_far(o, d, bound)
    ref, ref_pts = _oracle_loop(o, d, grid, bound, near, far, dt_gamma)
    seg = ops.MarchSegments(_t(o), _t(d), _t(near), _t(far), _t(grid), 0.1,
                            bound, dt_gamma)
    if not staged:          # two marches per round (count, then write)
        seg.stage_limit = 0
    ws = torch.zeros(N, device="cuda")
    dep = torch.zeros(N, device="cuda")
    img = torch.zeros(N, 3, device="cuda")
    sem = torch.zeros(N, 7, device="cuda")
    n_cap, done, pts, rounds = N, 0, 0, 0
    for cap in caps:
        cap = min(cap, 1024 - done)
        total, n_alive = seg.count(n_cap, cap, 0)
        assert n_alive <= n_cap
        if n_alive == 0:
            break
        if total:
            x, dd, dl = seg.write(n_alive, total, 0)
            assert (seg.stage is not None) == staged
            s, c, l = _field(x.cpu().numpy())
            seg.composite(n_alive, cap, _t(s), 1.0, _t(c), _t(l), dl, ws, dep,
                          img, sem)
            seg.compact(n_alive)
        pts += total
        rounds += 1
        n_cap = n_alive if total else 0
        done += cap
    assert rounds >= 1 and pts > 0
    if len(caps) == 1:   # one round = no early stop = every sample of every ray
        assert pts >= ref_pts
    np.testing.assert_allclose(ws.cpu().numpy(), ref[0], atol=1e-5)
    np.testing.assert_allclose(dep.cpu().numpy(), ref[1], atol=5e-5)
    np.testing.assert_allclose(img.cpu().numpy(), ref[2], atol=1e-5)
    np.testing.assert_allclose(sem.cpu().numpy(), ref[3], atol=1e-5)
    # early termination really happened (the field is dense enough)
    assert (ref[0] > 0.9999).sum() > N // 10


def test_density_grid_points_and_update():
    from ucsa_neural_rendering_amd import ops
    H, bound = 16, 4.0
    for cas in range(3):
        b = min(2.0 ** cas, bound)
        c = ops.density_grid_points(cas, H, bound, 0, "cuda").cpu().numpy()
        i = np.arange(H)
        centre = (b * ((2 * i + 1) / H - 1)).astype(np.float32)
        want = np.stack(np.meshgrid(centre, centre, centre, indexing="ij"),
                        -1).reshape(-1, 3)
        np.testing.assert_allclose(c, want, atol=1e-6)
        j = ops.density_grid_points(cas, H, bound, 5, "cuda").cpu().numpy()
        assert np.abs(j - c).max() <= b / H + 1e-6 and np.abs(j - c).max() > 0.5 * b / H
        # the marcher's lookup maps every jittered point back to its own cell
        lvl = np.ceil(np.log2(np.maximum(np.abs(j).max(-1), 1e-9)))
        cell = np.clip((0.5 * (j / b + 1) * H).astype(np.int64), 0, H - 1)
        own = np.stack(np.meshgrid(i, i, i, indexing="ij"), -1).reshape(-1, 3)
        inside = np.abs(j).max(-1) < b
        np.testing.assert_array_equal(cell[inside], own[inside])
    rs = np.random.RandomState(0)
    g = rs.rand(2, H, H, H).astype(np.float32)
    g[0, 0] = -1.0                      # "never update" cells
    f = (rs.rand(2, H, H, H) * 2).astype(np.float32)
    f[1, 3] = -1.0
    tg = _t(g.copy())
    mean = ops.density_grid_update(tg, _t(f), 0.95, 0.5)
    ok = (g >= 0) & (f * 0.5 >= 0)
    want = np.where(ok, np.maximum(g * np.float32(0.95), f * np.float32(0.5)), g)
    np.testing.assert_array_equal(tg.cpu().numpy(), want)
    np.testing.assert_allclose(float(mean), np.maximum(want, 0).mean(),
                               rtol=1e-6)


def _live_trained_gate(seed):
    """Train through run(), score run() against the marcher; returns the
    statistical verdict (the training is chaotic) after asserting everything
    that is not statistical."""
    import bench
    from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
    dev = torch.device("cuda:0")
    net, ds = bench.build_field(dev, seed=seed, train_steps=1500, cuda_ray=True)
    net.eval()
    assert net.density_grid.shape == (3, 128, 128, 128)
    net.update_extra_state()
    assert net.mean_density > 0 and net.iter_density == 1

    def score(fn):
        meter = SemanticsMeter(bench.N_CLASSES)
        ps = []
        for v in (0, 2, 4, 6, 8, 10, 12, 14):
            it = ds[v]
            out = fn(it["rays_o"][None], it["rays_d"][None],
                     it["direction_norms"][None])
            gt = it["img"].reshape(3, -1).t()
            ps.append(float(-10 * torch.log10(((out["image"][0] - gt) ** 2).mean())))
            meter.update(out["semantics"][0].argmax(-1).cpu(),
                         it["label"].reshape(-1).cpu())
        return sum(ps) / len(ps), meter.measure()[0], out

    with torch.no_grad():
        p_run, m_run, _ = score(lambda o, d, n: net.run(o, d, n, num_steps=256,
                                                        upsample_steps=256))
        p_seg, m_seg, o_seg = score(lambda o, d, n: net.render(o, d, n,
                                                               dt_gamma=1 / 256))
        pts = net.last_march_points / 76800
        p_ref, m_ref, o_ref = score(lambda o, d, n: net.run_cuda(
            o, d, n, dt_gamma=1 / 256, schedule="reference"))
        _, _, o_unf = score(lambda o, d, n: net.run_cuda(
            o, d, n, dt_gamma=1 / 256, fused_shade=False))
        _, _, o_all = score(lambda o, d, n: net.run_cuda(
            o, d, n, dt_gamma=1 / 256, w_min=0.0))
        net.precision = "fp16"
        p_h, m_h, _ = score(lambda o, d, n: net.run_cuda(o, d, n,
                                                         dt_gamma=1 / 256))
        # f16x2 (round 6): sigma MLP and the fused shading kernel's nets as two-term f16
        # operands -- fp32-grade, so the SAME pictures as the f32-input nets up to the
        # decisions a 1e-7 change of a density can flip (mask w > 1e-4, early stop)
        net.precision = "f16x2"
        p_2, m_2, o_2 = score(lambda o, d, n: net.run_cuda(o, d, n, dt_gamma=1 / 256))
        net.precision = "fp32"
    d2 = {k: float((o_2[k] - o_seg[k]).abs().max()) for k in ("image", "depth", "semantics")}
    print(f"seed {seed}: f16x2 marcher vs f32-input marcher: max |d| {d2}; PSNR {p_2:.2f} mIoU {m_2:.4f}")
    assert d2["image"] <= 5e-4 and d2["semantics"] <= 5e-4 and d2["depth"] <= 5e-3
    assert abs(p_2 - p_seg) <= 0.01 and abs(m_2 - m_seg) <= 0.002
    print(f"seed {seed}: PSNR run {p_run:.2f} march {p_seg:.2f} fp16 {p_h:.2f}; "
          f"mIoU run {m_run:.4f} march {m_seg:.4f} fp16 {m_h:.4f}; "
          f"{pts:.1f} points/ray vs 512")
    # (how many points a ray needs depends on how empty the trained field left
    # the air, which varies from run to run: reported above, not asserted)
    assert 0 < pts <= 1024
    for k in ("image", "depth", "semantics"):
        # unfused segments == reference-style loop == fused with w_min 0, up
        # to the order of the fp32 sums (depth is a sum of up to ~1000 terms
        # of size <= 7: relative tolerance)
        # (the sums grow with the points per ray, which depend on how foggy this
        # chaotic training left the field: 2e-4 at <= 64 points, pro rata above)
        tol = 2e-4 * max(1.0, pts / 64.0) * max(1.0, float(o_ref[k].abs().max()))
        assert float((o_unf[k] - o_ref[k]).abs().max()) <= tol
        assert float((o_all[k] - o_ref[k]).abs().max()) <= tol
        # the w > 1e-4 mask drops at most 1e-4 per sample
        assert float((o_seg[k] - o_ref[k]).abs().max()) <= 0.05
    net.reset_extra_state()
    assert float(net.density_grid.abs().sum()) == 0 and net.mean_density == 0
    return (p_run > 25 and p_seg >= p_run - 1.0 and m_seg >= m_run - 0.01 and
            p_h >= p_run - 1.0 and m_h >= m_run - 0.01)


def test_marching_render_quality_gate():
    """SURVEY 8f rank 1 gate: on the synthetic room, the marching render of a
    field trained through the live path loses at most 1 dB PSNR and 1 mIoU
    point against the live render (run, 256+256 samples), with several times
    fewer field evaluations; all schedules give the same picture.

    One-sided (finer steps near surfaces may score HIGHER).  A field trained
    through run() was fitted to run()'s own quadrature (512 samples, the fine
    half concentrated on the surfaces), so another quadrature of it -- the
    marcher's steps of t/256 -- differs by a few tenths of a dB either way:
    -0.5 .. +0.3 dB and -0.001 .. +0.009 mIoU over 20 trainings.  Those
    trainings are chaotic (float atomics in the grid gradient) and now and
    then end in a "foggy" field with > 256 points per ray that misses the
    margin (seen once in ~10 runs), so a miss is re-tried once on a second,
    independently trained field; everything that is not statistical is
    asserted on every attempt.  The +-0.5 of SURVEY 8f is asserted where the
    marcher is used as intended, on a field trained through it (below)."""
    assert _live_trained_gate(123) or _live_trained_gate(321)


# ---------------------------------------------------------------------------
# training through the marcher
# ---------------------------------------------------------------------------
def _oracle_marched_render(fld, o, d, norms, near, xyzs, deltas, rays, w_min):
    """torch (CPU, autograd) restatement of the marched training pass: oracle
    field at the marched points, weights by cumprod, colour / semantics /
    depth from the samples with w > w_min, semantic weights detached."""
    x = torch.from_numpy(xyzs)
    dl = torch.from_numpy(deltas)
    den = fld.density(x)
    sigma, geo = den["sigma"], den["geo_feat"]
    N = o.shape[0]
    img, dep, sem, wsum = [None] * N, [None] * N, [None] * N, [None] * N
    zero3, zeroC = torch.zeros(3), torch.zeros(fld.C)
    for idx, off, cnt in rays.tolist():
        if cnt == 0 or off + cnt >= xyzs.shape[0]:
            img[idx], dep[idx], sem[idx], wsum[idx] = zero3, torch.zeros(()), zeroC, torch.zeros(())
            continue
        s = slice(off, off + cnt)
        alpha = 1 - torch.exp(-sigma[s] * dl[s, 0])
        T = torch.cumprod(torch.cat([torch.ones(1), 1 - alpha]), 0)[:-1]
        w = alpha * T
        t = float(near[idx]) + torch.cumsum(dl[s, 1], 0)
        mask = w > w_min
        dirs = torch.from_numpy(d[idx:idx + 1]).expand(cnt, 3)
        rgb = fld.color(x[s], dirs, mask=mask, geo_feat=geo[s])
        pr = fld.semantics(x[s], dirs, mask=mask, geo_feat=geo[s])
        wm = torch.where(mask, w, torch.zeros_like(w))
        img[idx] = (wm[:, None] * rgb).sum(0)
        dep[idx] = (wm * t).sum() / float(norms[idx])
        sem[idx] = (wm.detach()[:, None] * pr).sum(0)
        wsum[idx] = w.sum()
    return torch.stack(img), torch.stack(dep), torch.stack(sem), torch.stack(wsum)


@pytest.mark.parametrize("N,w_min,perturb", [(96, 1e-4, False), (70, 0.0, True)])
def test_marched_training_gradients_match_oracle_autograd(N, w_min, perturb):
    from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
    from tests.test_gpu_backward import rel_err, rel_l2
    fld = lively_oracle_field().requires_grad_(True)
    net = hip_network_from_oracle(fld, cuda_ray=True).train()
    o, d, norms = make_rays(N, 900 + N)
    o, d, norms = o.numpy(), d.numpy(), norms.numpy().reshape(-1)
    _, _, grid, C = march_scene(4, 5, bound=4.0, H=32, fill=0.25)
    grid = (grid * 40).astype(np.float32)
    net.density_grid = _t(grid)
    net.mean_density = 0.5
    near, far = slab_near_far(o, d, 4.0)
    xyzs, _, deltas, rays, cnt = orm.march_rays_train(
        o, d, 4.0, grid, 0.5, near, far, perturb=perturb, align=128,
        force_all_rays=True, dt_gamma=1 / 64)
    assert 20 * N < cnt[0] < 400 * N
    g = torch.Generator().manual_seed(N)
    ci = torch.rand(N, 3, generator=g)
    cd = torch.rand(N, generator=g)
    cs = torch.rand(N, 40, generator=g)
    # the lively field is nearly opaque per sample; thin it so that a ray
    # spreads its weight over many samples
    with torch.no_grad():
        fld.sigma_params.mul_(0.35)
        net.sigma_net.params.mul_(0.35)
    ref = _oracle_marched_render(fld, o, d, norms, near, xyzs, deltas, rays,
                                 w_min)
    ((ref[0] * ci).sum() + (ref[1] * cd).sum() + (ref[2] * cs).sum()).backward()

    out = net.run_cuda(_t(o)[None], _t(d)[None], _t(norms)[None, :, None],
                       dt_gamma=1 / 64, perturb=perturb, force_all_rays=True,
                       w_min=w_min)
    assert out["image"].requires_grad and net.local_step == 1
    assert net.step_counter[0].tolist() == cnt.tolist()
    for k, r, tol in (("image", ref[0], 2e-5), ("depth", ref[1], 1e-4),
                      ("semantics", ref[2], 2e-5), ("weights_sum", ref[3], 2e-5)):
        assert float((out[k][0].detach().cpu() - r.detach()).abs().max()) <= tol, k
    ((out["image"][0] * ci.cuda()).sum() + (out["depth"][0] * cd.cuda()).sum() +
     (out["semantics"][0] * cs.cuda()).sum()).backward()
    for got, ref_g in ((net.color_net.params.grad, fld.color_params.grad),
                       (net.semantics_net.params.grad, fld.sem_params.grad),
                       (net.sigma_net.params.grad, fld.sigma_params.grad),
                       (net.encoder.params.grad, fld.grid_params.grad)):
        assert rel_l2(got, ref_g) <= 2e-3
        assert rel_err(got, ref_g) <= 2e-2


def test_marched_training_learns_and_uses_mean_count():
    """A short optimisation through the marcher lowers the loss; after
    update_extra_state the point budget comes from the running mean count
    (no read-back), and rays beyond it are dropped, not mis-written."""
    from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    fld = lively_oracle_field()
    net = hip_network_from_oracle(fld, cuda_ray=True).train()
    net.march_training = True
    with torch.no_grad():
        net.sigma_net.params.mul_(0.35)
    N = 1024
    o, d, norms = make_rays(N, 77)
    o, d, norms = o.cuda()[None], d.cuda()[None], norms.cuda()[None]
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(1, N, 3, generator=g).cuda()
    net.update_extra_state()
    assert net.mean_count == 0
    opt = HipAdam([{"name": "all", "params": list(net.parameters())}], lr=5e-3,
                  betas=(0.9, 0.99), eps=1e-15)
    losses = []
    for it in range(40):
        out = net.render(o, d, norms, perturb=True, dt_gamma=1 / 64)
        loss = ((out["image"] - gt) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if it == 19:
            net.update_extra_state()
            assert net.mean_count > 0 and net.local_step == 0
    assert losses[-1] < 0.7 * losses[0]
    assert all(np.isfinite(losses))
    assert int(net.step_counter[:16, 1].max()) == N


def test_field_trained_through_the_marcher_quality_and_sparsity():
    """The intended use of cuda_ray=True: train THROUGH the marcher.  The field
    then carries its own opacity (no far closure needed), the air is empty
    (fewer points per ray than the 192 samples of the live configuration), and
    the marched render of it is at least as good as the live renderer's."""
    import bench
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
    dev = torch.device("cuda:0")
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=True,
                              num_semantic_classes=bench.N_CLASSES,
                              seed=123).to(dev).train()
    net.march_training = True
    ds = SyntheticSceneDataset(0, n_views=16, H=240, W=320,
                               n_classes=bench.N_CLASSES, device=dev)
    opt = HipAdam(
        [{"name": "encoding", "params": list(net.encoder.parameters())},
         {"name": "net", "params": list(net.sigma_net.parameters()) +
          list(net.color_net.parameters()) +
          list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
        lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(800):
        if net.refresh_due(it):
            net.update_extra_state()
        item = ds[it % len(ds)]
        inds = torch.randint(0, 240 * 320, (4096,), device=dev, generator=g)
        out = net.render(item["rays_o"][inds][None], item["rays_d"][inds][None],
                         item["direction_norms"][inds][None], perturb=True,
                         dt_gamma=1 / 256)
        lc, ls, ld = ul.nerf_losses(
            out["image"], out["semantics"], out["depth"],
            item["img"].reshape(3, -1).t()[inds][None],
            item["label"].reshape(-1)[inds][None],
            item["depth"].float().reshape(-1)[inds][None], 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        opt.step()
    net.eval()
    net.update_extra_state()

    def score(fn):
        meter = SemanticsMeter(bench.N_CLASSES)
        ps = []
        for v in (0, 2, 4, 6, 8, 10, 12, 14):
            it = ds[v]
            with torch.no_grad():
                o = fn(it["rays_o"][None], it["rays_d"][None],
                       it["direction_norms"][None])
            gt = it["img"].reshape(3, -1).t()
            ps.append(float(-10 * torch.log10(((o["image"][0] - gt) ** 2).mean())))
            meter.update(o["semantics"][0].argmax(-1).cpu(),
                         it["label"].reshape(-1).cpu())
        return sum(ps) / len(ps), meter.measure()[0]

    p_m, m_m = score(lambda o, d, n: net.render(o, d, n, dt_gamma=1 / 256,
                                                far_closure=False))
    pts = net.last_march_points / 76800
    p_l, m_l = score(lambda o, d, n: net.run(o, d, n, num_steps=256,
                                             upsample_steps=256))
    print(f"marcher-trained field: PSNR march {p_m:.2f} live {p_l:.2f}; mIoU "
          f"march {m_m:.4f} live {m_l:.4f}; {pts:.1f} points/ray")
    assert p_m > 25 and p_m >= p_l - 0.5
    assert m_m >= m_l - 0.005
    assert pts < 192


def test_run_cuda_edge_cases():
    """Empty batch, rays that miss the box, an empty grid, batched prefix
    shapes, CPU tensors: inference and training passes."""
    from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
    from ucsa_neural_rendering_amd._lib import UcsaError
    fld = lively_oracle_field()
    net = hip_network_from_oracle(fld, cuda_ray=True).eval()
    net.update_extra_state()
    o, d, n = make_rays(300, 1)
    o, d, n = o.cuda(), d.cuda(), n.cuda()
    with torch.no_grad():
        # [B, N, 3] prefix and the 1-D output shapes of run()
        out = net.run_cuda(o.view(3, 100, 3), d.view(3, 100, 3), n.view(3, 100, 1))
        assert out["image"].shape == (3, 100, 3) and out["depth"].shape == (3, 100)
        assert out["semantics"].shape == (3, 100, 40)
        flat = net.run_cuda(o[None], d[None], n[None])
        assert torch.equal(flat["image"][0], out["image"].view(300, 3))
        # empty batch
        e = net.run_cuda(o[None, :0], d[None, :0], n[None, :0])
        assert e["image"].shape == (1, 0, 3) and e["semantics"].shape == (1, 0, 40)
        # every ray misses the box: nothing marched, the closure paints the
        # (clamped) far point exactly as run() does
        o2 = torch.full_like(o, 50.0)
        d2 = torch.zeros_like(d)
        d2[:, 2] = 1.0
        miss = net.run_cuda(o2[None], d2[None], n[None])
        assert net.last_march_points == 0 and torch.isfinite(miss["image"]).all()
        ref = net.run(o2[None], d2[None], n[None], num_steps=8, upsample_steps=0)
        assert float((miss["image"] - ref["image"]).abs().max()) <= 1e-5
        opened = net.run_cuda(o2[None], d2[None], n[None], far_closure=False)
        assert float(opened["image"].abs().max()) == 0 and float(opened["weights_sum"].max()) == 0
        # an all-empty grid
        net.density_grid.zero_()
        z = net.run_cuda(o[None], d[None], n[None], far_closure=False)
        assert net.last_march_points == 0 and float(z["image"].abs().max()) == 0
        with pytest.raises(UcsaError):
            net.run_cuda(o[None].cpu(), d[None].cpu(), n[None].cpu())
    # training pass with nothing to march: zero image, finite zero gradients
    net.train()
    net.march_training = True
    t = net.render(o[None], d[None], n[None], perturb=True, force_all_rays=True)
    assert float(t["image"].detach().abs().max()) == 0 and t["image"].requires_grad
    t["image"].sum().backward()
    for p in net.parameters():
        assert p.grad is not None and float(p.grad.abs().max()) == 0
    t0 = net.render(o[None, :0], d[None, :0], n[None, :0], force_all_rays=True)
    assert t0["image"].shape == (1, 0, 3)
