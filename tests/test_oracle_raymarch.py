"""KATs that pin the C oracle of the occupancy-grid marching functions
(``oracle/raymarch.c``; reference raymarching.cu:138-855, pcg32.h).  The
reference is CUDA-only and ships no vectors, so these are hand-derived."""
import numpy as np
import torch

from oracle import raymarch as rm
from tests.util import march_scene, slab_near_far

DT = rm.MIN_STEPSIZE


def test_pcg32_published_sequence():
    # pcg-c-basic demo, pcg32_srandom_r(&rng, 42u, 54u): first six outputs
    want = [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b,
            0xcbed606e]
    assert rm.pcg32_sequence(42, 54, 6).tolist() == want
    f = rm.pcg32_first_float(3)
    assert 0.0 <= f < 1.0


def _axis_rays():
    o = np.array([[0.1, 0.2, -3.0], [0.3, -3.0, 0.1]], np.float32)
    d = np.array([[0, 0, 1.0], [0, 1.0, 0]], np.float32)
    return o, d


def test_full_grid_constant_steps():
    """Everything occupied, dt_gamma=0: dt == MIN_STEPSIZE, the count is the
    fp32 accumulation near + k*dt < far, points lie on the ray."""
    o, d = _axis_rays()
    grid = np.ones((1, 16, 16, 16), np.float32)
    near, far = slab_near_far(o, d, 1.0)
    np.testing.assert_allclose(near, [2, 2])
    np.testing.assert_allclose(far, [4, 4])
    xyzs, dirs, deltas, rays, cnt = rm.march_rays_train(
        o, d, 1.0, grid, 1.0, near, far, force_all_rays=True)
    t, k = np.float32(2.0), 0
    while t < np.float32(4.0) and k < 1024:
        t = np.float32(t + DT)
        k += 1
    assert k == 592
    assert rays.tolist() == [[0, 0, k], [1, k, k]]
    assert cnt.tolist() == [2 * k, 2]
    assert xyzs.shape == (2 * k, 3)
    assert np.all(deltas[:, 0] == DT)
    # depth deltas telescope to t - t0
    np.testing.assert_allclose(deltas[:k, 1].sum(), float(t) - 2.0, rtol=1e-5)
    np.testing.assert_array_equal(dirs[:k], np.repeat(d[:1], k, 0))
    np.testing.assert_array_equal(xyzs[:k, 0], np.float32(0.1))
    assert xyzs[0, 2] == np.float32(-3.0) + np.float32(2.0)
    assert np.all(np.diff(xyzs[:k, 2]) > 0) and xyzs[k - 1, 2] <= 1.0


def test_step_cap_and_dt_gamma():
    o, d = _axis_rays()
    grid = np.ones((3, 16, 16, 16), np.float32)
    near, far = slab_near_far(o, d, 4.0)        # 8 long -> > 1024 min steps
    *_, rays, cnt = rm.march_rays_train(o, d, 4.0, grid, 1.0, near, far,
                                        force_all_rays=True)
    assert rays[:, 2].tolist() == [1024, 1024]
    # dt = clamp(t/16, dt_min, 2*bound/H = 0.5): far fewer, growing steps
    _, _, deltas, rays, _ = rm.march_rays_train(
        o, d, 4.0, grid, 1.0, near, far, force_all_rays=True, dt_gamma=1 / 16)
    n0 = rays[0, 2]
    assert 8 < n0 < 64
    dts = deltas[:n0, 0]
    assert np.all(np.diff(dts) >= 0) and dts.max() <= 0.5
    assert dts[0] == np.float32(near[0] * np.float32(1 / 16))


def test_empty_grid_and_threshold():
    o, d, grid, C = march_scene(64, 0)
    near, far = slab_near_far(o, d, 2.0)
    z = np.zeros_like(grid)
    xyzs, _, _, rays, cnt = rm.march_rays_train(o, d, 2.0, z, 1.0, near, far,
                                                force_all_rays=True)
    assert cnt.tolist() == [0, 64] and xyzs.shape[0] == 0
    assert np.all(rays[:, 2] == 0) and rays[:, 0].tolist() == list(range(64))
    # thresh = min(0.01, mean_density): 0.005 cells count only when mean < .005
    g = np.full_like(grid, 0.005)
    *_, c1 = rm.march_rays_train(o, d, 2.0, g, 1.0, near, far,
                                 force_all_rays=True)
    *_, c2 = rm.march_rays_train(o, d, 2.0, g, 0.001, near, far,
                                 force_all_rays=True)
    assert c1[0] == 0 and c2[0] > 0


def test_half_space_skips_empty_cells():
    """Cells with z < 0 empty: no sample may land in an empty cell, the first
    sample sits within one cell + one step of the z = 0 plane."""
    o, d = _axis_rays()
    H = 32
    grid = np.zeros((1, H, H, H), np.float32)
    grid[:, :, :, H // 2:] = 1.0
    near, far = slab_near_far(o[:1], d[:1], 1.0)
    xyzs, _, deltas, rays, _ = rm.march_rays_train(
        o[:1], d[:1], 1.0, grid, 1.0, near, far, force_all_rays=True)
    n = rays[0, 2]
    assert n > 0
    assert xyzs[:n, 2].min() >= 0.0
    assert xyzs[0, 2] <= 2.0 / H + DT
    assert abs(n - 1.0 / DT) <= 2.0 / H / DT + 2


def test_cascade_level_lookup():
    """bound 2, two cascades: level 0 covers |p|<1 at cell size 2/H, level 1
    the rest at 4/H.  Occupy only level 1 -> samples only where max|p| >= 1."""
    o = np.array([[0.05, 0.02, -3.5]], np.float32)
    d = np.array([[0, 0, 1.0]], np.float32)
    H = 16
    grid = np.zeros((2, H, H, H), np.float32)
    grid[1] = 1.0
    near, far = slab_near_far(o, d, 2.0)
    xyzs, _, _, rays, _ = rm.march_rays_train(o, d, 2.0, grid, 1.0, near, far,
                                              force_all_rays=True)
    n = rays[0, 2]
    z = xyzs[:n, 2]
    assert n > 0 and np.all(np.abs(z) >= 1.0)
    assert (z < 0).any() and (z > 0).any()
    grid[:] = 0
    grid[0] = 1.0
    xyzs, _, _, rays, _ = rm.march_rays_train(o, d, 2.0, grid, 1.0, near, far,
                                              force_all_rays=True)
    z = xyzs[:rays[0, 2], 2]
    assert len(z) > 0 and np.all(np.abs(z) < 1.0 + 4.0 / H)


def test_perturb_uses_pcg32_of_ray_index():
    o, d = _axis_rays()
    grid = np.ones((1, 16, 16, 16), np.float32)
    near, far = slab_near_far(o, d, 1.0)
    xyzs, _, _, rays, _ = rm.march_rays_train(o, d, 1.0, grid, 1.0, near, far,
                                              force_all_rays=True, perturb=True)
    for n in range(2):
        t0 = np.float32(near[n]) + DT * np.float32(rm.pcg32_first_float(n, 1))
        ax = 2 if n == 0 else 1
        assert xyzs[rays[n, 1], ax] == np.float32(-3.0) + t0


def test_overflowing_rays_are_dropped():
    o, d, grid, C = march_scene(32, 1)
    near, far = slab_near_far(o, d, 2.0)
    _, _, _, rays_all, cnt = rm.march_rays_train(o, d, 2.0, grid, 0.1, near,
                                                 far, force_all_rays=True)
    total = int(cnt[0])
    cap = total // 2
    xyzs, dirs, deltas, rays, cnt2 = rm.march_rays_train(
        o, d, 2.0, grid, 0.1, near, far, mean_count=cap)
    assert xyzs.shape[0] == cap and int(cnt2[0]) == total
    np.testing.assert_array_equal(rays, rays_all)
    dropped = rays[:, 1] + rays[:, 2] >= cap
    assert dropped.any() and (~dropped).any()
    first = rays[dropped][0]
    assert np.all(deltas[first[1]:cap] == 0)       # nothing written
    sig = np.ones(cap, np.float32)
    rgb = np.ones((cap, 3), np.float32)
    ws, _, img = rm.composite_rays_train(sig, rgb, deltas, rays)
    assert np.all(ws[rays[dropped][:, 0]] == 0)
    assert np.all(img[rays[dropped][:, 0]] == 0)


def _random_samples(seed, N=24, Cs=5):
    rs = np.random.RandomState(seed)
    counts = rs.randint(0, 90, N)
    counts[3] = 0
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    M = int(counts.sum()) + 7
    order = rs.permutation(N)          # rays[] row order != ray index order
    rays = np.stack([np.arange(N), offs, counts], 1).astype(np.int32)[order]
    sig = (rs.rand(M) ** 3 * 40).astype(np.float32)
    rgb = rs.rand(M, 3).astype(np.float32)
    ls = rs.rand(M, Cs).astype(np.float32)
    dl = np.stack([rs.rand(M) * 0.02 + 0.003, rs.rand(M) * 0.05 + 0.003],
                  1).astype(np.float32)
    return sig, rgb, ls, dl, rays


def _composite_torch(sig, rgb, ls, dl, rays, N):
    ws = [None] * N
    dep = [None] * N
    img = [None] * N
    sem = [None] * N
    for idx, off, cnt in rays.tolist():
        s = slice(off, off + cnt)
        alpha = 1 - torch.exp(-sig[s] * dl[s, 0])
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=sig.dtype),
                                     1 - alpha]), 0)[:-1]
        w = alpha * T
        ws[idx] = w.sum()
        dep[idx] = (w * torch.cumsum(dl[s, 1], 0)).sum()
        img[idx] = (w[:, None] * rgb[s]).sum(0)
        sem[idx] = (w.detach()[:, None] * ls[s]).sum(0)
    return torch.stack(ws), torch.stack(dep), torch.stack(img), torch.stack(sem)


def test_composite_train_forward_backward_closed_form():
    sig, rgb, ls, dl, rays = _random_samples(5)
    N = rays.shape[0]
    ws, dep, img, sem = rm.composite_rays_train(sig, rgb, dl, rays, ls)
    ts, tr, tl = (torch.tensor(a, dtype=torch.float64, requires_grad=True)
                  for a in (sig, rgb, ls))
    tws, tdep, timg, tsem = _composite_torch(ts, tr, tl,
                                             torch.tensor(dl).double(), rays, N)
    np.testing.assert_allclose(ws, tws.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(dep, tdep.detach().numpy(), atol=5e-6)
    np.testing.assert_allclose(img, timg.detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(sem, tsem.detach().numpy(), atol=2e-6)
    assert ws[3] == 0 and np.all(img[3] == 0)
    # without semantics: same first three outputs
    ws2, dep2, img2 = rm.composite_rays_train(sig, rgb, dl, rays)
    assert np.array_equal(ws, ws2) and np.array_equal(img, img2)

    rs = np.random.RandomState(9)
    g_ws, g_img, g_sem = (rs.randn(N).astype(np.float32),
                          rs.randn(N, 3).astype(np.float32),
                          rs.randn(N, 5).astype(np.float32))
    loss = ((tws * torch.tensor(g_ws)).sum() + (timg * torch.tensor(g_img)).sum()
            + (tsem * torch.tensor(g_sem)).sum())
    loss.backward()
    g_sig, g_rgb, g_ls = rm.composite_rays_train_backward(
        g_ws, g_img, sig, rgb, dl, rays, ws, img, g_sem)
    np.testing.assert_allclose(g_rgb, tr.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(g_ls, tl.grad.numpy(), atol=1e-5)
    np.testing.assert_allclose(g_sig, ts.grad.numpy(), atol=2e-6, rtol=2e-4)


def test_inference_loop_matches_train_composite():
    """march_rays / composite_rays / compact_rays driven like the reference's
    inference loop reproduce the training composite (up to the T < 1e-4 early
    stop), and compaction keeps order."""
    o, d, grid, C = march_scene(96, 3)
    bound = 2.0
    near, far = slab_near_far(o, d, bound)
    N = o.shape[0]
    xyzs, dirs, deltas, rays, _ = rm.march_rays_train(
        o, d, bound, grid, 0.1, near, far, force_all_rays=True, align=128)
    # (without the alignment padding the last ray has offset + count == M and
    # is dropped by the `>= M` test of reference :243,:338)

    def field(x):
        s = (8.0 * (1 + np.sin(7 * x[:, 0]) * np.cos(5 * x[:, 1]))).astype(np.float32)
        c = (0.5 + 0.5 * np.sin(x * 3)).astype(np.float32)
        l = np.abs(np.cos(x[:, :1] * np.arange(1, 5)[None])).astype(np.float32)
        return s, c, l

    s, c, l = field(xyzs)
    ws_t, dep_t, img_t, sem_t = rm.composite_rays_train(s, c, deltas, rays, l)

    ws = np.zeros(N, np.float32)
    dep = np.zeros(N, np.float32)
    img = np.zeros((N, 3), np.float32)
    sem = np.zeros((N, 4), np.float32)
    alive = [np.arange(N, dtype=np.int32), np.zeros(N, np.int32)]
    rt = [near.astype(np.float32).copy(), np.zeros(N, np.float32)]
    n_alive, i, step = N, 0, 0
    while step < 1024 and n_alive > 0:
        n_step = max(min(N // n_alive, 8), 1)
        x, dd, dl = rm.march_rays(n_alive, n_step, alive[i % 2], rt[i % 2], o,
                                  d, bound, grid, 0.1, near, far)
        s, c, l = field(x)
        rm.composite_rays(n_alive, n_step, alive[i % 2], rt[i % 2], s, c, dl,
                          ws, dep, img, l, sem)
        cnt = np.zeros(1, np.int32)
        rm.compact_rays(n_alive, alive[(i + 1) % 2], alive[i % 2],
                        rt[(i + 1) % 2], rt[i % 2], cnt)
        keep = rt[i % 2][:n_alive] >= 0
        np.testing.assert_array_equal(alive[(i + 1) % 2][:cnt[0]],
                                      alive[i % 2][:n_alive][keep])
        n_alive = int(cnt[0])
        step += n_step
        i += 1
    assert n_alive == 0
    np.testing.assert_allclose(ws, ws_t, atol=2e-4)
    np.testing.assert_allclose(img, img_t, atol=2e-4)
    np.testing.assert_allclose(sem, sem_t, atol=2e-4)
    # training depth counts t from the ray's start (reference :355,:369), the
    # inference loop from the origin (rays_t starts at near, :661,:697)
    np.testing.assert_allclose(dep - ws * near, dep_t, atol=2e-3)
