"""march_rays_train timing: dense vs sparse grids, staged (N <= 8192) vs not."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import march_scene, slab_near_far
from ucsa_neural_rendering_amd.nerf.raymarching import raymarching as rm
for fill, label in ((1.1, "dense"), (0.06, "sparse")):
    for N in (4096, 8192, 8200):
        o, d, grid, C = march_scene(N, 3, bound=4.0, H=128, fill=fill, outside=False)
        near, far = slab_near_far(o, d, 4.0)
        cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        a = [cu(o), cu(d), 4.0, cu(grid), 0.5, cu(near), cu(far)]
        cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
        for _ in range(3):
            cnt.zero_(); out = rm.march_rays_train(*a, cnt, -1, True, 128, True, 1 / 256)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            cnt.zero_(); out = rm.march_rays_train(*a, cnt, -1, True, 128, True, 1 / 256)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{label} N={N}: {dt*1e3:.3f} ms, {int(cnt[0])/N:.1f} pts/ray")
