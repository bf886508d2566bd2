"""Per-kernel stage times of a render chunk at the reference's native sizes
(320 x 240 frame, 256 + 256 samples) next to the cfg2 chunk (640-wide, 96 + 96):
bench.stage_times with the sizes swapped.  Shows where the native frame's
lower per-sample rate comes from."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=200)
for (H, W, T, t) in ((480, 640, 96, 96), (240, 320, 256, 256)):
    bench.T_COARSE, bench.T_FINE = T, t
    o, d, nrm = ops.get_rays(_slerp_loop_poses(4, seed=999)[1:2].to(dev),
                             (0.89 * W, 0.89 * W, W / 2, H / 2), H, W)
    chunk = min(H * W, 65536 - 65536 % (8 * W))
    u = torch.rand(H * W, t, device=dev)
    for mode in ("bf16x3", "fp16"):
        st, rho = bench.stage_times(net, o[0, :chunk].contiguous(), d[0, :chunk].contiguous(),
                                    nrm[0, :chunk, 0].contiguous(), u[:chunk], image_width=W, mode=mode)
        tot = sum(st.values())
        ns = chunk * (T + t)
        print(f"{W}x{H} T={T}+{t} chunk {chunk} rays ({ns/1e6:.1f} M samples) {mode}: "
              + " ".join(f"{k}={v:.3f}" for k, v in st.items())
              + f" | total {tot:.3f} ms = {ns/tot/1e6:.2f} G samples/s, rho {rho:.3f}", flush=True)
    net.precision = "bf16x3"
    net.hip_ray_chunk = 65536
    with torch.no_grad():
        for _ in range(2):
            net.render(o, d, nrm, staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            net.render(o, d, nrm, staged=True, num_steps=T, upsample_steps=t, rng_u=u, image_width=W)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"  whole view {W}x{H}: {dt*1e3:.2f} ms = {H*W/dt/1e6:.2f} M rays/s = {H*W*(T+t)/dt/1e9:.2f} G samples/s", flush=True)
