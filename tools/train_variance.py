"""Run-to-run variance of the synthetic-room pretraining (same seed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    log = {}
    net, ds = bench.build_field(dev, train_steps=steps, cuda_ray=True, log=log)
    net.eval(); net.update_extra_state()
    g = net.density_grid
    thr = min(0.01, net.mean_density)
    it = ds[3]
    with torch.no_grad():
        out = net.run(it["rays_o"][None], it["rays_d"][None], it["direction_norms"][None], num_steps=96, upsample_steps=96)
    gt = it["img"].reshape(3, -1).t()
    psnr = float(-10 * torch.log10(((out["image"][0] - gt) ** 2).mean()))
    print(rep, "loss", log["pretrain_final_loss"], "occupied", [round(float((g[c] > thr).float().mean()), 4) for c in range(3)], "psnr", round(psnr, 2), flush=True)
