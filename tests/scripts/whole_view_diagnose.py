"""Find a field on which the whole-view test has an unexplained ray and look
inside that ray: the oracle's nets in fp32 vs fp64 on the SAME samples / mask, the
HIP result in the three fp32-grade arithmetics."""
import io, sys, os, contextlib, re, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from oracle import renderer as oren
from oracle import field as ofield
from tests import parity_check as pc
from tests.util import AABB4
from tests.test_gpu_configs import _oracle_from_net
from ucsa_neural_rendering_amd import ops
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses

dev = torch.device("cuda:0")
H, W, T, t = bench.H, bench.W, bench.T_COARSE, bench.T_FINE
for seed in [int(x) for x in os.environ.get("SEEDS", "3,4,7,8,9,10").split(",")]:
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    net, _ = bench.build_field(dev, train_steps=200)
    fld = _oracle_from_net(net)
    net.hip_ray_chunk = 65536
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    pose = _slerp_loop_poses(23, seed=999)[11:12].to(dev)
    o, d, nrm = ops.get_rays(pose, intr, H, W)
    g = torch.Generator(device=dev).manual_seed(1001)
    u = torch.rand(H * W, t, device=dev, generator=g)
    outs = {}
    for prec in ("f16x2", "bf16x3", "fp32"):
        net.precision = prec
        with torch.no_grad():
            outs[prec] = net.render(o, d, nrm, staged=True, perturb=False, num_steps=T,
                                    upsample_steps=t, rng_u=u, image_width=W)
    oc, dc, nc, uc = o.cpu(), d.cpu(), nrm.cpu(), u.cpu()
    bad = []
    for head in range(0, H * W, 32768):
        sel = torch.arange(head, min(head + 32768, H * W))
        rays_sel = (oc[:, sel], dc[:, sel], nc[:, sel])
        with torch.no_grad():
            ref = oren.run(fld, *rays_sel, AABB4, num_steps=T, upsample_steps=t, u=uc[sel], return_aux=True)
        buf = io.StringIO()
        tail = []
        with contextlib.redirect_stdout(buf):
            pc.check_render(outs["f16x2"], ref, fld, rays_sel, AABB4, T, t, sel=sel, tag=f"b{head}",
                            collect_unexplained=tail)
        for line, resid, errs in tail:
            m = re.search(r"ray (\d+): err", line)
            print(f"seed {seed} block {head}: {line[:420]}")
            if max(resid[0], resid[1]) > float(os.environ.get("MIN_RESID", "6e-5")):
                bad.append((head, int(m.group(1)), ref))
    print(f"seed {seed}: {len(bad)} unexplained rays")
    for head, i, ref in bad[:3]:
        gi = head + i
        ro = pc.RayOracle(fld, oc[0, gi], dc[0, gi], nc[0, gi], AABB4, T, t, uc[gi])
        z, sigma, geo, xyz, order = ro.sorted_samples(None)
        weights, rgbs, probs = ro.shade_all(z, sigma, geo, xyz)
        mask = weights > 1e-4
        c32 = ro.composite(z, weights, rgbs, probs, mask)
        # the same nets in fp64 on the same inputs
        f64 = copy.copy(fld)
        f64.color_params, f64.sem_params = fld.color_params.double(), fld.sem_params.double()
        S = z.shape[1]
        dirs = ro.d.view(1, 1, 3).expand(1, S, 3).reshape(-1, 3).double()
        every = torch.ones(S, dtype=torch.bool)
        fg = geo.reshape(S, -1).double()
        logits = ofield.mlp_forward(fld.sem_spec, fg, f64.sem_params)
        p64 = torch.softmax(logits, -1)
        logits32 = ofield.mlp_forward(fld.sem_spec, fg.float(), fld.sem_params)
        w = torch.where(mask, weights, torch.zeros_like(weights)).double()
        sem64 = (w[:, None] * p64).sum(0)
        sem32 = c32["semantics"].double()
        print(f"  ray {gi}: depth {float(c32['depth']):.3f}; live samples {int(mask.sum())}; max weight {float(weights.max()):.4f}; "
              f"max |logit| {float(logits.abs().max()):.1f}; max |geo| {float(geo.abs().max()):.2f}; sigma max {float(sigma.max()):.1f}")
        print(f"    oracle fp32 vs fp64 nets (same samples, same mask): max |d sem| {float((sem32 - sem64).abs().max()):.3e}; "
              f"per-sample max |p32 - p64| {float((probs.double() - p64).abs().max()):.3e}; max |logit32 - logit64| {float((logits32.double() - logits).abs().max()):.3e}")
        for prec in ("f16x2", "bf16x3", "fp32"):
            hs = outs[prec]["semantics"][0, gi].cpu().double()
            print(f"    HIP {prec:7s}: max |sem - oracle32| {float((hs - sem32).abs().max()):.3e}   max |sem - oracle64nets| {float((hs - sem64).abs().max()):.3e}"
                  f"   img err {float((outs[prec]['image'][0, gi].cpu() - c32['image']).abs().max()):.2e}")
        # per sample: the HIP field (fp32 pointwise ops) on the ORACLE's sample positions
        net.precision = "fp32"
        with torch.no_grad():
            den = net.density(xyz.reshape(-1, 3).to(dev))
            hp = net.semantics(xyz.reshape(-1, 3).to(dev), dirs.float().to(dev), geo_feat=den["geo_feat"])
        d_sig = (den["sigma"].cpu() - sigma.reshape(-1)).abs() / sigma.reshape(-1).abs().clamp_min(1e-12)
        d_geo = (den["geo_feat"].cpu() - geo.reshape(S, -1)).abs().max(-1)[0]
        d_p = (hp.cpu().double() - p64).abs().max(-1)[0]
        contrib = w * (hp.cpu().double() - p64).abs().max(-1)[0]
        j = int(contrib.argmax())
        print(f"    HIP field on the oracle's positions: max rel |d sigma| {float(d_sig.max()):.2e}, max |d geo| {float(d_geo.max()):.2e}, "
              f"max |d p| {float(d_p.max()):.2e}; largest w*|dp| at sample {j}: w {float(w[j]):.3e} dp {float(d_p[j]):.3e} "
              f"|d geo| {float(d_geo[j]):.2e} z {float(z[0, j]):.5f}; sum_s w*dp (signed, class argmax) "
              f"{float((w[:, None] * (hp.cpu().double() - p64)).sum(0).abs().max()):.3e}")
        # the HIP pipeline on this ONE ray, stage by stage (fp32 ops), against the oracle's
        fpk = net._field()
        ro_, rd_, rn_ = oc[0, gi:gi + 1].to(dev), dc[0, gi:gi + 1].to(dev), nc[0, gi:gi + 1].to(dev).reshape(1)
        aabb_l = net._aabb_list(False)
        with torch.no_grad():
            nr, fr = ops.near_far_from_aabb(ro_, rd_, aabb_l, 0.2)
            zc_h = ops.sample_coarse(nr, fr, T)
            hc_h, sc_h = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(fpk["grid"], fpk["table"], ro_, rd_, zc_h, aabb_l), fpk["packed_sigma"])
            zf_h = ops.resample(zc_h, sc_h.view(1, T), uc[gi:gi + 1].to(dev), 1.0)
            hf_h, sf_h = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(fpk["grid"], fpk["table"], ro_, rd_, zf_h, aabb_l), fpk["packed_sigma"])
            im_h, dp_h, se_h, src_h, w_h = ops.composite_fwd(rd_, rn_, zc_h, sc_h.view(1, T), hc_h, zf_h, sf_h.view(1, t), hf_h,
                                                           fpk["packed_color"], fpk["packed_sem"], 40, 1.0, want_aux=True)
        zcat = torch.cat([zc_h, zf_h], 1)[0].cpu()
        z_hip = zcat[src_h[0].long().cpu()]
        w_hip = w_h[0].cpu()
        print(f"    single-ray HIP (fp32, ray-ordered): sem vs oracle32 {float((se_h[0].cpu().double() - sem32).abs().max()):.3e}, "
              f"vs the view's render {float((se_h[0].cpu() - outs['fp32']['semantics'][0, gi].cpu()).abs().max()):.3e}")
        print(f"    coarse z max |d| {float((zc_h[0].cpu() - ro.zc[0]).abs().max()):.2e}; sorted z max |d| {float((z_hip - z[0]).abs().max()):.2e}; "
              f"weights max |d| {float((w_hip - weights).abs().max()):.3e} at sample {int((w_hip - weights).abs().argmax())}; "
              f"masks differ at {[int(i) for i in torch.nonzero((w_hip > 1e-4) != (weights > 1e-4)).flatten()]}")
        dw = (w_hip - weights)
        for i in dw.abs().topk(4).indices.tolist():
            print(f"      sample {i}: z hip {float(z_hip[i]):.6f} oracle {float(z[0, i]):.6f}  w hip {float(w_hip[i]):.6e} oracle {float(weights[i]):.6e}  "
                  f"sigma oracle {float(sigma[0, i]):.4f}  p(class k) oracle {float(p64[i, int((se_h[0].cpu().double() - sem32).abs().argmax())]):.4f}")
        zz = z[0]
        gaps = (zz[1:] - zz[:-1])
        print(f"    smallest gaps between sorted depths: {[float(x) for x in gaps.topk(3, largest=False).values]}; "
              f"z[-3:] {[float(x) for x in zz[-3:]]}")
        k = int((outs["f16x2"]["semantics"][0, gi].cpu().double() - sem32).abs().argmax())
        contrib = (w * p64[:, k])
        top = torch.topk(contrib, 4)
        print(f"    class {k}: top contributions " + ", ".join(f"s{int(j)} w={float(weights[j]):.2e} p={float(p64[j, k]):.4f}" for j in top.indices))
    if bad and os.environ.get("STOP_AT_FIRST", "1") == "1":
        break
