"""k_shade16 vs the fused kernels ray by ray (debugging aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucsa_neural_rendering_amd import ops
from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
import ctypes as C
from ucsa_neural_rendering_amd._lib import check, lib
net = hip_network_from_oracle(lively_oracle_field()).eval()
for (N, T, t) in ((64, 16, 16), (700, 96, 96)):
    f = net._field()
    fh = net._field_f16()
    o, d, norms = make_rays(N, 500 + N)
    o, d, norms = o.cuda(), d.cuda(), norms.cuda()
    aabb = net._aabb_list(False)
    near, far = ops.near_far_from_aabb(o, d, aabb)
    zc = ops.sample_coarse(near, far, T)
    hc, sc = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb), f["packed_sigma"])
    sc = sc.view(N, T)
    zf = ops.resample(zc, sc, torch.rand(N, t, device="cuda"))
    hf, sf = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb), f["packed_sigma"])
    sf = sf.view(N, t)
    ref = ops.composite_fwd(d, norms, zc, sc, hc, zf, sf, hf, f["packed_color"], f["packed_sem"], 40)
    image = torch.empty(N, 3, device="cuda"); depth = torch.empty(N, device="cuda"); sem = torch.empty(N, 40, device="cuda")
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    check(lib().ucsa_composite_fwd_f16(p(d), p(norms.view(-1)), p(zc), p(sc), p(hc), p(zf), p(sf), p(hf),
          p(fh["packed_color"]), p(fh["packed_sem"]), N, T, t, 40, 1.0, p(image), p(depth), p(sem), None, None, ops._stream()), "f16")
    h16 = ops.composite_infer(d, norms, zc, sc, hc, zf, sf, hf, fh["packed_color"], fh["packed_sem"], 40, half=True)
    x3 = ops.composite_infer(d, norms, zc, sc, hc, zf, sf, hf, ops.mlp_pack_x3(1, net.color_net.params),
                             ops.mlp_pack_x3(2, net.semantics_net.params, 40), 40, x3=True)
    torch.cuda.synchronize()
    for name, got, want in (("f16 split vs fused f16", h16, (image, depth, sem)), ("x3 vs fused fp32", x3, ref)):
        ei = (got[0] - want[0]).abs().max(-1)[0]
        es = (got[2] - want[2]).abs().max(-1)[0]
        print(f"N={N} {name}: image max {float(ei.max()):.3e} sem max {float(es.max()):.3e}; "
              f"rays with image err > 1e-4: {int((ei > 1e-4).sum())}, sem: {int((es > 1e-4).sum())}")
        bad = (ei > 1e-4).nonzero().flatten()[:12].tolist()
        print("   bad image rays", bad, [float(ei[i]) for i in bad])
        if bad:
            i = bad[0]
            print("   got", got[0][i].tolist(), "want", want[0][i].tolist())
        bad = (es > 1e-4).nonzero().flatten()[:12].tolist()
        print("   bad sem rays", bad, [float(es[i]) for i in bad])
