"""can the TCP-bound encoder and the MFMA+VALU-bound shader
share a CU?  The cfg2 view (5 chunks of 61 440 rays) rendered
  serial     : every stage of every chunk on one stream (what render() does)
  pipelined  : density half (near/far .. sigma_f) of chunk k+1 on a HIGH
               priority stream while the shading half (weights + nets) of
               chunk k runs on a LOW priority stream -- the encoder's 16-wave
               blocks take 320 of a SIMD's 512 VGPRs, one k_shade16 wave per
               SIMD (128 VGPRs) fits next to them
through the staged ops (same kernels, same results).
    python tools/coresident_exp.py            (UCSA_EXP_VIEWS=10)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ucsa_neural_rendering_amd import ops  # noqa: E402
from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses  # noqa: E402

dev = torch.device("cuda:0")
net, _ = bench.build_field(dev, train_steps=int(os.environ.get("PRE", 200)))
H, W, T, t, C = bench.H, bench.W, bench.T_COARSE, bench.T_FINE, bench.N_CLASSES
f = net._field_x3()
aabb = net._aabb_list(False)
intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
pose = _slerp_loop_poses(8, seed=999)[3:4].to(dev)
o, d, nrm = ops.get_rays(pose, intr, H, W)
o, d, nrm = o[0].contiguous(), d[0].contiguous(), nrm[0, :, 0].contiguous()
u = torch.rand(H * W, t, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
chunk = 65536 - 65536 % (8 * W)
heads = list(range(0, H * W, chunk))


def density(k):
    s = slice(heads[k], min(heads[k] + chunk, H * W))
    oo, dd = o[s], d[s]
    n = oo.shape[0]
    near, far = ops.near_far_from_aabb(oo, dd, aabb)
    zc = ops.sample_coarse(near, far, T)
    hc, sc = ops.sigma_mlp_fwd_x3(ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, zc, aabb,
                                                           image_width=W), f["packed_sigma"])
    zf = ops.resample(zc, sc.view(n, T), u[s])
    hf, sf = ops.sigma_mlp_fwd_x3(ops.hashgrid_encode_rays(f["grid"], f["table"], oo, dd, zf, aabb,
                                                           image_width=W), f["packed_sigma"])
    return dict(s=s, n=n, zc=zc, sc=sc, hc=hc, zf=zf, sf=sf, hf=hf)


def shade(r):
    s, n = r["s"], r["n"]
    return ops.composite_infer(d[s], nrm[s], r["zc"], r["sc"].view(n, T), r["hc"], r["zf"],
                               r["sf"].view(n, t), r["hf"], f["packed_color"], f["packed_sem"],
                               C, 1.0, x3=True)


def serial():
    return [shade(density(k)) for k in range(len(heads))]


def pipelined(sd, ss, keep, marks=None):
    main = torch.cuda.current_stream()
    sd.wait_stream(main)
    ss.wait_stream(main)
    outs = []
    for k in range(len(heads)):
        with torch.cuda.stream(sd):
            if marks is not None:
                a = torch.cuda.Event(enable_timing=True)
                a.record(sd)
            r = density(k)
            ev = torch.cuda.Event(enable_timing=marks is not None)
            ev.record(sd)
        with torch.cuda.stream(ss):
            ss.wait_event(ev)
            if marks is not None:
                b = torch.cuda.Event(enable_timing=True)
                b.record(ss)
            outs.append(shade(r))
            if marks is not None:
                c = torch.cuda.Event(enable_timing=True)
                c.record(ss)
                marks.append((a, ev, b, c))
        keep.append(r)
    main.wait_stream(sd)
    main.wait_stream(ss)
    return outs


def stage_alone():
    """Duration of the two halves of a chunk when nothing else runs."""
    ev = lambda: torch.cuda.Event(enable_timing=True)
    dd = ds = 0.0
    for k in range(len(heads)):
        a, b, c = ev(), ev(), ev()
        a.record()
        r = density(k)
        b.record()
        shade(r)
        c.record()
        torch.cuda.synchronize()
        dd += a.elapsed_time(b)
        ds += b.elapsed_time(c)
    return dd / len(heads), ds / len(heads)


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


n_views = int(os.environ.get("UCSA_EXP_VIEWS", 10))
with torch.no_grad():
    ref = serial()
    ms = timed(serial, n_views)
    print(f"UCSA_ENC_LDS_PAD={os.environ.get('UCSA_ENC_LDS_PAD', '0')}: encoder workgroups of 4 waves, "
          "13 KB LDS + the pad; 74 VGPRs -> at most 6 waves / SIMD")
    print(f"serial: {ms:.2f} ms/view = {H * W / ms / 1e3:.2f} M rays/s")
    da, sa = stage_alone()
    print(f"  per chunk, alone: density half {da:.2f} ms, shading half {sa:.2f} ms")
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("stream priority range (low, high):", lo, hi)
    for name, pd, ps in (("density high / shade low", hi, lo), ("equal priority", 0, 0),
                         ("density low / shade high", lo, hi)):
        sd, ss = torch.cuda.Stream(priority=pd), torch.cuda.Stream(priority=ps)
        keep = []
        got = pipelined(sd, ss, keep)
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for x, y in zip(got, ref) for a, b in zip(x, y))

        def run():
            keep.clear()
            pipelined(sd, ss, keep)
        ms = timed(run, n_views)
        marks = []
        keep.clear()
        pipelined(sd, ss, keep, marks)
        torch.cuda.synchronize()
        dd = sum(a.elapsed_time(e) for a, e, _, _ in marks[1:-1]) / max(1, len(marks) - 2)
        dsh = sum(b.elapsed_time(c) for _, _, b, c in marks[1:-1]) / max(1, len(marks) - 2)
        print(f"pipelined [{name}]: {ms:.2f} ms/view = {H * W / ms / 1e3:.2f} M rays/s; "
              f"bit-identical to serial: {same}; per chunk while co-running: density half "
              f"{dd:.2f} ms, shading half {dsh:.2f} ms")
