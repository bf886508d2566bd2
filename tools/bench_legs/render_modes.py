"""Render-side side legs: per-stage times, the MLP-stage roofline object, the
three MLP arithmetics against fp64, the other precision modes."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import *  # noqa: F401,F403


def composite_roofline(mode, mlp_tf, sig_tf, launch_ms):
    """MFMA roofline object of the colour + semantics stage (algorithmic flop
    of the masked samples / launch time)."""
    if mode == "fp32":
        return {
            "kernel": "k_composite (colour+semantics MLPs, fp32 MFMA)",
            "bound": "mfma", "achieved": mlp_tf, "peak": F32_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": mlp_tf / F32_MFMA_PEAK_TF,
            "frac_of_fp16_dense_peak": mlp_tf / F16_MFMA_PEAK_TF,
            "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
            "note": "peak = fp32-input MFMA (the instruction this mode issues, "
                    "1/16 of the 16-bit rate; it runs at the vector FMA rate "
                    "and, measured, does not overlap with VALU work at all: "
                    "kernel time = MFMA busy + VALU issue); "
                    "frac_of_fp16_dense_peak is the same achieved rate against "
                    "SURVEY 8d's 2.5 PF line"}
    if mode == "f16x2":
        return {
            "kernel": "k_weights_compact + k_shade16<f16x2> (colour + "
                      "semantics MLPs, three f16 MFMA passes per fp32 product)",
            "bound": "mfma", "achieved": mlp_tf, "peak": F16_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": mlp_tf / F16_MFMA_PEAK_TF,
            "issued_mfma_tflops": mlp_tf * 3 * 22528 / 19584,
            "issued_frac": mlp_tf * 3 * 22528 / 19584 / F16_MFMA_PEAK_TF,
            "frac_of_fp32_mfma_peak": mlp_tf / F32_MFMA_PEAK_TF,
            "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
            "note": "achieved = ALGORITHMIC fp32 flop of the masked samples / "
                    "launch time against the 2.5 PF 16-bit dense line "
                    "(SURVEY 8d); issued_* counts the three f16 passes and the "
                    "padding (72 MFMAs per 16 samples); the kernel is bound by "
                    "instruction issue (MFMA + VALU cycles add up on a SIMD)"}
    if mode == "bf16x3":
        return {
            "kernel": "k_weights_compact + k_shade16<bf16x3> (colour + "
                      "semantics MLPs, six bf16 MFMA passes per fp32 product)",
            "bound": "mfma", "achieved": mlp_tf, "peak": F16_MFMA_PEAK_TF,
            "unit": "TFLOP/s", "frac": mlp_tf / F16_MFMA_PEAK_TF,
            "issued_mfma_tflops": mlp_tf * 6 * 22528 / 19584,
            "issued_frac": mlp_tf * 6 * 22528 / 19584 / F16_MFMA_PEAK_TF,
            "frac_of_fp32_mfma_peak": mlp_tf / F32_MFMA_PEAK_TF,
            "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
            "note": "achieved = ALGORITHMIC fp32 flop of the masked samples / "
                    "launch time against the 2.5 PF 16-bit dense line "
                    "(SURVEY 8d); issued_* counts the six bf16 passes and the "
                    "padding (144 MFMAs per 16 samples).  Measured (PMC, "
                    "profiles/r03_shade16_pmc.txt): kernel time = MFMA-busy "
                    "cycles + VALU issue cycles, the two do not overlap on a "
                    "SIMD shared by several waves"}
    return {
        "kernel": "k_weights_compact + k_shade16<f16> (colour+semantics "
                  "MLPs on 16x16x32 f16 MFMA, fp32 accumulate)",
        "bound": "mfma", "achieved": mlp_tf, "peak": F16_MFMA_PEAK_TF,
        "unit": "TFLOP/s", "frac": mlp_tf / F16_MFMA_PEAK_TF,
        "launch_ms": launch_ms, "traffic": None, "sigma_mlp_tflops": sig_tf,
        "note": "the nets are 24 MFMAs per 16 samples here: the kernel is "
                "bound by VALU issue (softmax, conversions, ordered per-ray "
                "sums), not by the matrix pipe"}


def mlp_error_vs_fp64(net, dev, M=16384):
    """Max error of the colour / class-probability outputs of the three
    shading arithmetics against an fp64 evaluation of the same nets (torch,
    double, on the device): M random samples, one per ray, through
    ucsa_composite_infer with T = 1 and a huge density (weight 1), so the
    composite returns the nets' outputs themselves.  Shows in the bench line
    that bf16x3 is as close to fp64 as the exact f32-input MFMA chain."""
    from ucsa_neural_rendering_amd import ops
    C = N_CLASSES
    g = torch.Generator(device=dev).manual_seed(11)
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=-1)
    h = torch.randn(M, 16, device=dev, generator=g)
    cp, sp = net.color_net.params.detach(), net.semantics_net.params.detach()
    # fp64 reference: SH-4 of the direction mapped as the reference does
    x, y, z = [(((d[:, i].double() + 1) / 2) * 2 - 1) for i in range(3)]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    sh = torch.stack([
        torch.full_like(x, 0.28209479177387814), -0.48860251190291987 * y,
        0.48860251190291987 * z, -0.48860251190291987 * x, 1.0925484305920792 * xy,
        -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2)], dim=-1)
    geo = h[:, 1:].double()
    one = torch.ones(M, 1, dtype=torch.float64, device=dev)
    cpd, spd = cp.double(), sp.double()
    w1, w2, w3 = cpd[:2048].view(64, 32), cpd[2048:6144].view(64, 64), cpd[6144:7168].view(16, 64)
    xin = torch.cat([sh, geo, one], -1)
    rgb64 = torch.sigmoid(torch.relu(torch.relu(xin @ w1.t()) @ w2.t()) @ w3.t())[:, :3]
    out_pad = (C + 15) // 16 * 16
    s1, s2 = spd[:1024].view(64, 16), spd[1024:1024 + out_pad * 64].view(out_pad, 64)
    p64 = torch.softmax((torch.relu(torch.cat([geo, one], -1) @ s1.t()) @ s2.t())[:, :C], -1)
    zc = torch.ones(M, 1, device=dev)
    sg = torch.full((M, 1), 50.0, device=dev)
    nrm = torch.ones(M, device=dev)
    args = (d, nrm, zc, sg, h, None, None, None)
    res = {}
    for name, pc, ps, kw in (
            ("f32_mfma", ops.mlp_pack(1, cp), ops.mlp_pack(2, sp, C), {}),
            ("bf16x3", ops.mlp_pack_x3(1, cp), ops.mlp_pack_x3(2, sp, C), {"x3": True}),
            ("f16x2", ops.mlp_pack_h2(1, cp), ops.mlp_pack_h2(2, sp, C), {"h2": True}),
            ("fp16", ops.mlp_pack_f16(1, cp), ops.mlp_pack_f16(2, sp, C), {"half": True})):
        img, _, sem = ops.composite_infer(*args, pc, ps, C, **kw)
        res[name] = {"rgb": float((img.double() - rgb64).abs().max()),
                     "class_probability": float((sem.double() - p64).abs().max())}
    res["note"] = ("max |kernel - fp64| over %d random samples on the benchmarked "
                   "field's colour / semantics nets" % M)
    return res


def stage_times(net, o, d, nrm, u, iters=5, image_width=0, half=False,
                mode=None):
    """Per-kernel durations of one chunk, measured with events on the stream
    the kernels run on (torch's current stream), launched the way
    ucsa_render_fwd[_f16|_x3] launches them.  mode: "fp32" (f32-input MFMA,
    fused composite), "fp16" / "bf16x3" / "f16x2" (sigma MLP and the split
    composite pair on the 16-bit MFMA pipe)."""
    from ucsa_neural_rendering_amd import ops
    mode = mode or ("fp16" if half else "fp32")
    half = mode == "fp16"
    x3 = mode == "bf16x3"
    h2 = mode == "f16x2"
    f = (net._field_f16() if half else net._field_x3() if x3 else
         net._field_h2() if h2 else net._field())
    sigma_mlp = (ops.sigma_mlp_fwd_f16 if half else ops.sigma_mlp_fwd_x3 if x3 else
                 ops.sigma_mlp_fwd_h2 if h2 else ops.sigma_mlp_fwd)
    aabb = net._aabb_list(False)
    N = o.shape[0]
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # the fine pass as ucsa_render_fwd* runs it for image-ordered rays (round 5):
    # depth order per tile, encoder and sigma MLP in that order, h / sigma
    # scattered back (csrc/hashgrid_sorted.hip); UCSA_ENC_SORTED=0: as the coarse pass
    sorted_f = bool(image_width) and os.environ.get("UCSA_ENC_SORTED", "2") != "0"
    smode = {"fp32": 0, "fp16": 1, "bf16x3": 2, "f16x2": 3}[mode]
    shipped = (sorted_f and (x3 or h2) and os.environ.get("UCSA_DENSITY_FUSED", "1") != "0")
    names = ["near_far+coarse", "encode_c", "sigma_c", "resample", "sort_f", "encode_f",
             "sigma_f", "composite"]
    acc = {k: 0.0 for k in names}
    rho = 0.0
    for it in range(iters + 1):
        marks = [ev() for _ in range(len(names) + 1)]
        marks[0].record()
        near, far = ops.near_far_from_aabb(o, d, aabb)
        zc = ops.sample_coarse(near, far, T_COARSE)
        marks[1].record()
        feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb,
                                        image_width=image_width, half_features=half)
        marks[2].record()
        hc, sc = sigma_mlp(feat, f["packed_sigma"])
        marks[3].record()
        zf = ops.resample(zc, sc.view(N, T_COARSE), u)
        marks[4].record()
        if sorted_f:
            zs, pix, slot = ops.tile_depth_order(zf, image_width)
            marks[5].record()
            feat = ops.hashgrid_encode_sorted(f["grid"], f["table"], o, d, zs, pix, aabb,
                                              T_FINE, image_width, half_features=half)
            marks[6].record()
            hf, sf = ops.sigma_mlp_fwd_scatter(smode, feat, f["packed_sigma"], slot)
        else:
            marks[5].record()
            feat = ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb,
                                            image_width=image_width, half_features=half)
            marks[6].record()
            hf, sf = sigma_mlp(feat, f["packed_sigma"])
        marks[7].record()
        if it == 0:   # the weights, for the masked fraction rho
            f32 = net._field()
            w = ops.composite_fwd(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                                  sf.view(N, T_FINE), hf, f32["packed_color"],
                                  f32["packed_sem"], N_CLASSES, 1.0, want_aux=True)[4]
        elif half or x3 or h2:
            ops.composite_infer(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                                sf.view(N, T_FINE), hf, f["packed_color"],
                                f["packed_sem"], N_CLASSES, 1.0, half=half, x3=x3, h2=h2)
        else:   # what ucsa_render_fwd launches for fp32: the fused kernel
            ops.composite_fwd(d, nrm, zc, sc.view(N, T_COARSE), hc, zf,
                              sf.view(N, T_FINE), hf, f["packed_color"],
                              f["packed_sem"], N_CLASSES, 1.0)
        marks[8].record()
        # round 6: the density passes AS SHIPPED for the bf16x3 / f16x2 nets on
        # image-ordered rays (csrc/render.hip): BOTH passes depth-ordered per tile
        # (UCSA_ENC_SORTED=2), levels 12-15 through the per-level depth-ordered encoder, levels 0-11 encoded inside the sigma MLP
        # (ucsa_density_sorted = k_hashgrid_encode_sorted + k_density_sorted); the
        # stages above keep the unfused pair for the encoder-only figures
        if shipped:
            extra = [ev() for _ in range(4)]
            extra[0].record()
            zs_c, pix_c, slot_c = ops.tile_depth_order(zc, image_width)
            extra[1].record()
            ops.density_sorted(smode, f["grid"], f["table"], o, d, zs_c, pix_c, slot_c, aabb,
                               T_COARSE, image_width, f["packed_sigma"])
            extra[2].record()
            ops.density_sorted(smode, f["grid"], f["table"], o, d, zs, pix, slot, aabb,
                               T_FINE, image_width, f["packed_sigma"])
            extra[3].record()
        torch.cuda.synchronize()
        if it == 0:
            rho = float((w > 1e-4).float().mean())
            continue
        for i, k in enumerate(names):
            acc[k] += marks[i].elapsed_time(marks[i + 1]) / iters
        if shipped:
            for i, k in enumerate(("order_c", "density_c", "density_f")):
                acc[k] = acc.get(k, 0.0) + extra[i].elapsed_time(extra[i + 1]) / iters
    return acc, rho


def render_mode_legs(result, net, step, chunk_in, args, world, mlp_flop, samples,
                     step_flop, dev):
    """--detail: the same workload in the other arithmetic modes of the three
    MLPs (never the headline `value`, which is --nerf-precision's mode), the
    fp16-table option, and the error of each arithmetic against fp64."""
    ref_img = step(args.warmup + args.steps - 1)["image"]
    n_alt = min(5, args.steps)

    def timed_views():
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_alt):
            step(args.warmup + i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n_alt

    for alt in ("fp32", "bf16x3", "f16x2", "fp16"):
        if alt == args.nerf_precision:
            continue
        net.precision = alt
        dta = timed_views()
        diff = (step(args.warmup + args.steps - 1)["image"] - ref_img).abs().max()
        sta, _ = stage_times(net, *chunk_in, image_width=W, mode=alt)
        cmp_tf = mlp_flop / (sta["composite"] * 1e-3) / 1e12
        sga_tf = samples * 6144 / (0.5 * (sta["sigma_c"] + sta["sigma_f"]) * 1e-3) / 1e12
        key = {"fp32": "f32_mfma_option", "bf16x3": "bf16x3_option",
               "f16x2": "f16x2_option", "fp16": "f16_mlp_option"}[alt]
        vkey = {"fp32": "value_f32_mfma_nets", "bf16x3": "value_bf16x3_nets",
                "f16x2": "value_f16x2_nets", "fp16": "value_fp16_nets"}[alt]
        result[vkey] = world * H * W / dta if world == 1 else None
        result[key] = {
            "rays_per_s": H * W / dta, "ms_per_view": dta * 1e3,
            "max_abs_image_diff_vs_value_mode": float(diff),
            "stage_ms_per_chunk": sta,
            "roofline_composite": composite_roofline(alt, cmp_tf, sga_tf, sta["composite"]),
            "roofline_step_mfma_frac_of_fp16_dense_peak":
                step_flop / (dta * 1e3) / 1e9 / F16_MFMA_PEAK_TF,
            "mlp_arithmetic": MLP_ARITHMETIC[alt],
            "select": "`nerf: {precision: %s}` / --nerf-precision %s" % (alt, alt)}
    # fp16 nets AND the hash grid read from an fp16 copy of the table: what
    # tiny-cuda-nn stores and computes with (`nerf: {precision: fp16,
    # fp16_table: true}`)
    net.precision, net.fp16_table = "fp16", True
    dth = timed_views()
    diff_h = (step(args.warmup + args.steps - 1)["image"] - ref_img).abs().max()
    net.fp16_table = False
    result["value_fp16_nets_fp16_table"] = world * H * W / dth if world == 1 else None
    result["fp16_table_option"] = {
        "rays_per_s": H * W / dth, "ms_per_view": dth * 1e3,
        "max_abs_image_diff_vs_value_mode": float(diff_h),
        "note": "fp16 nets + half2 hash table (26 MB instead of 52 MB; fp32 master copy "
                "with the optimizer); parity: tests/test_gpu_parity.py::test_fp16_table_*",
        "select": "`nerf: {precision: fp16, fp16_table: true}`"}
    net.precision = args.nerf_precision
    result["mlp_error_vs_fp64"] = mlp_error_vs_fp64(net, dev)
