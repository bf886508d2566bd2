"""Shared pieces of bench.py and its side legs (tools/bench_legs/*): workload
constants, the SURVEY 8d parameter state, wall-clock log."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 480, 640
T_COARSE, T_FINE = 96, 96
N_CLASSES = 40
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3  # fp32-input MFMA = fp32 vector peak
F16_MFMA_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA (SURVEY 8d's MLP roofline)
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
PMC_JSON = "profiles/r06_pmc_traffic.json"
TRAIN_PMC_JSON = "profiles/r06_train_pmc.json"
SEG_PMC_JSON = "profiles/r03_seg_pmc.json"
# round 5's encoder kernels (tools/encode_pmc.sh r05 + tools/encode_binding_json.py)
ENC_BINDING_JSON = "profiles/r05_encoder_binding.json"


_T0 = time.perf_counter()


def _tick(what):
    """Wall-clock log of the sections of a run (stderr; the JSON line stays
    alone on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:6.1f} s] {what}", file=sys.stderr,
          flush=True)


def build_field(device, seed=123, train_steps=200, log=None, cuda_ray=False,
                deterministic=None):
    """SURVEY 8d parameter state: tcnn-style init (grid U(-1e-4,1e-4), Xavier
    MLPs, seed 123), then `train_steps` Adam steps (lr 1e-2, the reference's
    NeRF optimizer) on the synthetic box-room scene so that sigma is
    non-trivial and the w > 1e-4 mask is selective.  Runs on the HIP training
    path; excluded from the timed region.

    ``deterministic=True`` (the parity tests): the 200 steps run
    with ``net.deterministic`` -- the table gradient through the order-independent
    fixed-point reduction -- so every box trains the SAME field, bit for bit
    (``field_checksum`` prints it), and a failing parity test can be replayed.
    ``None``: as the environment says (``UCSA_DETERMINISTIC``)."""
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import \
        SemanticNeRFNetwork
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    from ucsa_neural_rendering_amd.ops import tile_order
    net = SemanticNeRFNetwork(encoding="hashgrid", bound=4, cuda_ray=cuda_ray,
                              density_scale=1, num_semantic_classes=N_CLASSES,
                              seed=seed).to(device).train()
    env_det = net.deterministic
    if deterministic is not None:
        net.deterministic = bool(deterministic)
    ds = SyntheticSceneDataset(0, n_views=16, H=240, W=320,
                               n_classes=N_CLASSES, device=device)
    opt = HipAdam(
        [{"name": "encoding", "params": list(net.encoder.parameters())},
         {"name": "net", "params": list(net.sigma_net.parameters()) +
          list(net.color_net.parameters()) +
          list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
        lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    g = torch.Generator(device=device).manual_seed(seed)
    t0 = time.perf_counter()
    for it in range(train_steps):
        item = ds[it % len(ds)]
        inds = tile_order(torch.randint(0, 240 * 320, (4096,), device=device, generator=g), 320, H=240)
        o, d, nrm = item["rays_o"][inds], item["rays_d"][inds], item["direction_norms"][inds]
        gt_rgb = item["img"].reshape(3, -1).t()[inds][None]
        labels = item["label"].reshape(-1)[inds][None]
        gt_depth = item["depth"].float().reshape(-1)[inds][None]
        out = net.render(o[None], d[None], nrm[None], perturb=True,
                         num_steps=T_COARSE, upsample_steps=T_FINE,
                         rng_t=torch.rand(4096, T_COARSE, device=device, generator=g),
                         rng_u=torch.rand(4096, T_FINE, device=device, generator=g))
        lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                    gt_rgb, labels, gt_depth, 1.0)
        loss = ul.nerf_total_loss(lc, ls, ld)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    if log is not None:
        log["pretrain_steps"] = train_steps
        log["pretrain_s"] = time.perf_counter() - t0
        log["pretrain_final_loss"] = float(loss.detach())
        log["pretrain_deterministic"] = bool(net.deterministic)
    net.deterministic = env_det
    return net.eval(), ds


def field_checksum(net) -> str:
    """Order-independent integer checksum of the four parameter tensors' BITS (sum of
    the int32 views in int64): equal strings = the same field, bit for bit."""
    parts = []
    for p in (net.encoder.params, net.sigma_net.params, net.color_net.params,
              net.semantics_net.params):
        parts.append("%016x" % (int(p.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())
                                & 0xFFFFFFFFFFFFFFFF))
    return "-".join(parts)


def masked_fraction(net, o, d, nrm, T, t, rt, ru):
    """rho of a ray batch: fraction of the T + t samples per ray whose weight
    passes the reference's mask w > 1e-4 (renderer_semantics.py:249-250) --
    the samples the colour / semantics nets run on.  Staged ops, the
    composite's own aux weights."""
    from ucsa_neural_rendering_amd import ops
    with torch.no_grad():
        f = net._field()
        aabb = net._aabb_list(net.training)
        o, d, nrm = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous(), nrm.reshape(-1).contiguous()
        N = o.shape[0]
        near, far = ops.near_far_from_aabb(o, d, aabb)
        zc = ops.sample_coarse(near, far, T, rt)
        hc, sc = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zc, aabb),
                                   f["packed_sigma"])
        zf = ops.resample(zc, sc.view(N, T), ru)
        hf, sf = ops.sigma_mlp_fwd(ops.hashgrid_encode_rays(f["grid"], f["table"], o, d, zf, aabb),
                                   f["packed_sigma"])
        w = ops.composite_fwd(d, nrm, zc, sc.view(N, T), hc, zf, sf.view(N, t), hf,
                              f["packed_color"], f["packed_sem"], N_CLASSES, 1.0, want_aux=True)[4]
        return float((w > 1e-4).float().mean())


def effective_cores() -> int:
    """Cores this process may actually use: min(cpu_count, affinity, cgroup
    quota).  (The GPU box shows 256 CPUs but a 16-CPU cgroup quota; 256 OpenMP
    threads on 16 CPUs do not finish.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


MLP_ARITHMETIC = {
    "f16x2": "fp32-grade on the f16 MFMA pipe: every fp32 weight and layer input "
             "as two f16 terms, the second scaled by 2^11 (22 significant bits), "
             "three partial products per product (v_mfma_f32_16x16x32_f16), fp32 "
             "accumulation (csrc/mfma_mlp_h2.h); the same 2^-23-per-product error "
             "class as bf16x3 and the f32-input MFMA chain (mlp_error_vs_fp64); "
             "range = the reference's own fp16 nets (layer inputs and weights "
             "below 65504), hidden activations below 2^20",
    "bf16x3": "fp32-grade on the bf16 MFMA pipe: every fp32 weight and layer "
              "input split exactly into three bf16 terms, six partial products "
              "per product (v_mfma_f32_16x16x32_bf16), fp32 accumulation "
              "(csrc/mfma_mlp_x3.h); within 1-2 ulp of the f32-input MFMA mode "
              "(f32_mfma_option.max_abs_image_diff_vs_value_mode)",
    "fp32": "f32-input MFMA (v_mfma_f32_16x16x4_f32): bit for bit a k-ordered "
            "fmaf chain",
    "fp16": "tiny-cuda-nn's own numerics: fp16 weights / layer inputs, fp32 "
            "accumulation (v_mfma_f32_16x16x32_f16)",
}


def nerf_optimizer(net, world, replicated=False, comm_dtype=None):
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam, ShardedHipAdam
    groups = [{"name": "encoding", "params": list(net.encoder.parameters())},
              {"name": "net", "params": list(net.sigma_net.parameters()) +
               list(net.color_net.parameters()) +
               list(net.semantics_net.parameters()), "weight_decay": 1e-6}]
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    if (world > 1 or forced_dist()) and not replicated:
        return ShardedHipAdam(groups, comm_dtype=comm_dtype, **kw)
    return HipAdam(groups, **kw)


DIST_RECORD = None   # set by bench.main(): what torch.distributed actually ran


def forced_dist() -> bool:
    """UCSA_FORCE_DIST=1: a world-size-1 `nccl` (RCCL) process group and the
    distributed code path at N = 1 (ucsa_neural_rendering_amd.dist.forced)."""
    return os.environ.get("UCSA_FORCE_DIST", "") not in ("", "0")


def max_over_ranks(elapsed, dist, dev, backend):
    """The bench contract's timing: MAX over the ranks of the elapsed time."""
    if not dist:
        return elapsed
    tt = torch.tensor([elapsed], dtype=torch.float64,
                      device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def tuning_tables_matched(dev=None):
    """Did the shipped MIOpen find-db / TunableOp table match this box?  (They
    are version- and device-locked; a mismatch silently costs DeepLab ~12 %.)"""
    from ucsa_neural_rendering_amd import _miopen_db
    from ucsa_neural_rendering_amd.network import _gemm_tuning
    mi = _miopen_db.shipped_db_matches(dev)
    mi_use = _miopen_db.shipped_db_in_use()
    _gemm_tuning.ensure()
    tu = _gemm_tuning.table_matches()
    return {"miopen": bool(mi["matched"] and mi_use), "tunableop": bool(tu["matched"]),
            "why": {"miopen": mi["why"] + ("" if mi_use else "; MIOPEN_USER_DB_PATH is the user's"),
                    "tunableop": tu["why"]}}


def flush_c_stdio():
    """fflush(NULL): RCCL prints its banner (version, hostname, library path)
    with C stdio; redirected to a pipe or a file that sits in libc's buffer
    until the process exits -- i.e. AFTER the JSON line Python has long written.
    Flushed where it is produced, the banner stays in front of the line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def finish(dist, rank, result):
    """Tear the group down, then rank 0: full result -> bench_detail.json +
    stderr, compact line -> stdout (tools/bench_legs/headline.py) as the LAST
    thing this process writes there."""
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        from .headline import emit
        result["distributed"] = DIST_RECORD
        emit(result, ROOT)
