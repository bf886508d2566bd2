#!/usr/bin/env python3
"""Headline benchmark: Semantic-NeRF render throughput (rays/sec), BASELINE
cfg2 -- one 640x480 view per step, 192 samples/ray (96 coarse + 96 fine),
hash grid L=16, MLP width 64, 40 classes, fp32.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run, one rank per
GPU.  Rendering shards by view (rays are independent): every rank renders its
own 640x480 view per step with replicated parameters and NO data-path
collective ("scaling": "weak"); the only collectives are the timing barrier
and the max-over-ranks of the elapsed time.

stdout carries ONE compact JSON line (<= 4 KB, tools/bench_legs/headline.py):
metric / value / ms_per_step / config{workload, ...} / roofline (dominant
kernel: the hash-grid gather, algorithmic bytes per launch / live event-timed
launch duration against the 8 TB/s HBM line, PMC traffic, the unit that
actually binds it) / cpu_baseline (the CPU oracle, "port", on the host cores,
bounded sample, with gpu_over_cpu inside it) / distributed / tuning_tables_matched.
Everything else a run measures (per-stage times, the composite roofline, the
training step, with --detail the other arithmetic modes, the marcher, the
DeepLab step) goes to bench_detail.json and stderr.

Other modes (never the driver's default; code in tools/bench_legs/):
  --mode train   value = rays/s TRAINED (4096 rays x (256+256) per rank and
                 step, fwd + bwd + gradient collectives + Adam), weak scaling
  --mode cfg3    the joint step (8 renders + 8 NeRF steps + DeepLab step)
  --mode cfg4    BASELINE cfg4: 512 novel views of 640x480 round-robin over the
                 ranks, no data-path collective, optional gather to rank 0
  --mode cfg5    BASELINE cfg5: the continual loop over 10 synthetic rooms at
                 240x320, replay buffer, joint NeRF + DeepLab; final mIoU

UCSA_FORCE_DIST=1 at N = 1 creates a world-size-1 `nccl` (RCCL) process group
and takes every distributed branch (sharded Adam's reduce-scatter /
all-gather, coalesced all-reduce, collective found-inf, timing reductions).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.bench_legs import common  # noqa: E402
from tools.bench_legs.common import (ENC_BINDING_JSON, F16_MFMA_PEAK_TF,  # noqa: E402,F401
                                     F32_MFMA_PEAK_TF, H, HBM_PEAK_GBS,
                                     MLP_ARITHMETIC, N_CLASSES, PMC_JSON, T_COARSE,
                                     T_FINE, W, _tick, build_field, effective_cores,
                                     forced_dist, max_over_ranks)
from tools.bench_legs.cfg4 import cfg4_job  # noqa: E402,F401  (tests/test_gpu_configs.py)
from tools.bench_legs.render_modes import stage_times  # noqa: E402,F401  (tools/*.py)
from tools.bench_legs.train import train_throughput  # noqa: E402,F401  (tools/*.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=32768,
                    help="rays per pass of the CPU baseline (3 passes, median)")
    ap.add_argument("--pretrain-steps", type=int, default=200,
                    help="Adam steps on the synthetic scene before timing "
                         "(SURVEY 8d: 200)")
    ap.add_argument("--no-train-bench", action="store_true")
    ap.add_argument("--detail", action="store_true",
                    help="also run the side legs (other MLP arithmetics, fp16 "
                         "table, marcher, every training precision, DeepLab) "
                         "into bench_detail.json; the stdout line is the same")
    ap.add_argument("--scenes", type=int, default=10, help="cfg5: rooms in the loop")
    ap.add_argument("--nerf-epochs", type=int, default=2, help="cfg5: NeRF-only epochs per scene")
    ap.add_argument("--joint-epochs", type=int, default=1, help="cfg5: joint epochs per scene")
    ap.add_argument("--frames", type=int, default=16, help="cfg5: training frames per scene")
    ap.add_argument("--pretrain-seg-steps", type=int, default=400,
                    help="cfg5: Adam steps of the in-harness DeepLab pre-training on eight "
                         "other rooms before the clock starts (the reference loads a "
                         "ScanNet-25k checkpoint); 0 = random initialisation")
    ap.add_argument("--mode", choices=["render", "train", "cfg3", "cfg4", "cfg5"],
                    default="render")
    ap.add_argument("--backbone", default=None,
                    help="DeepLab backbone.  cfg3 default resnet50 (BASELINE cfg3 says "
                         "ResNet-50; the reference's model is resnet101); cfg5 default: "
                         "what cfg/exp/multi_step/cl_base.yml says (resnet101)")
    ap.add_argument("--seg-amp", default="", help="cfg3: '' (fp32, the reference) or bf16")
    ap.add_argument("--nerf-precision", default="f16x2",
                    choices=["fp32", "bf16x3", "f16x2", "fp16"],
                    help="arithmetic of the three MLPs in the no-grad renders: "
                         "f16x2 (default; fp32-grade on the f16 MFMA pipe, two-term "
                         "operands, three passes per product, csrc/mfma_mlp_h2.h), "
                         "bf16x3 (fp32-grade on the bf16 pipe, three-term operands, "
                         "six passes, csrc/mfma_mlp_x3.h), fp32 (f32-input MFMA, an "
                         "exact fmaf chain), fp16 (tiny-cuda-nn's own numerics)")
    ap.add_argument("--seg-find", action="store_true",
                    help="DeepLab legs with torch.backends.cudnn.benchmark (MIOpen's "
                         "exhaustive solver search, what scripts/train_joint.py "
                         "uses): several minutes of search on a fresh box, "
                         "49 -> 44 ms per R-101 fp32 step")
    ap.add_argument("--no-seg-find", action="store_true",
                    help="cfg3: immediate-mode MIOpen solvers instead of the "
                         "module's default exhaustive search (profiling runs)")
    ap.add_argument("--replicated-adam", action="store_true",
                    help="train legs: all-reduce + full Adam on every rank "
                         "instead of the sharded optimizer")
    ap.add_argument("--grad-comm-dtype", choices=["fp32", "fp16", "bf16"],
                    default="fp32")
    ap.add_argument("--fresh", action="store_true",
                    help="--mode train: start from the initialisation (quality "
                         "comparisons of the gradient payload precision)")
    ap.add_argument("--views", type=int, default=512, help="cfg4: views in total")
    ap.add_argument("--gather", action="store_true",
                    help="cfg4: gather the images on rank 0 inside the timed region")
    return ap.parse_args()


def cpu_baseline(net, pose, intr, n_rays, threads):
    """Time the CPU oracle on `n_rays` rays of the same view, same T/t."""
    from oracle import field as ofield
    from oracle import rays as orays
    from oracle import renderer as oren
    torch.set_num_threads(threads)
    fld = ofield.OracleField(bound=4.0, num_semantic_classes=N_CLASSES,
                             seed=None)
    fld.grid_params = net.encoder.params.detach().cpu()
    fld.sigma_params = net.sigma_net.params.detach().cpu()
    fld.color_params = net.color_net.params.detach().cpu()
    fld.sem_params = net.semantics_net.params.detach().cpu()
    o, d, nrm = orays.pixel_rays(pose[None].cpu(), intr, H, W)
    g = torch.Generator().manual_seed(5)
    sel = torch.randperm(H * W, generator=g)[:n_rays]
    o, d, nrm = o[:, sel], d[:, sel], nrm[:, sel]
    u = torch.rand(n_rays, T_FINE, generator=g)
    aabb = torch.tensor([-4.0, -4, -4, 4, 4, 4])
    times = []
    with torch.no_grad():
        for it in range(3):   # >= 3 repeats, median (~8 s per pass)
            t0 = time.perf_counter()
            ref = oren.run(fld, o, d, nrm, aabb, num_steps=T_COARSE,
                           upsample_steps=T_FINE, u=u)
            times.append(time.perf_counter() - t0)
        best = sorted(times)[1]
        # the same rays / uniforms through the HIP path: parity + matched PSNR
        dev = net.encoder.params.device
        got = net.render(o.to(dev), d.to(dev), nrm.to(dev), num_steps=T_COARSE,
                         upsample_steps=T_FINE, rng_u=u.to(dev))
    parity = {k: float((got[k].cpu() - ref[k]).abs().max())
              for k in ("image", "depth", "semantics")}
    return n_rays / best, times, parity, ref, got, (o, d)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start
    the N ranks as a CHILD `python -m torch.distributed.run` and relay its
    stdout / exit code.  The ranks are CHILD processes and the parent only
    waits: a process that has initialised the GPU must never exec or be
    replaced, and ``torch.cuda.device_count()`` may initialise HIP here (on
    ROCm without amdsmi it falls back to hipGetDeviceCount) -- harmless,
    because nothing is exec'ed from this process.  The reference's DDP site: scripts/train_joint.py:137-142
    (Lightning spawns the ranks there)."""
    import socket
    import subprocess
    backend = os.environ.get("UCSA_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        raise SystemExit(
            f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s). One rank "
            "per GPU over RCCL needs that many; UCSA_BENCH_BACKEND=gloo runs all "
            "ranks on cuda:0 (code-path check on a 1-GPU box, not a measurement)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["UCSA_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    _tick("self-launch: " + " ".join(cmd[1:]))
    # stdout through a FILE, not a pipe: helper processes the ranks leave
    # behind for a while would keep an inherited pipe open past torchrun's exit
    import tempfile
    with tempfile.TemporaryFile("w+") as fo:
        proc = subprocess.Popen(cmd, env=env, stdout=fo, stdin=subprocess.DEVNULL)
        rc = proc.wait()
        fo.seek(0)
        sys.stdout.write(fo.read())   # rank 0's JSON line (stderr passed through)
        sys.stdout.flush()
    return rc


def dist_record(dist, world, rank, backend, dev, args):
    """What actually ran: world size as torch.distributed sees it, backend,
    the device of every rank.  Fails loudly when the process group has a
    different number of ranks than --gpus."""
    launcher = os.environ.get("UCSA_BENCH_LAUNCHER", "torchrun" if world > 1 else "none")
    if dist is None:
        return {"world_size": 1, "backend": None, "launcher": launcher,
                "devices": [f"{dev} ({torch.cuda.get_device_name(dev)})"]}
    ws = dist.get_world_size()
    if ws != args.gpus:
        raise SystemExit(f"process group has {ws} ranks, --gpus {args.gpus}")
    mine = (rank, str(dev), torch.cuda.get_device_name(dev))
    devs = [None] * ws
    dist.all_gather_object(devs, mine)
    if backend == "nccl" and len({d for _, d, _ in devs}) != ws:
        raise SystemExit(f"RCCL ranks share a device: {devs}")
    devs = [f"rank {r}: {d} ({n})" for r, d, n in devs]
    rec = {"world_size": ws, "backend": dist.get_backend() + (" (RCCL)" if backend == "nccl" else ""),
           "launcher": launcher, "devices": devs}
    if ws == 1:
        rec["forced_world_1"] = True     # UCSA_FORCE_DIST=1
    return rec


def init_distributed(args):
    """(dist module or None, world, rank, device, backend).  One rank per GPU
    under torchrun; at N = 1 a process group exists only under
    UCSA_FORCE_DIST=1 (world-size-1 RCCL group: every distributed branch runs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # test hook: UCSA_BENCH_BACKEND=gloo runs all ranks on cuda:0 (the 1-GPU
    # dev box) to exercise the N>1 code path; the driver's real runs use RCCL
    backend = os.environ.get("UCSA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or forced_dist():
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            s.close()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # create the communicator now (RCCL prints its banner then) and push the
        # banner out of libc's buffer on EVERY rank: nothing but rank 0's JSON
        # line may reach the job's stdout after this
        dist.barrier()
        common.flush_c_stdio()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    return dist, world, rank, dev, backend


# the encoder launches of one density pass (round 5): the coarse samples stay in
# image order, the fine samples are encoded in per-tile depth order
ENC_PASS_KERNELS = {
    "coarse": ("k_hashgrid_encode_tiled", "k_hashgrid_encode_tiled_ml"),
    "fine": ("k_hashgrid_encode_sorted", "k_hashgrid_encode_sorted_ml"),
}


# round 6: a density pass of the shipped render path (both passes the same two kernels)
DENSITY_PASS_KERNELS = ("k_hashgrid_encode_sorted", "k_density_sorted")


def encoder_roofline(st, chunk, pretrain_steps):
    """`roofline` of the dominant stage, the hash-grid encoder of one density
    pass: algorithmic gather bytes (L x 8 corners x F x 4 B = 1024 B
    per sample x the samples of one pass) / the pass's duration, event-timed live
    in this process (stage_times) / the 8 TB/s HBM line.  Since round 5 a pass is
    TWO launches back to back -- levels 9-15 one level per grid row, levels 0-8 in
    the several-levels kernel (ENC_PASS_KERNELS; the fine pass on the depth-ordered
    samples, after the per-tile sort `sort_f`) -- `launch_ms` is their sum, the
    mean of the coarse and the fine pass; the rocprofv3 summary
    (profiles/r05_bench_kernel_trace.txt) lists the kernels singly.  `traffic`
    and the binding unit come from the committed PMC passes (rocprofv3 cannot run
    inside this process), only when collected on the same parameter state and
    launch size."""
    samples = chunk * T_COARSE
    enc_bytes = samples * 16 * 8 * 2 * 4
    enc_ms = 0.5 * (st["encode_c"] + st["encode_f"])
    enc_gbs = enc_bytes / (enc_ms * 1e-3) / 1e9
    if "density_c" in st:
        # round 6: what ships is the DENSITY pass -- k_hashgrid_encode_sorted (levels
        # 12-15) + k_density_sorted (levels 0-11 encoded inside the sigma MLP): the
        # encoder of levels 0-11 is no kernel of its own any more.  Algorithmic bytes of
        # density() per sample: the 1024 B of gathers + 68 B out (h row, sigma) + 4 B in
        # (depth); the feature round trip between the two launches is not algorithmic.
        pass_bytes = samples * (1024 + 68 + 4)
        pass_ms = 0.5 * (st["density_c"] + st["density_f"])
        pass_gbs = pass_bytes / (pass_ms * 1e-3) / 1e9
        r = {"kernel": "density pass = 2 launches: k_hashgrid_encode_sorted (hash-grid levels 12-15, one "
                       "level per grid row) + k_density_sorted (levels 0-11 encoded inside the sigma MLP, "
                       "wave-private LDS tile); both passes on per-tile depth-ordered samples",
             "bound": "hbm", "achieved": pass_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": pass_gbs / HBM_PEAK_GBS, "traffic": None, "launch_ms": pass_ms,
             "algorithmic_bytes_per_launch": pass_bytes,
             "algorithmic_bytes_per_sample": {"gathers_16_levels": 1024, "h_row_and_sigma_out": 68,
                                              "depth_in": 4},
             "pass_ms": {"coarse": st["density_c"], "fine": st["density_f"],
                         "coarse_sort": st.get("order_c", 0.0), "fine_sort": st.get("sort_f", 0.0)},
             "encoder_only_unfused": {
                 "what": "round 5's figure, kept for continuity: the two encoder launches of a pass "
                         "(all 16 levels written to HBM) without the sigma MLP, 1024 B per sample",
                 "launch_ms": enc_ms, "achieved": enc_gbs, "frac": enc_gbs / HBM_PEAK_GBS,
                 "pass_ms": {"coarse": st["encode_c"], "fine": st["encode_f"]}},
             "timing": "HIP events on the launch stream around ucsa_density_sorted (its two launches), "
                       "mean of coarse+fine pass, 5 iterations",
             "note": "achieved = ALGORITHMIC bytes of density() / pass time: a nominal figure against "
                     "the HBM line, NOT HBM utilisation.  The table slice a launch phase works on is "
                     "L2/MALL resident; what binds the levels-12-15 launch is the L2 -> L1 fill of one "
                     "128-B line per 8-byte corner pair (binding_resource)"}
    else:
        r = {"kernel": "hash-grid encoder, one density pass = 2 launches: k_hashgrid_encode_tiled + "
                       "_tiled_ml (coarse) / k_hashgrid_encode_sorted + _sorted_ml (fine, depth-ordered)",
             "bound": "hbm", "achieved": enc_gbs,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": enc_gbs / HBM_PEAK_GBS,
             "traffic": None, "launch_ms": enc_ms, "algorithmic_bytes_per_launch": enc_bytes,
             "pass_ms": {"coarse": st["encode_c"], "fine": st["encode_f"],
                         "fine_sort": st.get("sort_f", 0.0)},
             "timing": "HIP events on the launch stream around the two encoder launches of a pass, "
                       "mean of coarse+fine pass, 5 iterations",
             "note": "achieved = ALGORITHMIC gather bytes (1024 B/sample) / pass time: a "
                     "nominal figure against the HBM line, NOT HBM utilisation "
                     "(hbm_utilisation is: measured traffic / time / peak).  The table slice a "
                     "launch phase works on is L2/MALL resident; what binds the fine levels is the "
                     "L2 -> L1 fill of one 128-B line per 8-byte corner pair (binding_resource)"}
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_JSON)))
        eb = json.load(open(os.path.join(ROOT, ENC_BINDING_JSON)))
    except OSError as e:
        # (only while a round's PMC passes are being re-collected: tools/
        # refresh_profiles.sh runs this very program under rocprofv3)
        print(f"[bench] WARNING: committed PMC passes missing ({e}): roofline.traffic and "
              "binding_resource are not reported", file=sys.stderr)
        return r
    except ValueError as e:
        raise SystemExit(f"bench.py: {PMC_JSON} / {ENC_BINDING_JSON} malformed: {e!r}")
    # ... on launches of THIS size: 256 threads per 1024 samples and level
    per_level = chunk * T_COARSE // 4
    shipped = "density_c" in st
    names = (list(DENSITY_PASS_KERNELS) if shipped
             else [k for ks in ENC_PASS_KERNELS.values() for k in ks])
    same_launch = all(
        "fetch_bytes" in pmc.get(k, {}) and int(pmc[k].get("grid_threads", -1)) % per_level == 0
        for k in names)
    if int(pmc.get("pretrain_steps", -1)) == int(pretrain_steps) and same_launch:
        # (shipped path: both passes launch the same two kernels at the same size -- the
        # per-dispatch averages ARE the mean over the passes; round 5's four kernels: half
        # the sum)
        tr = (1.0 if shipped else 0.5) * sum(pmc[k]["fetch_bytes"] + pmc[k]["write_bytes"]
                                             for k in names)
        r["traffic"] = tr
        r["traffic_source"] = PMC_JSON + " (mean over the two passes of the sum of their launches)"
        r["traffic_by_kernel"] = {k: pmc[k]["fetch_bytes"] + pmc[k]["write_bytes"] for k in names}
        r["hbm_utilisation"] = tr / (r["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    l2 = 0.5 * (eb["coarse"]["l2_request_frac_of_34500"] + eb["fine"]["l2_request_frac_of_34500"])
    keys = ("launch_us", "tcp_accesses_per_clock_per_cu", "l1_hit_rate", "l2_latency_cycles",
            "misses_in_flight_per_tcp", "tcp_pending_stall_frac", "valu_issue_frac",
            "l2_request_frac_of_34500")
    fine_top = eb["fine"].get("kernels", {}).get("k_hashgrid_encode_sorted", {})
    r["binding_resource"] = {
        "resource": "L2 -> L1 line fills of the gathers (128-B lines, ~34.5 TB/s): the per-level "
                    "launch of the finest levels (12-15 since round 6; these counters: round 5's "
                    "levels-9-15 launches, the same kernel); TCP look-ups 1 per clock and CU next",
        "achieved": l2, "peak": 1.0, "unit": "fraction of the L2 request rate", "frac": l2,
        "fine_levels_kernel": {k: fine_top[k] for k in keys if k in fine_top},
        "fine_pass": {k: eb["fine"][k] for k in keys if k in eb["fine"]},
        "coarse_pass": {k: eb["coarse"][k] for k in keys if k in eb["coarse"]},
        "source": ENC_BINDING_JSON + " (tools/encode_pmc.sh)"}
    return r


def composite_pmc(roof, mode, pretrain_steps):
    """PMC figures of the colour / semantics stage into its roofline object."""
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_JSON)))
    except (OSError, ValueError):
        return roof
    if int(pmc.get("pretrain_steps", -1)) != int(pretrain_steps):
        return roof
    kn = {"fp32": "k_composite", "bf16x3": "k_shade16_x3", "f16x2": "k_shade16_h2",
          "fp16": "k_shade16_f16"}[mode]
    k = pmc.get(kn, {})
    if "fetch_bytes" in k:
        tr = k["fetch_bytes"] + k["write_bytes"]
        if kn.startswith("k_shade16") and "fetch_bytes" in pmc.get("k_weights_compact", {}):
            tr += pmc["k_weights_compact"]["fetch_bytes"] + pmc["k_weights_compact"]["write_bytes"]
        roof["traffic"] = tr
        roof["traffic_source"] = PMC_JSON
        roof["hbm_utilisation"] = tr / (roof["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if "mfma_busy_frac" in k:
        roof["mfma_pipe_busy_frac"] = k["mfma_busy_frac"]
    if "valu_issue_frac" in k:
        roof["valu_issue_frac"] = k["valu_issue_frac"]
    return roof


def main():
    if os.environ.get("UCSA_BENCH_WATCHDOG"):   # debugging aid: stacks of a stuck rank
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UCSA_BENCH_WATCHDOG"]), exit=True)
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    dist, world, rank, dev, backend = init_distributed(args)
    common.DIST_RECORD = dist_record(dist, world, rank, backend, dev, args)

    from ucsa_neural_rendering_amd import ops
    if args.mode == "cfg3":
        args.backbone = args.backbone or "resnet50"
        from tools.bench_legs.cfg3 import main_cfg3
        return main_cfg3(args, dev, dist, world, rank, backend)
    if args.mode == "cfg5":
        from tools.bench_legs.cfg5 import main_cfg5
        return main_cfg5(args, dev, dist, world, rank, backend)
    prelog = {}
    _tick("imports, device")
    net, scene_ds = build_field(dev, train_steps=args.pretrain_steps, log=prelog)
    _tick("field pre-trained")
    if dist:
        # the pre-training is not bit-reproducible (float atomics in the grid
        # backward): all ranks render / train rank 0's field
        from ucsa_neural_rendering_amd import dist as udist_
        udist_.broadcast_parameters_(net)
    net.hip_ray_chunk = 65536
    comm_dtype = {"fp32": None, "fp16": torch.float16,
                  "bf16": torch.bfloat16}[args.grad_comm_dtype]
    if args.mode == "train":
        from tools.bench_legs.train import main_train
        return main_train(args, net, scene_ds, dev, dist, world, rank, backend,
                          prelog, comm_dtype)
    if args.mode == "cfg4":
        from tools.bench_legs.cfg4 import main_cfg4
        return main_cfg4(args, net, scene_ds, dev, dist, world, rank, backend, prelog)
    from tools.bench_legs.render_modes import (composite_roofline, render_mode_legs,
                                               stage_times)
    intr = (0.89 * W, 0.89 * W, W / 2.0, H / 2.0)
    n_views = args.steps + args.warmup
    from ucsa_neural_rendering_amd.dataset.synthetic_scene import _slerp_loop_poses
    poses = _slerp_loop_poses(n_views * world, seed=999)[rank::world].to(dev)
    # inputs resident in HBM before the timed region: rays of every view this
    # rank renders, and the uniforms for the inverse-CDF resampling
    rays = [ops.get_rays(poses[i:i + 1], intr, H, W) for i in range(n_views)]
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    u = torch.rand(H * W, T_FINE, device=dev, generator=g)

    def step(i):
        o, d, nrm = rays[i]
        with torch.no_grad():
            return net.render(o, d, nrm, staged=True, perturb=False,
                              num_steps=T_COARSE, upsample_steps=T_FINE,
                              rng_u=u, image_width=W)

    net.precision = args.nerf_precision
    for i in range(args.warmup):
        step(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev, backend)
    rays_total = world * args.steps * H * W
    value = rays_total / elapsed

    result = None
    if rank == 0:
        assert torch.isfinite(out["image"]).all()
        # the chunk render() really launches (whole 8-row bands of the image,
        # balanced over the pipelined call's chunks)
        chunk = net.infer_chunk(H * W, W)[0]
        o, d, nrm = rays[0]
        chunk_in = (o[0, :chunk].contiguous(), d[0, :chunk].contiguous(),
                    nrm[0, :chunk, 0].contiguous(), u[:chunk])
        st, rho = stage_times(net, *chunk_in, image_width=W, mode=args.nerf_precision)
        samples = chunk * T_COARSE
        masked = rho * chunk * (T_COARSE + T_FINE)
        mlp_flop = masked * (12544 + 7040)
        mlp_tf = mlp_flop / (st["composite"] * 1e-3) / 1e12
        sig_tf = samples * 6144 / (0.5 * (st["sigma_c"] + st["sigma_f"]) * 1e-3) / 1e12
        ms_step = elapsed / args.steps * 1e3
        S = T_COARSE + T_FINE
        # whole view, SURVEY 8d / BASELINE.md 2.4 definitions
        step_bytes_fp32 = H * W * (204 + S * 1024)        # this build: fp32 table
        step_bytes_fp16 = H * W * (204 + S * 512)         # the definition's fp16 table
        step_flop = H * W * S * (6144 + rho * 19584)
        roof_enc = encoder_roofline(st, chunk, args.pretrain_steps)
        roof_cmp = composite_pmc(composite_roofline(args.nerf_precision, mlp_tf, sig_tf,
                                                    st["composite"]),
                                 args.nerf_precision, args.pretrain_steps)
        enc_share = st["encode_c"] + st["encode_f"]
        result = {
            "metric": "rays/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: Semantic-NeRF render, one 640x480 view per step per GPU, "
                            "192 samples/ray (96 coarse + 96 fine), hash grid L=16 F=2 "
                            "T=2^19, MLP width 64, 40 classes",
                "rays_per_step_per_gpu": H * W,
                "ray_chunk": chunk,
                "parameter_state": "tcnn-style init (seed 123) + %d Adam steps on the "
                                   "synthetic box-room scene (SURVEY 8d)"
                                   % prelog.get("pretrain_steps", 0),
                "pretrain": prelog,
                "masked_fraction_rho": rho,
                "sharding": "views round-robin over ranks, no data-path collective",
                "timed_region": "net.render() of one view per step (rows a2-a10), rays and "
                                "uniforms resident in HBM; get_rays (a1, ~10 us/view) runs "
                                "before it (--mode cfg4 times it per view)",
                "mlp_arithmetic": MLP_ARITHMETIC[args.nerf_precision],
            },
            # "roofline" = the kernel with the largest share of the step
            "roofline": roof_enc if enc_share >= st["composite"] else roof_cmp,
            "roofline_encode": roof_enc,
            "roofline_composite": roof_cmp,
            "roofline_step": {
                "what": "one 640x480 view end to end (ms_per_step)",
                "hbm": {"algorithmic_bytes_fp32_table": step_bytes_fp32,
                        "achieved_gbs_fp32_table": step_bytes_fp32 / ms_step / 1e6,
                        "frac_fp32_table": step_bytes_fp32 / ms_step / 1e6 / HBM_PEAK_GBS,
                        "algorithmic_bytes_fp16_table_definition": step_bytes_fp16,
                        "frac_fp16_table_definition":
                            step_bytes_fp16 / ms_step / 1e6 / HBM_PEAK_GBS,
                        "peak_gbs": HBM_PEAK_GBS},
                "mfma": {"algorithmic_flop": step_flop,
                         "achieved_tflops": step_flop / ms_step / 1e9,
                         "frac_of_fp16_dense_peak":
                             step_flop / ms_step / 1e9 / F16_MFMA_PEAK_TF,
                         "frac_of_fp32_mfma_peak":
                             step_flop / ms_step / 1e9 / F32_MFMA_PEAK_TF,
                         "peak_fp16_dense_tflops": F16_MFMA_PEAK_TF},
            },
            "stage_ms_per_chunk": st,
        }
        # quality of the timed renders: last view vs the analytic ground truth
        t_hit, gt_rgb, gt_lab = scene_ds.room.cast(rays[n_views - 1][0][0], rays[n_views - 1][1][0])
        mse = torch.mean((out["image"][0] - gt_rgb) ** 2)
        _, pred_lab = ops.semantic_postproc(out["semantics"][0], want_normalised=False)
        from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
        meter = SemanticsMeter(N_CLASSES)
        meter.update(pred_lab, gt_lab)
        result["quality"] = {"psnr_db": float(-10 * torch.log10(mse)),
                             "miou": meter.measure()[0]}
        _tick("headline measured")
        # The side measurements below are single-GPU figures: at N > 1 the
        # other ranks would only wait for rank 0, so they run at N = 1 only.
        extras = world == 1
        if extras and args.detail:
            render_mode_legs(result, net, step, chunk_in, args, world, mlp_flop, samples,
                             step_flop, dev)
            _tick("render modes measured")
            from tools.bench_legs.march import march_option
            try:
                result["march_option"] = march_option(net, scene_ds, rays, n_views, out,
                                                      dev, args)
            except Exception as e:  # the headline line must survive, loudly
                import traceback
                traceback.print_exc(file=sys.stderr)
                result["march_option"] = {"error": repr(e), "failed": True}
            _tick("marcher option done")
        if extras and not args.no_train_bench:
            from tools.bench_legs.train import train_legs
            train_legs(result, net, scene_ds, dev, all_modes=args.detail)
            _tick("training legs done")
        if extras and args.detail:
            from tools.bench_legs.seg import seg_throughput
            result["seg"] = seg_throughput(dev, find=args.seg_find)
            _tick("DeepLab leg done")
        result["tuning_tables_matched"] = common.tuning_tables_matched(dev)
        if extras and not args.no_cpu_baseline:
            threads = effective_cores()
            v, dt, parity, ref, got, (co, cd) = cpu_baseline(
                net, poses[0], intr, args.cpu_rays, threads)
            _, gt_c, _ = scene_ds.room.cast(co[0].to(dev), cd[0].to(dev))
            psnr = lambda img: float(-10 * torch.log10(torch.mean((img - gt_c.to(img.device)) ** 2)))
            result["cpu_vs_gpu"] = {
                "max_abs_diff": parity,
                "psnr_db_cpu": psnr(ref["image"][0]),
                "psnr_db_gpu": psnr(got["image"][0]),
            }
            result["cpu_baseline"] = {
                "value": v,
                "unit": "rays/s",
                "cores": threads,
                "kind": "port",
                "sample": f"{args.cpu_rays} random rays of view 0, same "
                          f"T={T_COARSE}/t={T_FINE}; median of 3 passes ("
                          + ", ".join(f"{x:.1f}" for x in dt) + " s)",
                "pass_seconds": dt,
                # (inside this object, next to the sample it was measured on -- not a
                # headline figure: VERDICT r5 hygiene)
                "gpu_over_cpu": value / v,
            }
            _tick("CPU baseline done")
    if dist:
        # the data-parallel training step with its gradient collectives: all
        # ranks take part; rank 0 reports it in the detail file
        from tools.bench_legs.train import dp_train_leg
        tr = dp_train_leg(net, scene_ds, dev, dist, world, rank, backend,
                          steps=min(args.steps, 20), replicated=args.replicated_adam,
                          comm_dtype=comm_dtype)
        if rank == 0:
            result["train_dp"] = tr
    common.finish(dist, rank, result)


if __name__ == "__main__":
    main()
