"""`bench.py --mode cfg5`: BASELINE cfg5 -- the continual loop of
``run_scripts/multi_step.sh`` / ``scripts/cl_deeplab.py`` (reference
scripts/cl_deeplab.py:11-22,53-91) AT ITS WORKLOAD: ten synthetic rooms (seeds
0-9) of 240x320 frames written in the reference's ScanNet layout, stage i =
NeRF-only epochs on scene i, then joint NeRF + DeepLab epochs on scene i plus
the replay of the i earlier scenes (``cl.replay_buffer_size`` 100,
cfg/exp/multi_step/cl_base.yml), test passes, the predict pass that writes the
pseudo-label PNGs the later stages replay, checkpoint chaining.  4096 rays x
(256+256) samples per NeRF step, batch size and backbone from the YAML.

Timed: the whole ``cl_deeplab.main`` call (every stage: data-module setup,
PNG reads, training, test / predict passes, checkpoint writes) -- the scene
export (data generation) runs before the clock.  Reported: `value` = NeRF rays
(trained + rendered) per second over the loop, joint steps/s, per-stage wall
time and metrics, and the FINAL stage's mIoU / PSNR (what BASELINE.json asks:
"throughput + final mIoU").  The number of epochs is a flag: the reference
runs 10 + 10 per stage, hours of wall clock; the default here is small and is
stated in the output."""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time

import torch

from .common import *  # noqa: F401,F403
from .common import _tick, finish, max_over_ranks


def pretrain_seg(exp, dev, steps, path, palette_seed, B=8):
    """The reference starts the loop from a DeepLab pre-trained on ScanNet-25k
    (cfg/exp/multi_step/cl_base.yml `checkpoint_load`; its pseudo-labels are
    what the NeRF's semantic head learns, joint_train_lightning_net.py:171-176).
    No checkpoint can be fetched here and a randomly initialised network labels
    everything with one class (final NeRF mIoU 0.0, measured), so the stand-in
    is pre-trained in-harness, BEFORE the clock starts: `steps` Adam steps
    (lr 1e-4, batch 8, CE-on-softmax like the reference) on eight OTHER
    synthetic rooms (seeds 100-107) that share the class -> colour table of the
    loop's rooms (`synthetic.palette_seed`), written in the reference's
    Lightning checkpoint layout.  Returns its mIoU on a ninth room."""
    from ucsa_neural_rendering_amd import losses as ul, ops
    from ucsa_neural_rendering_amd.dataset import SyntheticSceneDataset
    from ucsa_neural_rendering_amd.network import DeepLabV3
    from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter
    torch.manual_seed(5)
    m = DeepLabV3(exp["model"]).to(dev).train().to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    rooms = [SyntheticSceneDataset(100 + i, n_views=12, H=240, W=320, n_classes=N_CLASSES,
                                   device=dev, palette_seed=palette_seed) for i in range(8)]
    frames = [(r[i]["img"], r[i]["label"]) for r in rooms for i in range(12)]
    g = torch.Generator().manual_seed(6)
    for it in range(steps):
        idx = torch.randint(0, len(frames), (B,), generator=g).tolist()
        x = torch.stack([frames[i][0] for i in idx]).contiguous(memory_format=torch.channels_last)
        y = torch.stack([frames[i][1] for i in idx])
        loss = ul.seg_loss(m(x)["out"].float().contiguous(), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
    m.eval()
    val = SyntheticSceneDataset(200, n_views=8, H=240, W=320, n_classes=N_CLASSES, device=dev,
                                palette_seed=palette_seed)
    meter = SemanticsMeter(N_CLASSES)
    with torch.no_grad():
        for i in range(8):
            pred = ops.seg_tail(m(val[i]["img"][None])["out"].float().contiguous(), None,
                                want_prob=False)["argmax"]
            meter.update(pred, val[i]["label"][None])
    sd = {"_model." + k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.save({"state_dict": sd}, path)
    return float(meter.measure()[0]), float(loss.detach())


def main_cfg5(args, dev, dist, world, rank, backend):
    from scripts import cl_deeplab
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd.lightning import joint_train_lightning_net as jl
    from ucsa_neural_rendering_amd.nerf import renderer_semantics as rs
    from ucsa_neural_rendering_amd.utils import load_yaml

    exp = load_yaml(os.path.join(ROOT, "cfg/exp/multi_step/cl_base.yml"))
    n_views = int(round(args.frames / 0.8))           # 80 % train / 20 % val split
    palette_seed = 7 if args.pretrain_seg_steps > 0 else None
    exp["synthetic"].update(n_views=n_views, H=240, W=320, palette_seed=palette_seed)
    if args.backbone:
        exp["model"]["backbone"] = args.backbone
    if args.seg_amp:
        exp["model"]["amp"] = args.seg_amp
    exp["nerf"]["precision"] = args.nerf_precision
    if os.environ.get("UCSA_CFG5_PREFETCH"):       # `trainer: {prefetch: N}` (lightning/trainer.py)
        exp["trainer"]["prefetch"] = int(os.environ["UCSA_CFG5_PREFETCH"])
    exp["trainer"]["cudnn_benchmark"] = (True if args.seg_find else
                                         False if args.no_seg_find else None)
    if exp["trainer"]["cudnn_benchmark"] is None:
        del exp["trainer"]["cudnn_benchmark"]
    root = tempfile.mkdtemp(prefix="ucsa_cfg5_")
    env = {"results": os.path.join(root, "experiments"), "scannet": os.path.join(root, "scans")}
    scenes = cl_deeplab.SCENE_ORDER[:args.scenes]
    t0 = time.perf_counter()
    cl_deeplab.ensure_synthetic_scenes(exp, env, scenes)     # rank 0 writes, the rest wait
    export_s = time.perf_counter() - t0
    _tick(f"cfg5: {len(scenes)} synthetic rooms exported ({export_s:.1f} s, not timed)")
    pre = None
    if args.pretrain_seg_steps > 0:
        ck = os.path.join(root, "pretrain_deeplab.ckpt")
        if rank == 0:
            t1 = time.perf_counter()
            miou0, loss0 = pretrain_seg(exp, dev, args.pretrain_seg_steps, ck, palette_seed)
            pre = {"steps": args.pretrain_seg_steps, "seconds_not_timed": time.perf_counter() - t1,
                   "miou_on_an_unseen_room": miou0, "final_loss": loss0}
            _tick(f"cfg5: DeepLab stand-in pre-trained, mIoU {miou0:.3f} on an unseen room "
                  f"({pre['seconds_not_timed']:.1f} s, not timed)")
        if dist:
            dist.barrier()
        exp["general"]["checkpoint_load"] = ck

    # counters: rays through render() with / without grad, training steps
    count = {"rays_trained": 0, "rays_rendered": 0, "nerf_steps": 0, "joint_steps": 0,
             "seg_images": 0}
    orig_render = rs.SemanticNeRFRenderer.render

    def render_spy(self, rays_o, *a, **k):
        n = rays_o.shape[0] * rays_o.shape[1] if rays_o.dim() == 3 else rays_o.shape[0]
        count["rays_trained" if torch.is_grad_enabled() and self.training else
              "rays_rendered"] += int(n)
        return orig_render(self, rays_o, *a, **k)

    orig_step = jl.JointTrainLightningNet.training_step

    def step_spy(self, batch, batch_idx):
        count["joint_steps" if self.joint_train else "nerf_steps"] += 1
        return orig_step(self, batch, batch_idx)

    stage_marks = []
    orig_train = cl_deeplab.train

    def train_spy(cfg, env_, a, b, cargs):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        before = dict(count)
        r = orig_train(cfg, env_, a, b, cargs)
        torch.cuda.synchronize()
        stage_marks.append({"stage": cfg["general"]["name"].split("/")[-1],
                            "scenes": len(cfg["scenes"]),
                            "seconds": time.perf_counter() - t1,
                            **{k: count[k] - before[k] for k in count}})
        return r

    rs.SemanticNeRFRenderer.render = render_spy
    jl.JointTrainLightningNet.training_step = step_spy
    cl_deeplab.train = train_spy
    argv = ["--exp_name", "cfg5", "--scenes", str(len(scenes)),
            "--nerf_train_epoch", str(args.nerf_epochs),
            "--joint_train_epoch", str(args.joint_epochs), "--seed", "123"]
    try:
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prof_to = os.environ.get("UCSA_CFG5_PROFILE")   # host-side profile of the loop
        if prof_to and rank == 0:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            results = pr.runcall(cl_deeplab.main, argv, exp=exp, env=env)
            with open(prof_to, "w") as fh:
                st = pstats.Stats(pr, stream=fh)
                st.sort_stats("cumulative").print_stats(70)
                st.sort_stats("tottime").print_stats(40)
        else:
            results = cl_deeplab.main(argv, exp=exp, env=env)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    finally:
        rs.SemanticNeRFRenderer.render = orig_render
        jl.JointTrainLightningNet.training_step = orig_step
        cl_deeplab.train = orig_train
    elapsed = max_over_ranks(elapsed, dist, dev, backend)
    # whole-job counts: every rank ran its shard of every loader
    tot = torch.tensor([count["rays_trained"], count["rays_rendered"], count["nerf_steps"],
                        count["joint_steps"]], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    if dist:
        dist.all_reduce(tot)
    rays_trained, rays_rendered, nerf_steps, joint_steps = [float(x) for x in tot]
    stages = []
    for mark, res in zip(stage_marks, results):
        row = dict(mark)
        for phase in ("test_after_nerf", "val", "test_after_joint"):
            for k, v in (res.get(phase) or {}).items():
                row[f"{phase}.{k}"] = v
        stages.append(row)
    last = results[-1]["test_after_joint"]
    final = {k: last.get(k) for k in ("test_nerf_mIoU", "test_nerf_PSNR")}
    # DeepLab on the new scene's validation frames (before its joint epochs: the
    # network as the previous stages left it; reference train_joint.py:143-146)
    for k, v in (results[-1].get("val") or {}).items():
        final["seg_" + k] = v
    result = {
        "metric": "rays/sec", "value": (rays_trained + rays_rendered) / elapsed,
        "unit": "rays/s", "n_gpus": world, "steps": int(joint_steps + nerf_steps),
        "warmup": 0, "ms_per_step": elapsed / max(1.0, joint_steps + nerf_steps) * 1e3 * world,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"cfg5: continual loop over {len(scenes)} synthetic rooms (seeds 0-"
                        f"{len(scenes) - 1}), 240x320 frames, {args.frames} training frames per "
                        f"scene, 4096 rays x (256+256) per NeRF step, batch "
                        f"{exp['data_module']['batch_size']}, replay buffer "
                        f"{exp['cl']['replay_buffer_size']}, DeepLabV3-{exp['model']['backbone']}, "
                        f"{args.nerf_epochs} NeRF-only + {args.joint_epochs} joint epochs per stage "
                        "(reference: 10 + 10)",
            "mode": "cfg5", "timed_region": "scripts/cl_deeplab.main: all stages incl. data-module "
                                            "setup, PNG I/O, test / predict passes, checkpoints; "
                                            "scene export excluded",
            "rays_per_step_per_gpu": 4096,
            "nerf_epochs": args.nerf_epochs, "joint_epochs": args.joint_epochs,
            "frames_per_scene": args.frames, "seg_precision": args.seg_amp or "fp32",
            "nerf_render_nets": args.nerf_precision,
            "trainer_prefetch": int(exp["trainer"].get("prefetch", 0) or 0),
        },
        "seg_pretraining": pre,
        "quality": {"final_stage": final,
                    "note": "test pass after the last stage's joint epochs (reference "
                            "train_joint.py: trainer_joint.test on the NeRF train loader)"},
        "throughput": {"total_s": elapsed, "joint_steps": joint_steps, "nerf_only_steps": nerf_steps,
                       "joint_steps_per_s": joint_steps / elapsed,
                       "rays_trained": rays_trained, "rays_rendered": rays_rendered,
                       "rays_trained_per_s": rays_trained / elapsed,
                       "export_s_not_timed": export_s},
        "stages": stages,
    }
    try:      # decoded-frame cache of the dataset (round 5): how many PNG decodes it saved
        from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import decode_cache
        dc = decode_cache()
        result["decode_cache"] = {"hits": dc.hits, "misses": dc.misses, "mb": dc.used / 2 ** 20,
                                  "budget_mb": dc.budget / 2 ** 20}
    except Exception as e:      # never let bookkeeping sink a benchmark record
        result["decode_cache"] = {"error": repr(e)}
    import shutil
    if rank == 0:
        shutil.rmtree(root, ignore_errors=True)
    finish(dist, rank, result)
