"""Two ranks (gloo rendezvous, both on cuda:0 -- the dev box has one GPU)
drive the HIP training path on different ray shards: after the SUM all-reduce
of the four parameter gradients every rank must hold bit-identical parameters,
and sharded rendering must reproduce the single-process image.  The driver's
multi-GPU runs use the same code over RCCL.  ``-m gpu``."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd import losses as ul
    from ucsa_neural_rendering_amd.nerf.optim import HipAdam
    from tests.util import hip_network_from_oracle, lively_oracle_field, make_rays
    torch.cuda.set_device(0)
    udist.init_from_env("gloo")
    try:
        net = hip_network_from_oracle(lively_oracle_field()).train()
        opt = HipAdam([{"params": list(net.encoder.parameters())},
                       {"params": list(net.sigma_net.parameters()) +
                        list(net.color_net.parameters()) +
                        list(net.semantics_net.parameters()), "weight_decay": 1e-6}],
                      lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        N, T, t = 512, 32, 32
        o, d, n = make_rays(N, 4)
        g = torch.Generator().manual_seed(4)
        gt_rgb = torch.rand(1, N, 3, generator=g)
        gt_depth = torch.rand(1, N, generator=g) * 3 + 0.5
        labels = torch.randint(0, 40, (1, N), generator=g)
        u = torch.rand(N, t, generator=g)
        tr = torch.rand(N, T, generator=g)
        b, e = udist.shard_range(N, rank, world)
        sl = slice(b, e)
        losses = []
        for it in range(3):
            out = net.render(o[None, sl].cuda(), d[None, sl].cuda(), n[None, sl].cuda(),
                             perturb=True, num_steps=T, upsample_steps=t,
                             rng_t=tr[sl].cuda(), rng_u=u[sl].cuda())
            lc, ls, ld = ul.nerf_losses(out["image"], out["semantics"], out["depth"],
                                        gt_rgb[:, sl].cuda(), labels[:, sl].cuda(),
                                        gt_depth[:, sl].cuda(), 1.0)
            loss = ul.nerf_total_loss(lc, ls, ld)
            opt.zero_grad()
            loss.backward()
            ps = list(net.parameters())
            udist.allreduce_grads_(ps)
            for p in ps:
                p.grad.div_(world)
            opt.step()
            losses.append(float(loss.detach()))
        # sharded inference render of all rays, gathered on rank 0
        net.eval()
        with torch.no_grad():
            part = net.render(o[None, sl].cuda(), d[None, sl].cuda(), n[None, sl].cuda(),
                              num_steps=T, upsample_steps=t, rng_u=u[sl].cuda())
            full = net.render(o[None].cuda(), d[None].cuda(), n[None].cuda(),
                              num_steps=T, upsample_steps=t, rng_u=u.cuda())
        img = udist.gather_rows(part["image"][0].cpu(), [udist.shard_range(N, r, world)[1] -
                                                         udist.shard_range(N, r, world)[0]
                                                         for r in range(world)])
        ret[rank] = dict(
            sums=[float(p.detach().double().sum()) for p in net.parameters()],
            head=net.sigma_net.params.detach().cpu()[:16].clone(),
            losses=losses,
            gathered_equal=None if img is None else bool(torch.equal(img, full["image"][0].cpu())))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_keeps_replicas_identical_and_sharded_render_matches():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["sums"] == r1["sums"]
    assert torch.equal(r0["head"], r1["head"])
    assert r0["gathered_equal"] is True
    assert all(l == l for l in r0["losses"] + r1["losses"])  # finite


# ---- the LightningModule under torch.distributed --------
def _module_worker(rank, world, port, root, sharded, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import argparse
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd.lightning import joint_train_lightning_net as jl
    from scripts import train_joint as tj
    torch.cuda.set_device(0)
    udist.init_from_env("gloo")
    try:
        drawn = []
        orig = jl.JointTrainLightningNet.get_rays_train

        def spy(self, batch, bs, N=None):
            out = orig(self, batch, bs, N)
            if len(drawn) < 4:
                drawn.append((str(batch["current_index"][bs]), out[3][0, :64].cpu().clone()))
            return out

        jl.JointTrainLightningNet.get_rays_train = spy
        keep = {}
        orig_end = jl.JointTrainLightningNet.on_predict_epoch_end

        def grab(self):   # NeRF parameters at the end of the run
            keep["params"] = [p.detach().cpu().clone()
                              for p in self.nerf_model.parameters()]
            return orig_end(self)

        jl.JointTrainLightningNet.on_predict_epoch_end = grab
        orig_save = torch.save

        def save_spy(obj, path, *a, **k):
            keep["saved_by"] = rank
            return orig_save(obj, path, *a, **k)

        torch.save = save_spy
        exp = {
            "general": {"name": "joint_train/dist_tiny", "clean_up_folder_if_exists": True,
                        "checkpoint_load": ""},
            "model": {"pretrained": False, "pretrained_backbone": False,
                      "num_classes": 40, "backbone": "resnet50"},
            "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
            "trainer": {"load_from_checkpoint": False, "cudnn_benchmark": False},
            "data_module": {"batch_size": 2},
            "scenes": ["scene0000_00"],
            "synthetic": {"n_views": 10, "H": 48, "W": 64},
            "nerf": {"n_rays": 512, "num_steps": 32, "upsample_steps": 32,
                     "sharded_optimizer": sharded},
            "nerf_seed": 1,
        }
        env = {"results": os.path.join(root, "experiments"), "scannet": root}
        cfgp = os.path.join(root, "exp.yml")
        if rank == 0:
            open(cfgp, "w").write("x: 1\n")
        torch.distributed.barrier()
        args = argparse.Namespace(exp_name="t", fix_nerf=False, seed=123,
                                  nerf_train_epoch=2, joint_train_epoch=1,
                                  limit_batches=None)
        res = tj.train(exp, env, cfgp, cfgp, args)
        ret[rank] = dict(drawn=drawn, results=res, saved_by=keep.get("saved_by"),
                         params=keep.get("params"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [True, False])
def test_two_rank_train_joint_shards_frames_and_pixels(tmp_path, sharded):
    """scripts/train_joint.py under two ranks: the ranks train on DIFFERENT
    frames and draw DIFFERENT pixels (rank-offset seed, DistributedSampler),
    the NeRF replicas stay bit-identical after the run (sharded optimizer and
    replicated all-reduce alike), evaluation is split and its metrics reduced
    (both ranks report the same numbers), and only rank 0 saves the
    checkpoint."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_module_worker, args=(2, _free_port(), str(tmp_path), sharded, ret),
             nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    frames0 = [f for f, _ in r0["drawn"]]
    frames1 = [f for f, _ in r1["drawn"]]
    assert frames0 and frames1 and frames0[0] != frames1[0]       # different frames
    assert not torch.equal(r0["drawn"][0][1], r1["drawn"][0][1])   # different pixels
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                                   # replicas identical
    for k in ("test_after_nerf", "test_after_joint"):
        assert r0["results"][k]["test_nerf_PSNR"] == pytest.approx(
            r1["results"][k]["test_nerf_PSNR"], abs=1e-9)
        assert r0["results"][k]["test_nerf_mIoU"] == pytest.approx(
            r1["results"][k]["test_nerf_mIoU"], abs=1e-12)
    assert r0["saved_by"] == 0 and r1["saved_by"] is None


# ---- continual stage with replay under two ranks ----------
def _cl_worker(rank, world, port, root, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from ucsa_neural_rendering_amd import dist as udist
    from ucsa_neural_rendering_amd.lightning import joint_train_lightning_net as jl
    from scripts import cl_deeplab
    torch.cuda.set_device(0)
    udist.init_from_env("gloo")
    try:
        log = {"real": 0, "idle": 0, "per_step": [], "mixed": 0}
        o_real = jl.JointTrainLightningNet._nerf_update
        o_idle = jl.JointTrainLightningNet._nerf_update_idle
        o_step = jl.JointTrainLightningNet.training_step_joint
        o_end = jl.JointTrainLightningNet.on_predict_epoch_end
        keep = {}

        def real(self, *a, **k):
            log["real"] += 1
            return o_real(self, *a, **k)

        def idle(self, *a, **k):
            log["idle"] += 1
            return o_idle(self, *a, **k)

        def step(self, batch):
            before = log["real"] + log["idle"]
            old, new, _ = batch
            if old is not None and (new is None or new["img"].shape[0] < 2):
                log["mixed"] += 1
            r = o_step(self, batch)
            log["per_step"].append(log["real"] + log["idle"] - before)
            return r

        def grab(self):
            keep[self._exp["general"]["name"]] = [
                p.detach().cpu().clone() for p in self.nerf_model.parameters()]
            return o_end(self)

        jl.JointTrainLightningNet._nerf_update = real
        jl.JointTrainLightningNet._nerf_update_idle = idle
        jl.JointTrainLightningNet.training_step_joint = step
        jl.JointTrainLightningNet.on_predict_epoch_end = grab
        exp = {
            "general": {"name": "x", "clean_up_folder_if_exists": True,
                        "checkpoint_load": ""},
            "model": {"pretrained": False, "pretrained_backbone": False,
                      "num_classes": 40, "backbone": "resnet50"},
            "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
            "trainer": {"load_from_checkpoint": True, "resume_from_checkpoint": False,
                        "cudnn_benchmark": False},
            "data_module": {"batch_size": 2, "output_size": (48, 64)},
            "scenes": ["scene0000_00"],
            "cl": {"active": False, "use_novel_viewpoints": False,
                   "replay_buffer_size": 6},
            "synthetic": {"n_views": 6, "H": 48, "W": 64},
            "nerf": {"n_rays": 256, "num_steps": 16, "upsample_steps": 16,
                     "sharded_optimizer": True},
            "nerf_seed": 1,
        }
        env = {"results": os.path.join(root, "experiments"),
               "scannet": os.path.join(root, "scans")}
        res = cl_deeplab.main(["--exp_name", "cl", "--scenes", "2",
                               "--nerf_train_epoch", "1", "--joint_train_epoch", "2"],
                              exp=exp, env=env)
        last = sorted(keep)[-1]
        ret[rank] = dict(log=log, stages=len(res), params=keep[last])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_continual_stage_with_replay_keeps_collectives_aligned(tmp_path):
    """Stage 1 of the continual loop mixes replayed old-scene frames into the
    joint loader: a rank's batch then holds 0, 1 or 2 NEW frames, and each
    NeRF update contains collectives.  Every rank must run the same number of
    updates per step (idle ones with zero gradients), else the next
    collective pairs with the wrong one (hang / size mismatch).  The run
    finishes, both ranks made the same number of updates in every step, at
    least one step had ranks holding different numbers of new frames, and
    the replicas are bit-identical at the end."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cl_worker, args=(2, _free_port(), str(tmp_path), ret),
             nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["stages"] == r1["stages"] == 2
    assert r0["log"]["per_step"] == r1["log"]["per_step"]
    assert r0["log"]["idle"] + r1["log"]["idle"] > 0, (r0["log"], r1["log"])
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
