"""The RCCL branches, executed on the one GPU there is:
``UCSA_FORCE_DIST=1`` makes ``dist.init_from_env`` / ``bench.py`` /
``scripts/train_joint.py`` create a world-size-1 ``nccl`` process group and
take the distributed code path -- sharded Adam's reduce_scatter_tensor /
all_gather_into_tensor on device tensors, the coalesced MLP all-reduce, the
collective found-inf flag, metric reductions, ``cfg4 --gather``, buffer
broadcasts, all_gather_object -- with results bit-equal to the
non-distributed run.  The gloo two-rank tests (tests/test_dist_gloo.py,
tests/test_gpu_dist.py) cover N = 2 semantics through CPU-staged branches;
this file covers the branches only RCCL takes.  Reference DDP site:
scripts/train_joint.py:137-142.  ``-m gpu``."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "UCSA_BENCH_BACKEND")}
    env.update(UCSA_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def _run(cmd, tmp_path, env, timeout=900):
    out, err = tmp_path / "out.txt", tmp_path / "err.txt"
    with open(out, "w") as fo, open(err, "w") as fe:
        rc = subprocess.Popen(cmd, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL,
                              env=env, cwd=ROOT).wait(timeout=timeout)
    text = out.read_text()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert rc == 0 and lines, (rc, err.read_text()[-3000:])
    if os.path.basename(cmd[1]) == "bench.py":
        # the record is the LAST stdout line even though RCCL prints a banner with
        # C stdio (which a pipe / file buffers until flushed: common.flush_c_stdio)
        assert text.strip().splitlines()[-1] == lines[-1], text[-600:]
    return json.loads(lines[-1]), err.read_text()


def test_every_collective_branch_runs_under_nccl_and_equals_the_plain_path(tmp_path):
    res, _ = _run([sys.executable, os.path.join(ROOT, "tests", "scripts", "rccl_world1.py")],
                  tmp_path, _env())
    assert res["backend"] == "nccl" and res["active"] is True
    # 3 training steps, every gradient payload: ShardedHipAdam over RCCL ==
    # HipAdam on the same (payload-rounded) gradients, bit for bit, every step
    for k in ("fp32", "fp16", "bf16"):
        assert res["sharded_equals_plain"][k] == [True, True, True], k
    assert res["last_comm_bytes"] > 2 * 13_000_000 * 4        # reduce-scatter + all-gather
    assert res["overflow_step_skipped"] is True and res["scale_after_overflow"] == 512.0
    for k in ("allreduce_sum", "average_grads", "broadcast", "confusion", "gather_rows",
              "rs_ag_roundtrip", "cfg4_view_equal"):
        assert res[k] is True, k
    assert res["global_count"] == 7.0 and res["global_mean_scale"] == [1.0, 1.0]
    assert res["all_gather_ints"] == [5] and res["allreduce_max"] == 3.0
    assert res["all_gather_object"] == "rank0"


def test_bench_default_line_under_a_forced_rccl_group(tmp_path):
    """`python bench.py` with UCSA_FORCE_DIST=1: the same compact line, a
    world-1 RCCL group recorded, and the data-parallel training leg (sharded
    Adam over RCCL) in the detail file."""
    res, err = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup",
                     "1", "--pretrain-steps", "30", "--no-cpu-baseline", "--no-train-bench"],
                    tmp_path, _env())
    d = res["distributed"]
    assert d["world_size"] == 1 and d["backend"].startswith("nccl") and d["forced_world_1"] is True
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["roofline"]["bound"] == "hbm"
    full = json.load(open(os.path.join(ROOT, res["detail"])))
    dp = full["train_dp"]
    assert dp["collective_backend"] == "nccl" and dp["collective_ranks"] == 1
    assert dp["optimizer"].startswith("ShardedHipAdam") and dp["replicas_identical"] is True
    assert dp["comm_bytes_per_step_per_rank"] > 2 * 13_000_000 * 4
    assert dp["reduce_scatter_allgather_ms"] > 0 and dp["allreduce_ms"] > 0


def test_train_joint_entrypoint_under_a_forced_rccl_group(tmp_path):
    """scripts/train_joint.train with a world-1 RCCL group: DistributedSampler
    loaders, parameter broadcast, ShardedHipAdam + CollectiveGradScaler, DeepLab
    gradient all-reduce, confusion-matrix / PSNR reductions, BN buffer
    broadcasts -- all on device tensors; finite metrics come back."""
    code = r'''
import argparse, json, os, sys, torch
sys.path.insert(0, %r)
from scripts import train_joint as tj
import torch.distributed as dist
root = %r
exp = {"general": {"name": "joint_train/rccl1", "clean_up_folder_if_exists": True, "checkpoint_load": ""},
       "model": {"pretrained": False, "pretrained_backbone": False, "num_classes": 40, "backbone": "resnet50"},
       "optimizer": {"lr_seg": 1e-5, "lr_nerf": 1e-2, "name": "Adam"},
       "trainer": {"load_from_checkpoint": False, "cudnn_benchmark": False},
       "data_module": {"batch_size": 2}, "scenes": ["scene0000_00"],
       "synthetic": {"n_views": 6, "H": 48, "W": 64},
       "nerf": {"n_rays": 512, "num_steps": 32, "upsample_steps": 32, "sharded_optimizer": True},
       "nerf_seed": 1}
env = {"results": os.path.join(root, "experiments"), "scannet": root}
cfgp = os.path.join(root, "exp.yml"); open(cfgp, "w").write("x: 1\n")
args = argparse.Namespace(exp_name="t", fix_nerf=False, seed=123, nerf_train_epoch=1,
                          joint_train_epoch=1, limit_batches=None)
seen = {}
from ucsa_neural_rendering_amd.nerf import optim
orig = optim.ShardedHipAdam.step
def spy(self, *a, **k):
    seen["sharded_steps"] = seen.get("sharded_steps", 0) + 1
    seen["backend"] = dist.get_backend()
    return orig(self, *a, **k)
optim.ShardedHipAdam.step = spy
res = tj.train(exp, env, cfgp, cfgp, args)
print(json.dumps({"results": res, "seen": seen, "comm": True}))
''' % (ROOT, str(tmp_path))
    res, _ = _run([sys.executable, "-c", code], tmp_path, _env())
    assert res["seen"]["backend"] == "nccl" and res["seen"]["sharded_steps"] >= 3
    for k in ("test_after_nerf", "test_after_joint"):
        v = res["results"][k]
        assert v["test_nerf_PSNR"] == v["test_nerf_PSNR"] and 0.0 <= v["test_nerf_mIoU"] <= 1.0
