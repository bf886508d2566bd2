"""bench.py's launcher logic where no GPU is needed: `--gpus N` without a
launcher refuses, loudly and before any GPU work, when the node has fewer
than N devices for RCCL (here: none)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_devices_refuses_before_touching_a_gpu():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UCSA_BENCH_BACKEND")}
    env["HIP_VISIBLE_DEVICES"] = ""      # also on a GPU box: no devices for this check
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "this node shows 0 GPU(s)" in p.stderr and p.stdout.strip() == ""
