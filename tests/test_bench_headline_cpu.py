"""The one stdout line of bench.py (tools/bench_legs/headline.py): built from
a canned full result the size of round 3's (which did not parse in the
driver), it must stay <= 4096 bytes, survive a json round trip and keep the
fields the driver and the judge read."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.bench_legs import headline as hl  # noqa: E402


def _canned():
    big = {f"void k_kernel_{i}<{i}, HIP_vector_type<float, 2>>": 1.234567890123e8 * i
           for i in range(60)}
    note = "prose " * 200
    return {
        "metric": "rays/sec", "value": 14861234.56789, "unit": "rays/s", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 20.6712345, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg2: " + "x" * 500, "rays_per_step_per_gpu": 307200,
                   "ray_chunk": 61440, "timed_region": note, "pretrain": {"a": note},
                   "masked_fraction_rho": 0.9018521308898926, "mlp_arithmetic": note,
                   "parameter_state": note, "sharding": note},
        "roofline": {"kernel": "k_hashgrid_encode_tiled", "bound": "hbm",
                     "achieved": 5651.123456, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.7063904, "traffic": 1950000000, "launch_ms": 1.07,
                     "algorithmic_bytes_per_launch": 6039797760, "note": note,
                     "hbm_utilisation": 0.2279,
                     "binding_resource": {"resource": "TCP " + note, "frac": 0.785,
                                          "fine_pass": big, "coarse_pass": big}},
        "roofline_composite": {"note": note, "by_kernel": big},
        "train": {"ms_per_step": 3.4012345678, "rays_per_s": 1204280.123456,
                  "workload": "NeRF train step, 4096 rays x (256+256) samples " + note,
                  "ms_per_step_blocks": [3.4] * 20,
                  "roofline": {"hbm": {"frac": 0.1712345678, "algorithmic_bytes": 4.66e9},
                               "traffic": {"fetch_by_kernel_bytes": big,
                                           "write_by_kernel_bytes": big}}, "note": note},
        "seg": {m: {"note": note, "k": big} for m in ("fp32", "bf16", "g1", "g2")},
        "march_option": {"note": note, "fp32": big, "fp16": big},
        "cpu_baseline": {"value": 4948.19, "unit": "rays/s", "cores": 16, "kind": "port",
                         "sample": "32768 random rays " + note, "pass_seconds": [6.6, 6.7, 5.8],
                         "gpu_over_cpu": 3003.33},
        "quality": {"psnr_db": 27.1, "miou": 0.83},
        "tuning_tables_matched": {"miopen": True, "tunableop": False,
                                  "why": {"miopen": "x", "tunableop": "PT_VERSION differs"}},
        "distributed": {"world_size": 8, "backend": "nccl (RCCL)", "launcher": "torchrun",
                        "devices": [f"rank {r}: cuda:{r} (AMD Instinct MI355X)" for r in range(8)]},
    }


def test_headline_fits_and_round_trips():
    res = _canned()
    assert len(json.dumps(res)) > 20000          # the size that broke round 3
    line = hl.headline_line(res)
    assert len(line.encode()) <= hl.MAX_LINE == 4096 and "\n" not in line
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in back
    assert back["config"]["workload"].startswith("cfg2") and "pretrain" not in back["config"]
    r = back["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["traffic"] == 1950000000 and set(r["binding_resource"]) == {"resource", "frac"}
    c = back["cpu_baseline"]
    assert (c["kind"], c["cores"], c["unit"]) == ("port", 16, "rays/s") and c["value"] > 0
    # the GPU / CPU ratio sits INSIDE the baseline object, next to its sample size
    assert abs(c["gpu_over_cpu"] - 3003.33) < 1e-6 and "speedup_vs_cpu" not in back
    assert back["tuning_tables_matched"]["miopen"] is True
    assert back["distributed"]["world_size"] == 8 and len(back["distributed"]["devices"]) == 8
    # BASELINE.json's metric is train + render: the training step's figures are
    # in the line, its per-kernel dictionaries are not
    t = back["train"]
    assert set(t) == {"ms_per_step", "rays_per_s", "workload", "hbm_frac_algorithmic"}
    assert abs(t["ms_per_step"] - 3.40123) < 1e-4 and abs(t["rays_per_s"] - 1204280) < 1
    assert len(t["workload"]) <= 160
    for k in ("seg", "march_option", "roofline_composite"):
        assert k not in back                     # detail file only


def test_emit_prints_the_line_last_and_writes_the_detail_file(tmp_path):
    code = ("import sys, json; sys.path.insert(0, %r); "
            "from tools.bench_legs.headline import emit; "
            "from tests.test_bench_headline_cpu import _canned; "
            "print('noise before'); emit(_canned(), %r)" % (ROOT, str(tmp_path)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    rec = json.loads(last)
    assert len(last.encode()) <= 4096 and rec["detail"] == "bench_detail.json"
    full = json.load(open(tmp_path / "bench_detail.json"))
    assert "seg" in full and "train" in full and full["value"] == _canned()["value"]
    assert "full result" in p.stderr


def test_none_roofline_traffic_is_kept_as_null():
    res = _canned()
    del res["roofline"]["traffic"]
    assert json.loads(hl.headline_line(res))["roofline"]["traffic"] is None
