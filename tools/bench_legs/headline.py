"""The ONE line bench.py prints on stdout.

The driver parses the last stdout line of `python bench.py ...` as JSON and
keeps ~12 fields of it; round 3's line had grown to 22.7 KB (per-kernel PMC
dictionaries, prose notes, seven side legs) and did not parse, so the round's
headline number was not recorded.  Contract now:

* ``headline(result)`` keeps exactly the keys below, shortens free text, and
  the serialised line is <= ``MAX_LINE`` bytes (asserted; a CPU test pins it);
* everything else a run measured goes to ``bench_detail.json`` (next to
  bench.py, and under ``gpurun_out/`` when that directory exists) and to
  stderr -- never to stdout.

No torch import here: the CPU test builds a line from a canned result."""
from __future__ import annotations

import json
import os
import sys

MAX_LINE = 4096

TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "mode", "rays_per_step_per_gpu", "ray_chunk", "timed_region",
               "masked_fraction_rho", "parameter_state", "mlp_arithmetic", "sharding")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                 "launch_ms", "algorithmic_bytes_per_launch", "algorithmic_flop_per_launch",
                 "hbm_utilisation", "timing")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "gpu_over_cpu")


def _short(s, n):
    if isinstance(s, str) and len(s) > n:
        return s[:n - 1] + "…"
    return s


def _num(x):
    """Floats to 6 significant digits: the line is a record, not a dump."""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _num(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v) for v in x]
    return x


def headline(result: dict, text: int = 300) -> dict:
    """The compact record of `result` (any bench mode)."""
    out = {k: result.get(k) for k in TOP_KEYS}
    cfg = result.get("config") or {}
    out["config"] = {k: _short(cfg[k], text) for k in CONFIG_KEYS if k in cfg}
    roof = result.get("roofline")
    if roof:
        r = {k: roof[k] for k in ROOFLINE_KEYS if k in roof}
        r["kernel"] = _short(r.get("kernel"), 200)
        r.setdefault("traffic", None)
        b = roof.get("binding_resource")
        if b:
            r["binding_resource"] = {"resource": _short(b.get("resource"), 120),
                                     "frac": b.get("frac")}
        out["roofline"] = r
    cpu = result.get("cpu_baseline")
    if cpu:
        c = {k: cpu[k] for k in CPU_KEYS if k in cpu}
        c["sample"] = _short(c.get("sample"), 200)
        out["cpu_baseline"] = c
    for k in ("quality", "tuning_tables_matched"):
        if k in result:
            out[k] = result[k]
    # BASELINE.json's metric is "rays/sec (train+render)": the NeRF training
    # step measured in the same run (tools/bench_legs/train.py)
    tr = result.get("train")
    if isinstance(tr, dict) and "ms_per_step" in tr:
        t = {"ms_per_step": tr["ms_per_step"], "rays_per_s": tr.get("rays_per_s"),
             "workload": _short(tr.get("workload"), min(text, 160))}
        hbm = (tr.get("roofline") or {}).get("hbm") or {}
        if "frac" in hbm:
            t["hbm_frac_algorithmic"] = hbm["frac"]
        out["train"] = t
    d = result.get("distributed")
    if d:
        dd = {k: d.get(k) for k in ("world_size", "backend", "launcher", "forced_world_1")
              if k in d}
        devs = d.get("devices") or []
        dd["devices"] = [_short(x, 60) for x in devs[:8]]
        out["distributed"] = dd
    if "detail" in result:
        out["detail"] = result["detail"]
    return _num(out)


def headline_line(result: dict) -> str:
    """Serialised compact record, guaranteed <= MAX_LINE bytes: free text is
    shortened further until it fits (numbers are never dropped)."""
    for text in (300, 160, 80, 40):
        line = json.dumps(headline(result, text), separators=(",", ":"))
        if len(line.encode()) <= MAX_LINE:
            return line
    raise AssertionError(f"bench headline does not fit {MAX_LINE} bytes: {len(line)}")


def emit(result: dict, root: str) -> str:
    """Write the full result to bench_detail.json (+ gpurun_out/ when present),
    echo it to stderr, print the compact line as the LAST stdout line."""
    paths = [os.path.join(root, "bench_detail.json")]
    scratch = os.path.join(root, "gpurun_out")
    if os.path.isdir(scratch):
        paths.append(os.path.join(scratch, "bench_detail.json"))
    blob = json.dumps(result, indent=1, default=str)
    written = []
    for p in paths:
        try:
            with open(p, "w") as fh:
                fh.write(blob + "\n")
            written.append(os.path.relpath(p, root))
        except OSError as e:   # read-only checkout: the line must still appear
            print(f"[bench] cannot write {p}: {e}", file=sys.stderr)
    result = dict(result, detail=written[0] if written else None)
    print("[bench] full result:\n" + blob, file=sys.stderr, flush=True)
    line = headline_line(result)
    sys.stdout.flush()
    print(line, flush=True)
    return line
