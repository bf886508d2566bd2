#!/usr/bin/env python3
"""Sum the PMC counters of a rocprofv3 rocpd database over the LAST `TAIL_FRAC`
of its dispatches (by dispatch id = launch order) and report them per
occurrence of a marker kernel (one per step) -- e.g. HBM bytes per training
step from a `--pmc FETCH_SIZE` pass of tools/profile_train.py.

    TAIL_FRAC=0.1 MARKER=k_nerf_loss_grad python tools/pmc_window.py x_results.db
"""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select dispatch_id, {kn}, counter_name, value "
                     "from counters_collection").fetchall()
    ids = sorted({r[0] for r in rows})
    tail = float(os.environ.get("TAIL_FRAC", "0.1"))
    cut = ids[int(len(ids) * (1.0 - tail))]
    marker = os.environ.get("MARKER", "k_nerf_loss_grad")
    # whole steps only: the window runs from just AFTER the first marker at or
    # beyond the tail cut to the LAST marker (inclusive) -- a partial first
    # step would otherwise inflate the per-step figures
    mark_ids = sorted({r[0] for r in rows if marker in r[1] and r[0] >= cut})
    if len(mark_ids) >= 2:
        lo, hi = mark_ids[0], mark_ids[-1]
        steps = len(mark_ids) - 1
    else:                       # fewer than two markers: the raw tail, one "step"
        lo, hi, steps = cut - 1, ids[-1], max(1, len(mark_ids))
    tot = defaultdict(float)
    per_kernel = defaultdict(lambda: defaultdict(float))
    for did, name, cn, v in rows:
        if did <= lo or did > hi:
            continue
        tot[cn] += v
        per_kernel[name.split("(")[0][:60]][cn] += v
    out = {"db": os.path.basename(path), "tail_frac": tail, "marker": marker,
           "steps_in_window": steps,
           "per_step": {k: v / steps for k, v in tot.items()},
           "per_step_by_kernel": {
               k: {cn: v / steps for cn, v in d.items()}
               for k, d in sorted(per_kernel.items(),
                                  key=lambda kv: -sum(kv[1].values()))[:14]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
