#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace / --pmc) as a
small text table: per-kernel calls, total/avg/min/max duration, share, and
(when counters were collected) per-kernel counter sums per dispatch.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x.txt
"""
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    import os
    by_grid = os.environ.get("BY_GRID") and "grid_x" in cols
    sel = f"{name_col}, start, end" + (", grid_x" if by_grid else ", 0")
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    # TAIL_FRAC=0.2: only the kernels launched in the last 20 % of the traced
    # wall time (e.g. the steady state after a warm-up phase)
    tail = float(os.environ.get("TAIL_FRAC", "0") or 0)
    if tail > 0 and rows:
        t0, t1 = rows[0][1], rows[-1][2]
        cut = t1 - tail * (t1 - t0)
        rows = [r for r in rows if r[1] >= cut]
    agg = defaultdict(list)
    for n, s, e, gx in rows:
        agg[f"[{gx}] {n}" if by_grid else n].append(e - s)
    tot = sum(sum(v) for v in agg.values()) or 1
    print(f"# rocprofv3 kernel-trace summary of {path}")
    if rows:
        span = rows[-1][2] - rows[0][1]
        print(f"# {len(rows)} dispatches over {span/1e6:.1f} ms, GPU busy "
              f"{tot/1e6:.1f} ms ({100*tot/max(span,1):.0f} %)"
              + (f", last {tail:.2f} of the trace" if tail > 0 else ""))
    print(f"# {'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} "
          f"{'min_us':>10s} {'max_us':>10s} {'share%':>7s}")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = n if len(n) <= 60 else n[:57] + "..."
        print(f"  {short:60s} {len(v):7d} {sum(v)/1e6:10.3f} "
              f"{sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} "
              f"{max(v)/1e3:10.2f} {100*sum(v)/tot:7.2f}")
    try:
        pm = c.execute("select * from counters_collection limit 1").fetchall()
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except sqlite3.Error:
        pm, ccols = [], []
    if pm:
        kn = "kernel_name" if "kernel_name" in ccols else "name"
        gs = "grid_size" if "grid_size" in ccols else ("grid_size_x" if "grid_size_x" in ccols else None)
        grp = f"{kn}, {gs}" if gs else kn
        sel = f"{kn}, {gs if gs else 0}"
        q = (f"select {sel}, counter_name, sum(value), count(distinct dispatch_id) "
             f"from counters_collection group by {grp}, counter_name")
        print("\n# counters: kernel, grid size, counter, sum over dispatches, "
              "dispatches, per dispatch")
        for n, gsz, cn, v, nd in c.execute(q):
            if not n.startswith(("k_", "void k_")):
                continue
            short = n if len(n) <= 44 else n[:41] + "..."
            print(f"  {short:44s} {gsz:10d} {cn:26s} {v:18.0f} {nd:5d} {v/max(nd,1):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
