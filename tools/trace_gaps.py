#!/usr/bin/env python3
"""Idle gaps of a rocprofv3 kernel trace (rocpd sqlite): the union of kernel
intervals over the last TAIL_FRAC of the trace, the largest gaps between busy
periods with the kernels on either side, and the gap time by (before -> after).
    TAIL_FRAC=0.4 python tools/trace_gaps.py x_results.db
    MARKER=multi_tensor_apply STEPS=4 python tools/trace_gaps.py x_results.db   # whole steps"""
import os
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, start, end from kernels order by start").fetchall()
tail = float(os.environ.get("TAIL_FRAC", "0.4"))
t0, t1 = rows[0][1], rows[-1][2]
marker, nsteps = os.environ.get("MARKER"), int(os.environ.get("STEPS", "4"))
if marker:   # window = the STEPS steps between the last STEPS + 1 dispatches of MARKER (+ SKIP_LAST)
    skip = int(os.environ.get("SKIP_LAST", "0"))
    ms = [i for i, r in enumerate(rows) if marker in r[0]]
    ms = ms[:len(ms) - skip] if skip else ms
    lo, hi = ms[-(nsteps + 1)], ms[-1]
    rows = rows[lo:hi + 1]
    print(f"# window: {nsteps} steps between dispatches of '{marker}': "
          f"{(rows[-1][1] - rows[0][1]) / 1e6 / nsteps:.2f} ms per step")
else:
    rows = [r for r in rows if r[1] >= t1 - tail * (t1 - t0)]
busy_end, last = rows[0][2], rows[0][0]
gaps = []
busy = 0
seg_start = rows[0][1]
for n, s, e in rows[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, last, n, busy_end))
        busy += busy_end - seg_start
        seg_start = s
    if e > busy_end:
        busy_end, last = e, n
busy += busy_end - seg_start
span = rows[-1][2] - rows[0][1]
idle = sum(g[0] for g in gaps)
print(f"# {len(rows)} dispatches over {span/1e6:.1f} ms: busy (union) {busy/1e6:.1f} ms, "
      f"idle {idle/1e6:.1f} ms ({100*idle/span:.0f} %) in {len(gaps)} gaps")
short = lambda n: n.replace("void ", "")[:44]
by = defaultdict(lambda: [0, 0])
for g, a, b, _ in gaps:
    by[(short(a), short(b))][0] += g
    by[(short(a), short(b))][1] += 1
print("# idle time by (kernel before -> kernel after), top 25")
for (a, b), (g, k) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g/1e6:8.2f} ms in {k:4d} gaps   {a:44s} -> {b}")
