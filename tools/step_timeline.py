#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: every dispatch between
two consecutive occurrences of a marker kernel (default k_nerf_loss_grad, one
per NeRF training step), in start order -- start offset, duration, idle gap
since the latest end of anything before it, queue -- and the step's totals
(busy time, idle gaps, number of launches).  Shows where a chain of dependent
launches loses time BETWEEN kernels.

    MARKER=k_nerf_loss_grad STEP=-2 python tools/step_timeline.py x_results.db"""
import os
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = c.execute(f"select {name}, start, end, {q} from kernels order by start").fetchall()
    marker = os.environ.get("MARKER", "k_nerf_loss_grad")
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    k = int(os.environ.get("STEP", "-2"))
    a, b = marks[k], marks[k + 1]
    step = rows[a:b]
    t0 = step[0][1]
    latest_end = step[0][1]
    busy = gaps = 0.0
    print(f"# step = dispatches {a}..{b} of {path}: {len(step)} launches, "
          f"{(rows[b][1] - t0) / 1e3:.1f} us from marker to marker")
    print(f"# {'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'queue':>6s}  kernel")
    merged_end = t0
    for n, s, e, qid in step:
        gap = max(0.0, (s - latest_end) / 1e3)
        print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {str(qid):>6s}  {n[:90]}")
        gaps += gap
        busy += max(0, e - max(s, merged_end)) / 1e3
        merged_end = max(merged_end, e)
        latest_end = max(latest_end, e)
    print(f"# union of kernel time {busy:.1f} us, idle gaps {gaps:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
