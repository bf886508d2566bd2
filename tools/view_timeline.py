#!/usr/bin/env python3
"""Timeline of the LAST pipelined view of a rocprofv3 kernel trace of tools/view_time.py: every dispatch with
its queue, start and end (ms from the view's first kernel) -- which of the two internal streams of
ucsa_render_view is busy when.
    rocprofv3 --kernel-trace -d /tmp/pv -o v -- python3 tools/view_time.py ; python tools/view_timeline.py <db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
sel = f"select {name}, start, end" + (f", {q}" if q else ", 0") + " from kernels order by start"
rows = c.execute(sel).fetchall()
# the last view: from the last k_near_far that follows a gap back to ... take the last 5 k_near_far dispatches
nf = [i for i, r in enumerate(rows) if "k_near_far" in r[0]]
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lo = nf[-n_chunks]
rows = rows[lo:]
t0 = rows[0][1]
short = lambda n: n.replace("void ", "").split("(")[0][:34]
queues = sorted({r[3] for r in rows})
print(f"# last view: {len(rows)} dispatches, {(max(r[2] for r in rows) - t0) / 1e6:.3f} ms; queues {queues}")
busy = {qq: 0 for qq in queues}
for n, s, e, qq in rows:
    busy[qq] += e - s
    print(f"  q{queues.index(qq)}  {(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  ({(e - s) / 1e3:7.1f} us)  {short(n)}")
print("# busy per queue (ms):", {f"q{queues.index(k)}": round(v / 1e6, 3) for k, v in busy.items()})
