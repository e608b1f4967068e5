"""Prints the kernels of one steady-state frame of a rocprofv3 kernel trace (rocpd database)
in start order: start offset from the frame's first kernel, duration, queue, name; then the
frame period.  A frame starts at every k_vs_geometry launch.
Usage: python scripts/frame_timeline.py results.db [frame index from the end, default 5]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = cur.execute(sel).fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("wr::", "")
starts = [i for i, r in enumerate(rows) if "k_vs_geometry" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a, b = starts[-back - 1], starts[-back]
# a frame's first kernel may be a clear that precedes k_vs_geometry
while a > 0 and rows[a - 1][1] > rows[a][1] - 30000 and "k_clear_words" in rows[a - 1][0]:
    a -= 1
while b > 0 and rows[b - 1][1] > rows[b][1] - 30000 and "k_clear_words" in rows[b - 1][0]:
    b -= 1
t0 = rows[a][1]
last_end = t0
for n, s, e, q in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q}  {short(n)}")
    last_end = max(last_end, e)
print(f"frame period {(rows[b][1] - t0) / 1e3:.1f} us, last kernel ends at {(last_end - t0) / 1e3:.1f} us")
periods = [(rows[starts[i + 1]][1] - rows[starts[i]][1]) / 1e3 for i in range(len(starts) // 2, len(starts) - 1)]
print(f"mean period over the second half of the trace: {sum(periods) / len(periods):.1f} us")
