"""Summarises a rocprofv3 rocpd database (kernel trace) into a per-kernel table:
calls, total/average/min/max duration in microseconds, share of GPU kernel time.
Usage: python scripts/rocpd_summary.py results.db [> profiles/NAME.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values()) or 1
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:60]:60s} {a[0]:6d} {a[1]/1e3:12.1f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:10.2f} {a[3]/1e3:10.2f} {100*a[1]/total:6.2f}")
