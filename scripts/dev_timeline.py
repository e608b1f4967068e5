"""Developer helper: overlap statistics of a rocprofv3 kernel trace (rocpd database)."""
import re, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
# keep the timed region: last 60% of march launches
march = [(s, e) for n, s, e in rows if "k_fog_march<false" in n]
t0, t1 = march[15][0], march[95][1]
ev = []
for n, s, e in rows:
    if e < t0 or s > t1: continue
    kind = "march" if "k_fog_march" in n else "other"
    ev.append((max(s, t0), 1, kind)); ev.append((min(e, t1), -1, kind))
ev.sort()
cnt = {"march": 0, "other": 0}; last = t0
acc = {}
for t, d, k in ev:
    key = (min(cnt["march"], 4), cnt["other"] > 0)
    acc[key] = acc.get(key, 0) + (t - last); last = t
    cnt[k] += d
span = t1 - t0
print("span ms", span / 1e6, "marches in span", sum(1 for s, e in march if s >= t0 and e <= t1))
for key in sorted(acc): print(f"marches running={key[0]} others running={key[1]}: {100 * acc[key] / span:5.1f} %")
