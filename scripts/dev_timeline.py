"""Developer helper: overlap statistics of a rocprofv3 kernel trace (rocpd database).
Prints, for the steady-state part of the trace, how much of the time 0, 1, 2 ... kernels ran
together and each kernel's share of the machine time when every instant is split evenly among
the kernels running at it."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_shade<false"
frames = [(s, e) for n, s, e in rows if marker in n]
lo, hi = int(len(frames) * 0.3), int(len(frames) * 0.9)
t0, t1 = frames[lo][0], frames[hi][0]
nframes = hi - lo
ev = []
for i, (n, s, e) in enumerate(rows):
    if e <= t0 or s >= t1: continue
    short = re.sub(r"\(.*", "", n).replace("void ", "").replace("wr::", "")
    ev.append((max(s, t0), 1, short)); ev.append((min(e, t1), -1, short))
ev.sort(key=lambda x: (x[0], x[1]))
running = {}
last = t0
level = {}
share = {}
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        n = sum(running.values())
        level[n] = level.get(n, 0) + dt
        if n:
            for kk, c in running.items():
                if c: share[kk] = share.get(kk, 0) + dt * c / n
    last = t
    running[k] = running.get(k, 0) + d
span = t1 - t0
print(f"span {span / 1e6:.3f} ms, {nframes} frames, {span / nframes / 1e3:.1f} us per frame")
for n in sorted(level): print(f"  {n} kernels running: {100 * level[n] / span:5.1f} %")
print("machine-time share per frame (us), instants split evenly among running kernels:")
for k, v in sorted(share.items(), key=lambda kv: -kv[1]): print(f"  {k:28s} {v / nframes / 1e3:8.1f}")
