"""Timeline of the LAST train step(s) in a rocprofv3 rocpd database traced with the default (multi-stream) execution: how much of the
wall time the GPU runs >= 1 kernel, the average number of kernels in flight, the per-queue busy time, and the idle gaps.  Answers whether
a step is paced by kernel durations, by dependent-dispatch gaps or by the host.   Usage: python tools/timeline_summary.py db out.md [last_ms]"""
import sqlite3
import sys

db, out = sys.argv[1:3]
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
def pick(*names):
    for n in names:
        if n in cols:
            return n
    return None
name, st, en = pick("name", "kernel_name"), pick("start"), pick("end")
qid = pick("queue_id", "queue", "stream_id", "stream")
rows = list(c.execute(f"select {name}, {st}, {en}, {qid if qid else 0} from kernels order by {st}"))
t_end = max(r[2] for r in rows)
if last_ms > 0:
    rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
t0, t1 = rows[0][1], max(r[2] for r in rows)
wall = (t1 - t0) / 1e6
ev = []
for _, s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = area = 0.0
depth, prev = 0, t0
gaps = []
for t, d in ev:
    if depth > 0:
        busy += t - prev
        area += (t - prev) * depth
    elif t > prev:
        gaps.append(t - prev)
    depth += d
    prev = t
per_q = {}
for _, s, e, q in rows:
    a = per_q.setdefault(q, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e6
ksum = sum((e - s) for _, s, e, _ in rows) / 1e6
with open(out, "w") as f:
    f.write(f"window {wall:.2f} ms, {len(rows)} dispatches; sum of kernel durations {ksum:.2f} ms\n")
    f.write(f"GPU runs >= 1 kernel for {busy / 1e6:.2f} ms ({100 * busy / 1e6 / wall:.1f} % of the window); kernels in flight while busy: {area / max(busy, 1):.2f}\n")
    f.write(f"idle gaps: {len(gaps)} totalling {sum(gaps) / 1e6:.2f} ms; gaps > 20 us: {sum(1 for g in gaps if g > 20e3)} totalling {sum(g for g in gaps if g > 20e3) / 1e6:.2f} ms\n\n")
    f.write("| queue | dispatches | kernel time ms | share of window |\n|---|---|---|---|\n")
    for q, (n, t) in sorted(per_q.items(), key=lambda kv: -kv[1][1]):
        f.write(f"| {q} | {n} | {t:.2f} | {100 * t / wall:.1f} % |\n")
print(open(out).read())
