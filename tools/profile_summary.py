"""Summarise a rocprofv3 rocpd sqlite database (top kernels) as markdown.  Usage: python tools_profile_summary.py db title out.md"""
import sqlite3
import sys

db, title, out = sys.argv[1:4]
c = sqlite3.connect(db)
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# {title}\n\n")
    f.write(f"total kernel time {tot/1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches (rocprofv3 --kernel-trace --stats)\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for n, k, t, a, p in rows[:80]:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        f.write(f"| `{n[:90]}` | {k} | {t/1e3:.1f} | {a:.1f} | {p:.2f} |\n")
print(open(out).read())
