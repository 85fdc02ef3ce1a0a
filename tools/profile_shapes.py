"""Per (kernel, grid) launch statistics from a rocprofv3 rocpd sqlite database: which SHAPES of a kernel class take the time.
Usage: python tools/profile_shapes.py db out.md [name-substring ...]"""
import sqlite3
import sys

db, out = sys.argv[1:3]
subs = sys.argv[3:]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
def pick(*names):
    for n in names:
        if n in cols:
            return n
    raise SystemExit(f"no column of {names} in view kernels: {cols}")
name = pick("name", "kernel_name")
gx, gy, gz = pick("grid_x", "grid_size_x"), pick("grid_y", "grid_size_y"), pick("grid_z", "grid_size_z")
wx = pick("workgroup_x", "workgroup_size_x")
st, en = pick("start"), pick("end")
q = f"select {name}, {gx}/{wx}, {gy}, {gz}, count(*), avg({en}-{st})/1e3, sum({en}-{st})/1e6 from kernels group by 1,2,3,4 order by 7 desc"
rows = [r for r in c.execute(q) if not subs or any(s in r[0] for s in subs)]
with open(out, "w") as f:
    f.write("| kernel | workgroups (x,y,z) | calls | avg us | total ms |\n|---|---|---|---|---|\n")
    for n, x, y, z, k, a, t in rows[:80]:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        f.write(f"| `{n[:60]}` | {x},{y},{z} | {k} | {a:.1f} | {t:.1f} |\n")
print(open(out).read())
