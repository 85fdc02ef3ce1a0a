#!/bin/bash
# Effective shader clock of the Winograd GEMM kernels by phase-skipping mode (GRBM_GUI_ACTIVE / kernel duration; MI355X_MICROARCH.md DVFS).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/x6pclk
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p --output-format csv -- python $R/tools/x6p_pp_breakdown.py > $OUT/run.log 2>&1
python - <<PY
import csv, glob, collections
cc = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
kt = glob.glob("$OUT/p/**/*kernel_trace.csv", recursive=True)
print(cc, kt)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
rows = []
for f in cc:
    rd = csv.DictReader(open(f))
    print(rd.fieldnames)
    for r in rd:
        if "gemm_x6p" not in r["Kernel_Name"]: continue
        d = dur.get(r["Dispatch_Id"])
        if not d: continue
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-30:], d[0], float(r["Counter_Value"])))
rows.sort()
# groups of 11 launches (1 warm + 10 timed) per (n, pp, dbg) in the order of tools/x6p_pp_breakdown.py
labels = [(n, pp, dbg) for n in (8, 16) for pp in (0, 1) for dbg in (0, 32, 192, 224)]
for i, lab in enumerate(labels):
    g = rows[i * 11 + 1:(i + 1) * 11]
    if not g: break
    ns = sum(x[2] for x in g) / len(g); cy = sum(x[3] for x in g) / len(g)
    print("n=%d pp=%d dbg=%3d %-28s avg %7.1f us  GRBM_GUI_ACTIVE %10.0f  -> %6.0f MHz (if the counter is per-chip cycles)" % (lab + (g[0][1], ns / 1e3, cy, cy / ns * 1e3)))
PY
