#!/bin/bash
# LDS bank-conflict share per kernel class of one training step (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; PMC pass with --kernel-trace only).
# Usage on the GPU box: bash tools/lds_conflicts.sh   ->  gpurun_out/lds_conflicts.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ldsc
rm -rf $OUT; mkdir -p $OUT
SS_DUAL_STREAM=0 SS_UNET_BRANCHES=0 SS_UNET_WGRAD_STREAM=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<PY > $R/gpurun_out/lds_conflicts.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void\s+", "", name); name = name.split("(")[0][:60]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name].add(r["Dispatch_Id"])
rows = []
for n, c in agg.items():
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        rows.append((c["SQ_LDS_IDX_ACTIVE"], n, len(cnt[n]), c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], c["SQ_LDS_IDX_ACTIVE"] / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
print("| kernel | launches | LDS array cycles (all launches) | bank-conflict share | LDS cycles / wave cycles |\n|---|---|---|---|---|")
for a, n, l, s, w in sorted(rows, reverse=True)[:30]:
    print("| \`%s\` | %d | %.3g | %.3f | %.3f |" % (n, l, a, s, w))
PY
cat $R/gpurun_out/lds_conflicts.txt
rm -rf $OUT
