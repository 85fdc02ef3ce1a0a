#!/bin/bash
# SQ counters of the MultiResUNet step's kernels (bench.py --only-unet, one step).  Usage (GPU box): bash tools/pmc_unet.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcu
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --only-unet"
SS_DUAL_STREAM=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
SS_DUAL_STREAM=0 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA --kernel-trace -d $OUT/p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("tconv_kernel", "twgrad_kernel", "wgrad_mfma")): continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k] += 1
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0)))[:6]:
            t = v.get("SQ_WAVE_CYCLES", 0)
            if t: print(d, k, "launches", n[k] // 8, {a: round(100 * b / t, 1) for a, b in v.items() if a != "SQ_WAVE_CYCLES"}, "wave Mcycles", round(t / 1e6, 1))
            else: print(d, k, {a: round(b / 1e6, 2) for a, b in v.items()})
PY
