#!/bin/bash
# PMC passes on the gather weight-gradient kernels (one test case, both kernels).  Usage (GPU box): bash tools/pmc_wgrad.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcw
mkdir -p $OUT
CMD="python -m pytest $R/tests/test_direct_gpu.py -q -m gpu -k weight_gradient_w -p no:cacheprovider"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --kernel-trace -d $OUT/p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wgrad_x6" not in k: continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60] + " grid" + r.get("Grid_Size", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in sorted(agg.items()):
            print(d, k[:90], {a: round(b) for a, b in v.items()})
PY
