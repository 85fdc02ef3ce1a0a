#!/bin/bash
# SQ counters of the Winograd trunk layer's kernels (forward + backward of one 3x3 256 -> 256 convolution on (8, 128, 128, 256)).
# Usage (GPU box): bash tools/pmc_trunk.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmct
mkdir -p $OUT
cat > /tmp/trunk_once.py <<PY
import importlib, sys, torch
sys.path.insert(0, "$R")
BASE = "automatic-sem-image-segmentation_amd"
E, LY = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers"))
dev = torch.device("cuda:0")
arena = E.ParamArena(dev)
conv = LY.Conv2D(arena, "c", 3, 256, 256, padding=("reflect", 1), use_bias=False)
arena.materialize(); arena["c/kernel"].normal_(0, 0.02)
xt = torch.randn((8, 128, 128, 256), device=dev); gt = torch.randn((8, 128, 128, 256), device=dev)
for _ in range(3):
    tape = E.Tape(); x = E.Act(xt, requires_grad=True); y = conv(tape, x)
    g, _ = y.grad_target(); g.t.copy_(gt); tape.backward()
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p1 --output-format csv -- python /tmp/trunk_once.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/p2 --output-format csv -- python /tmp/trunk_once.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("gemm_", "wino_")): continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60] + " grid" + r.get("Grid_Size", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, v in sorted(agg.items()):
            print(d, k, {a: round(b / n[(k, a)]) for a, b in v.items()})
PY
