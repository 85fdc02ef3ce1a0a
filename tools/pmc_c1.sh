#!/bin/bash
# SQ counters of the one-channel 7x7 kernels (head 64 -> 1 and stem 1 -> 64 of a generator on (8, 512, 512, .), forward + backward).
# Usage (GPU box): bash tools/pmc_c1.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcc1
mkdir -p $OUT
cat > /tmp/c1_once.py <<PY
import importlib, sys, torch
sys.path.insert(0, "$R")
BASE = "automatic-sem-image-segmentation_amd"
E, LY = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers"))
dev = torch.device("cuda:0")
arena = E.ParamArena(dev)
head = LY.Conv2D(arena, "h", 7, 64, 1, padding=("reflect", 3), use_bias=True, act="tanh")
stem = LY.Conv2D(arena, "s", 7, 1, 64, padding=("reflect", 3), use_bias=False)
arena.materialize(); arena["h/kernel"].normal_(0, 0.02); arena["s/kernel"].normal_(0, 0.02)
x64 = torch.randn((8, 512, 512, 64), device=dev); x1 = torch.randn((8, 512, 512, 1), device=dev)
for _ in range(2):
    for layer, xt in ((head, x64), (stem, x1)):
        tape = E.Tape(); x = E.Act(xt, requires_grad=True); y = layer(tape, x)
        g, _ = y.grad_target(); g.t.normal_(); tape.backward()
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $OUT/p1 --output-format csv -- python /tmp/c1_once.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/p2 --output-format csv -- python /tmp/c1_once.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("_c1_", "conv_out1", "conv_in1")): continue
            k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60] + " grid" + r["Grid_Size"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, v in sorted(agg.items()):
            print(d, k, {a: round(b / n[(k, a)]) for a, b in v.items()})
PY
