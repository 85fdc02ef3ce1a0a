#!/bin/bash
# SQ counters of the one-phase Winograd GEMM by phase-skipping mode (tile_dbg): where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/x6psq
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/x6p_once.py <<PY
import importlib, sys, torch
sys.path.insert(0, "$R")
BASE = "automatic-sem-image-segmentation_amd"
E, LY, L = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers", "_lib"))
dev = torch.device("cuda:0")
arena = E.ParamArena(dev)
conv = LY.Conv2D(arena, "c", 3, 512, 512, padding=("reflect", 1), use_bias=False)
arena.materialize(); arena["c/kernel"].uniform_(-0.05, 0.05)
x = E.Act(torch.randn((8, 64, 64, 512), device=dev))
for dbg in (0, 32, 192, 224):
    with L.config(tile_dbg=dbg):
        for _ in range(4):
            conv(E.Tape(enabled=False), x)
        torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/p1 --output-format csv -- python /tmp/x6p_once.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/p2 --output-format csv -- python /tmp/x6p_once.py > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_x6p" in r["Kernel_Name"]]
        disp = sorted({int(r["Dispatch_Id"]) for r in rows})
        # 4 launches per mode, in order 0, 32, 192, 224
        for mi, mode in enumerate((0, 32, 192, 224)):
            ids = set(disp[mi * 4 + 1:(mi + 1) * 4])
            agg = collections.defaultdict(float)
            for r in rows:
                if int(r["Dispatch_Id"]) in ids: agg[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids)
            print(d, "dbg", mode, {k: round(v) for k, v in sorted(agg.items())})
PY
