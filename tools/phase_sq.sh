#!/bin/bash
# SQ counters of the fused sub-pixel kernel (conv_phase.hip) and of the per-phase gather kernels on the same layers: where do the wave cycles go?
# (PMC passes with --kernel-trace only, as the pool requires.)  Usage on the GPU box: bash tools/phase_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/phasesq
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/phase_once.py <<PY
import importlib, sys, torch
sys.path.insert(0, "$R")
BASE = "automatic-sem-image-segmentation_amd"
E, LY, L = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers", "_lib"))
dev = torch.device("cuda:0")
for (cin, cout, hw) in ((128, 64, 256), (256, 128, 128)):
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, cin, cout, stride=2, padding="same", use_bias=False, transposed=True)
    arena.materialize(); arena["c/kernel"].uniform_(-0.05, 0.05)
    x = E.Act(torch.randn((16, hw, hw, cin), device=dev), requires_grad=False)
    for fused in (0, 1):
        with L.config(phases_fused=fused):
            for _ in range(3):
                conv(E.Tape(enabled=False), x)
            torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/p1 --output-format csv -- python /tmp/phase_once.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/p2 --output-format csv -- python /tmp/phase_once.py > $OUT/p2.log 2>&1
python - <<PY > $R/gpurun_out/phase_sq.txt
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gconv_" in r["Kernel_Name"] and "wprep" not in r["Kernel_Name"]]
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "")[:44] + " grid " + r.get("Grid_Size", "?")
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name].add(r["Dispatch_Id"])
        for name in agg:
            n = len(cnt[name])
            print(d, name, "launches", n, {k: round(v / n) for k, v in sorted(agg[name].items())})
PY
cat $R/gpurun_out/phase_sq.txt
rm -rf $OUT
