#!/bin/bash
# How busy is the GPU in the DEFAULT (multi-stream) step?  rocprofv3 kernel trace of bench.py, then: union of the kernel intervals,
# sum of the kernel durations, average number of kernels in flight.  Usage (GPU box): bash tools/dual_stream_concurrency.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dual
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/t --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.log
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# the timed region: the last 4 steps = the last 4/6 of the dispatches (identical steps)
n = len(ev); ev = ev[n - n * 4 // 6:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
b = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("# default (multi-stream) step, rocprofv3 --kernel-trace of bench.py --steps 4 --warmup 2 (last 4 steps, %d dispatches)" % len(ev))
print("bench line of this run: %.2f tiles/s, %.1f ms per step" % (b["value"], b["ms_per_step"]))
print("wall %.1f ms per step | GPU busy (union of kernel intervals) %.1f ms = %.1f %% | sum of kernel durations %.1f ms | kernels in flight while busy: %.2f on average"
      % ((t1 - t0) / 4e6, busy / 4e6, 100.0 * busy / (t1 - t0), tot / 4e6, tot / busy))
PY
