"""Prints the headline and the per-class kernel table of a bench.py JSON line.  Usage: python tools/bench_classes.py <file.json> [substring ...]"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("tiles/s", d["value"], "ms/step", d["ms_per_step"], "| dominant", d["roofline"]["kernel"], "frac", d["roofline"]["frac"])
for name, v in d["roofline"]["kernels"].items():
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f"{name:36s} launches {v['launches']:4d}  avg {v['avg_ms'] * 1e3:7.1f} us  per step {v['total_ms_per_step']:7.2f} ms  frac {v['frac']:.3f}")
