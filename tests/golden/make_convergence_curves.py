"""Loss curves of the fp32 ORACLE on the convergence task of tests/test_convergence_gpu.py, five data orders / image-buffer draws
(the seed-to-seed band the HIP curve is held against).  ~7 min of CPU;  python tests/golden/make_convergence_curves.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
spec = importlib.util.spec_from_file_location("conv_test", os.path.join(os.path.dirname(HERE), "test_convergence_gpu.py"))
T = importlib.util.module_from_spec(spec)
spec.loader.exec_module(T)

if __name__ == "__main__":
    torch.set_num_threads(8)
    a, b = T.dataset()
    out = dict(steps=T.STEPS, window=T.WINDOW, seeds=5, data_checksum=float(a.double().sum() + b.double().sum()))
    for seed in range(5):
        c = T.oracle_curves(seed, a, b)
        for k, v in c.items():
            out[f"seed{seed}/{k}"] = v
        print(seed, {k: np.round(v, 3).tolist() for k, v in c.items()}, flush=True)
    np.savez(os.path.join(HERE, "convergence_oracle_curves.npz"), **out)
