"""MultiResUNet train step at a small per-GPU batch: eager issue vs hipGraph replay -- host time to issue a step (no device read) and wall
time per step.  Usage: python tools/unet_graph_probe.py [batch] [size]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
N, UN, OPT = (importlib.import_module(f"{PKG}.{m}") for m in ("nets", "UNet_Segmentation", "optim"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
x = torch.rand((B, S, S, 1), device=dev)
y = (torch.rand((B, S, S, 1), device=dev) > 0.9).float()
for mode in (False, True):
    um = UN.UNetModel(N.MultiResUNet(16, device=dev, seed=5), 9.0, OPT.Adam(1e-3))
    um.graph = mode
    for _ in range(5):
        um.train_step((x, y))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        um.train_step((x, y))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) * 50
    um.sync_metrics = False
    um.train_step((x, y)); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        um.train_step((x, y))
    host = (time.perf_counter() - t) * 50
    torch.cuda.synchronize()
    both = (time.perf_counter() - t) * 50
    captured = isinstance(um._graphs.get((B, S, S)), dict)
    print(f"batch {B} size {S} graph={mode} captured={captured}: wall {wall:.2f} ms/step (metrics read each step); "
          f"host issue {host:.2f} ms/step; back-to-back {both:.2f} ms/step", flush=True)
