"""Host-side cost of a MultiResUNet batch-1 train step (cProfile) next to its wall time: is the launch path the bottleneck?"""
import cProfile, pstats, importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG="automatic-sem-image-segmentation_amd"
N=importlib.import_module(PKG+".nets"); UN=importlib.import_module(PKG+".UNet_Segmentation"); OPT=importlib.import_module(PKG+".optim")
unet=N.MultiResUNet(16, device="cuda:0", seed=5)
um=UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))
um.sync_metrics=False if hasattr(um,"sync_metrics") else None
x=torch.rand((1,512,512,1)).cuda(); y=(torch.rand((1,512,512,1))>0.9).float().cuda()
for _ in range(3): um.train_step((x,y))
torch.cuda.synchronize()
import time
t=time.perf_counter()
for _ in range(10): um.train_step((x,y))
t1=time.perf_counter()-t
torch.cuda.synchronize()
t2=time.perf_counter()-t
print("host ms/step", t1*100, "wall ms/step", t2*100)
pr=cProfile.Profile(); pr.enable()
for _ in range(5): um.train_step((x,y))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
