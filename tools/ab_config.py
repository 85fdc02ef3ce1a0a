"""In-process A/B of one ss_config key (or engine / layers module switch) on the train steps: the same models, buffers and box, the
settings alternating in blocks of steps -- separate processes on one box differ by +-2 ms at per-GPU batch 1, more than most switches move.

    python tools/ab_config.py norm_fuse_fin 0 2 --global-batch 1 --rounds 4 --steps 15
    python tools/ab_config.py engine.WPREP_BATCH 0 1 --global-batch 1
"""
import argparse, importlib, os, statistics, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench as BN          # synthetic_tiles
PKG = "automatic-sem-image-segmentation_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("key"); ap.add_argument("values", nargs="+")
    ap.add_argument("--global-batch", type=int, default=8); ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--only", default="both", choices=["both", "cyclegan", "unet"])
    args = ap.parse_args()
    E = importlib.import_module(PKG + ".engine"); L = importlib.import_module(PKG + "._lib"); CG = importlib.import_module(PKG + ".CycleGAN")
    UN = importlib.import_module(PKG + ".UNet_Segmentation"); NETS = importlib.import_module(PKG + ".nets"); OPT = importlib.import_module(PKG + ".optim")
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    S, GB = args.size, args.global_batch
    ga = NETS.ResnetGenerator(filters=64, device=dev, seed=1); gb = NETS.ResnetGenerator(filters=64, device=dev, seed=2)
    da = NETS.PatchDiscriminator(filters=128, device=dev, seed=3); db = NETS.PatchDiscriminator(filters=128, device=dev, seed=4)
    unet = NETS.MultiResUNet(16, device=dev, seed=5)
    model = CG.CycleGanModel(ga, gb, da, db, image_pool_a=CG.ImagePool(2, 50), image_pool_b=CG.ImagePool(2, 50))
    model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
    umodel = UN.UNetModel(unet, 9.0, OPT.Adam(1e-3))
    a_all, b_all = BN.synthetic_tiles(GB, S, 1234)
    a = E.Act(a_all.to(dev).contiguous(), requires_grad=False); b = E.Act(b_all.to(dev).contiguous(), requires_grad=False)
    ux = E.Act(((a.t + 1) / 2).contiguous(), requires_grad=False); uy = E.Act(((b.t + 1) / 2).contiguous(), requires_grad=False)

    def step():
        if args.only != "unet":
            model.train_step((a, b))
        if args.only != "cyclegan":
            umodel.train_step((ux.t, uy.t))

    def apply(v):
        if args.key.startswith(("umodel.", "model.")):          # an attribute of the UNet / CycleGAN train-step object
            obj, attr = (umodel if args.key.startswith("umodel.") else model), args.key.split(".", 1)[1]
            setattr(obj, attr, type(getattr(obj, attr))(int(v)))
        elif "." in args.key:
            mod, attr = args.key.split(".")
            setattr(importlib.import_module(PKG + "." + mod), attr, type(getattr(importlib.import_module(PKG + "." + mod), attr))(int(v)))
        else:
            L.config_set(args.key, int(v))

    for v in args.values:          # warm every setting (caches keyed on the configuration, plans, allocator pools)
        apply(v)
        for _ in range(4):
            step()
    res = {v: [] for v in args.values}
    for r in range(args.rounds):
        for v in (args.values if r % 2 == 0 else args.values[::-1]):
            apply(v)
            step(); step()
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.steps):
                t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
            res[v].append(statistics.median(ts) * 1e3)
    for v in args.values:
        print(f"{args.key}={v}: median ms per step by round {['%.2f' % t for t in res[v]]}  -> {statistics.median(res[v]):.2f}", flush=True)


if __name__ == "__main__":
    main()
