"""Diagnostic: outputs of every Norm call of the first golden CycleGAN step kept (no synchronisation inside the step), fused finalize off vs on."""
import importlib, os, random, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
B = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(B + ".engine"); LY = importlib.import_module(B + ".layers"); L = importlib.import_module(B + "._lib")
CG = importlib.import_module(B + ".CycleGAN"); N = importlib.import_module(B + ".nets"); OPT = importlib.import_module(B + ".optim")
z = np.load(os.path.join(REPO, "tests", "golden", "cyclegan_step_n2_s64_f4.npz"))
n, size, filters, n_steps, seed = (int(v) for v in z["meta"])
orig = LY.Norm.__call__
orig_bwd = None
R = {}
for fuse in (0, 1):
    rec = []

    def wrapped(self, tape, x, act=None, act_alpha=0.0, residual=None, out=None, training=True, defer_to=None):
        y = orig(self, tape, x, act=act, act_alpha=act_alpha, residual=residual, out=out, training=training, defer_to=defer_to)
        rec.append((self.name, x, y))          # kept alive: read after the step
        return y

    LY.Norm.__call__ = wrapped
    with L.config(norm_fuse_fin=fuse):
        nets = dict(gen_a=N.ResnetGenerator(filters=filters, device="cuda:0"), gen_b=N.ResnetGenerator(filters=filters, device="cuda:0"),
                    disc_a=N.PatchDiscriminator(filters=2 * filters, device="cuda:0"), disc_b=N.PatchDiscriminator(filters=2 * filters, device="cuda:0"))
        for nm, net in nets.items():
            net.set_weights([z[f"init/{nm}/{i}"] for i in range(len(net.variable_names))])
        random.seed(seed)
        model = CG.CycleGanModel(nets["gen_a"], nets["gen_b"], nets["disc_a"], nets["disc_b"], image_pool_a=CG.ImagePool(2, 3), image_pool_b=CG.ImagePool(2, 3),
                                 lambda_cycle_a=10, lambda_cycle_b=10, lambda_identity_a=0.5, lambda_identity_b=0.5)
        model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
        model.train_step((z["step0/real_a"], z["step0/real_b"]))
        torch.cuda.synchronize()
    LY.Norm.__call__ = orig
    R[fuse] = [(nm, x.dense().cpu().double(), y.dense().cpu().double(), (x.get_grad().dense().cpu().double() if x.get_grad() is not None else None),
                (y.get_grad().dense().cpu().double() if (not isinstance(y, E.DeferredNorm) and y.get_grad() is not None) else None)) for nm, x, y in rec]
print("calls", len(R[0]), len(R[1]))
gam = {nm: (nets[g_].arena[f"{nm}/gamma"].double().cpu(), nets[g_].arena[f"{nm}/beta"].double().cpu()) for g_ in ("gen_a", "gen_b") for nm in ("c7_in", "up2")}
for fuse in (0, 1):
    for i, (nm, x, y, dx, dy) in enumerate(R[fuse]):
        if x.shape[1] * x.shape[2] > 1024 and dy is not None and dx is not None:
            xx = x.clone().requires_grad_(True)
            m = xx.mean((1, 2), keepdim=True); v = (xx * xx).mean((1, 2), keepdim=True) - m * m
            # (weights after the step differ from those of the forward pass by one Adam step: gamma / beta enter dx only through the mask and a scale)
            best = None
            for g_ in ("gen_a", "gen_b"):
                ga, be = nets[g_].arena[f"{nm}/gamma"].double().cpu(), nets[g_].arena[f"{nm}/beta"].double().cpu()
                t = torch.relu((xx - m) / torch.sqrt(v + 1e-5) * ga + be)
                gx, = torch.autograd.grad(t, xx, dy, retain_graph=True)
                e = [float((dx[k] - gx[k]).abs().max() / gx[k].abs().max().clamp_min(1e-30)) for k in range(x.shape[0])]
                if best is None or max(e) < max(best):
                    best = e
            print("fuse", fuse, "call", i, nm, tuple(x.shape), "dx error per sample vs float64 (post-step gamma/beta):", ["%.1e" % v_ for v_ in best])
def rd(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
rows = []
for i, (a, b) in enumerate(zip(R[0], R[1])):
    rows.append((i, a[0], tuple(a[1].shape), rd(b[1], a[1]), rd(b[2], a[2]), rd(b[3], a[3]) if a[3] is not None and b[3] is not None else -1,
                 rd(b[4], a[4]) if a[4] is not None and b[4] is not None else -1))
for r in rows:
    if r[2][1] * r[2][2] > 1024 or max(r[3:]) > 1e-4:
        print("call %3d %-10s %-18s  x diff %.1e  y diff %.1e  dx diff %.1e  dy diff %.1e" % r)
