"""Diagnostic: every Norm call of the first CycleGAN step of a golden (reference-generated) case, error of its output against a float64
evaluation of the same call's input, with the fused finalize off / on."""
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
for fuse in (0, 1):
    rec = []

    def wrapped(self, tape, x, act=None, act_alpha=0.0, residual=None, out=None, training=True, defer_to=None):
        y = orig(self, tape, x, act=act, act_alpha=act_alpha, residual=residual, out=out, training=training, defer_to=defer_to)
        if not isinstance(y, E.DeferredNorm):
            torch.cuda.synchronize()
            xd = x.dense().double().cpu()
            dims = (1, 2) if self.kind == "instance" else (0, 1, 2)
            m = xd.mean(dims, keepdim=True); v = (xd * xd).mean(dims, keepdim=True) - m * m
            gam = self.arena[f"{self.name}/gamma"].double().cpu() if self.scale else 1.0
            t = (xd - m) / torch.sqrt(v + self.eps) * gam + self.arena[f"{self.name}/beta"].double().cpu()
            if residual is not None:
                t = t + residual.dense().double().cpu()
            t = torch.relu(t) if act == "relu" else (torch.where(t > 0, t, t * act_alpha) if act == "lrelu" else t)
            got = y.dense().double().cpu()
            rec.append((self.name, tuple(xd.shape), act, float((got - t).abs().max() / t.abs().max().clamp_min(1e-30)), float(v.min()), float((m * m).max())))
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
        model.dual_stream = False
        m = model.train_step((z["step0/real_a"], z["step0/real_b"]))
    LY.Norm.__call__ = orig
    print("fuse", fuse, "calls", len(rec))
    for r in sorted(rec, key=lambda r: -r[3])[:6]:
        print("   %-28s %-18s %-6s err %.2e  min var %.2e  max mean^2 %.2e" % r)
    big = sorted(set((r[0], r[1]) for r in rec if r[1][1] * r[1][2] * (r[1][0] if r[0].startswith("zzz") else 1) > 1024))
    print("   groups of more than 1024 pixels:", big)
    print("   errors of those:", [(r[0], r[1], "%.2e" % r[3]) for r in rec if r[1][1] * r[1][2] > 1024])
    print("   metrics", {k: float(v) for k, v in m.items()})
