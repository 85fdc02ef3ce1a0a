"""Diagnostic: per-tensor count of weights that differ from the reference-generated golden by more than lr / 2 (a flipped Adam sign step)."""
import importlib, os, random, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
B = "automatic-sem-image-segmentation_amd"
L = importlib.import_module(B + "._lib"); CG = importlib.import_module(B + ".CycleGAN"); N = importlib.import_module(B + ".nets"); OPT = importlib.import_module(B + ".optim")
z = np.load(os.path.join(REPO, "tests", "golden", "cyclegan_step_n2_s64_f4.npz"))
n, size, filters, n_steps, seed = (int(v) for v in z["meta"])
for fuse in [int(a) for a in sys.argv[1:]] or [0, 1]:
    with L.config(norm_fuse_fin=fuse):
        nets = dict(gen_a=N.ResnetGenerator(filters=filters, device="cuda:0"), gen_b=N.ResnetGenerator(filters=filters, device="cuda:0"),
                    disc_a=N.PatchDiscriminator(filters=2 * filters, device="cuda:0"), disc_b=N.PatchDiscriminator(filters=2 * filters, device="cuda:0"))
        for nm, net in nets.items():
            net.set_weights([z[f"init/{nm}/{i}"] for i in range(len(net.variable_names))])
        random.seed(seed)
        model = CG.CycleGanModel(nets["gen_a"], nets["gen_b"], nets["disc_a"], nets["disc_b"], image_pool_a=CG.ImagePool(2, 3), image_pool_b=CG.ImagePool(2, 3),
                                 lambda_cycle_a=10, lambda_cycle_b=10, lambda_identity_a=0.5, lambda_identity_b=0.5)
        model.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
        for s in range(n_steps if os.environ.get("STEPS") is None else int(os.environ["STEPS"])):
            model.train_step((z[f"step{s}/real_a"], z[f"step{s}/real_b"]))
    rows = []
    for nm, net in nets.items():
        for i, (name, w) in enumerate(zip(net.variable_names, net.get_weights())):
            d = np.abs(np.asarray(w, np.float64) - z[f"final/{nm}/{i}"])
            k = int((d > 1e-4).sum())
            if k:
                rows.append((k, d.size, nm, name))
    print("fuse", fuse, "total flips", sum(r[0] for r in rows))
    for r in sorted(rows, reverse=True)[:14]:
        print("   %6d of %7d  %s/%s" % r)
