"""Two generator chains (A: G_a then G_b, B: G_b then G_a -- the CycleGAN generator phase without the discriminators) forward and
backward on two HIP streams: iterations 0-1 run serialised (workspaces grow, reference results), the rest concurrently; every
output, input gradient and parameter gradient must stay bit-identical.  Usage: python tools/check_fwd_race.py [S N F]"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "automatic-sem-image-segmentation_amd"
E = importlib.import_module(PKG + ".engine"); NETS = importlib.import_module(PKG + ".nets"); LY = importlib.import_module(PKG + ".layers")
dev = torch.device("cuda:0")
S, N, F = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 4, 32)))
ga = NETS.ResnetGenerator(filters=F, device="cuda:0", seed=1); gb = NETS.ResnetGenerator(filters=F, device="cuda:0", seed=2)
g = torch.Generator().manual_seed(1)
a = (torch.rand((N, S, S, 1), generator=g) * 2 - 1).to(dev); b = ((torch.rand((N, S, S, 1), generator=g) > 0.9).float() * 2 - 1).to(dev)
gy = torch.randn((N, S, S, 1), device=dev)
s1, s2 = E.side_streams(dev)
ref = None
for it in range(7):
    serial = it < 2
    cur = torch.cuda.current_stream()
    for net in (ga, gb):
        net.zero_grad(); net.arena.zero_grad_alt()
    s1.wait_stream(cur); s2.wait_stream(cur)
    ta, tb = E.Tape(), E.Tape()
    with torch.cuda.stream(s1):
        oa = ga(E.Act(torch.cat([a, b], 0), requires_grad=False), True, ta); fa, _ = LY.batch_split(ta, oa, [N, N]); ca = gb(fa, True, ta)
        t, _ = ca.grad_target(); t.t.copy_(gy)
    if serial: torch.cuda.synchronize()
    BMODE = os.environ.get("BMODE", "full")
    with torch.cuda.stream(s2):
        if BMODE != "none":
            ob = gb(E.Act(torch.cat([b, a], 0), requires_grad=False), True, tb); fb, _ = LY.batch_split(tb, ob, [N, N])
            cb = ga(fb, True, tb) if BMODE in ("full", "fwd2") else fb
            t, _ = cb.grad_target(); t.t.copy_(gy)
        else:
            ob = cb = oa
    if serial: torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        ta.backward()
    if serial: torch.cuda.synchronize()
    ga.arena.swap_grads(); gb.arena.swap_grads()
    with torch.cuda.stream(s2):
        if BMODE == "full":
            tb.backward()
    ga.arena.swap_grads(); gb.arena.swap_grads()
    torch.cuda.synchronize()
    extra = [fa.get_grad().dense().clone(), oa.get_grad().dense().clone()]
    if it == 1: ref_extra = extra
    elif it > 1: print("   input grads fa, oa equal:", [bool(torch.equal(x, y)) for x, y in zip(extra, ref_extra)])
    res = [t.dense().clone() for t in (oa, ca, ob, cb)] + [ga.arena.grads.clone(), ga.arena.grads_alt.clone(), gb.arena.grads.clone(), gb.arena.grads_alt.clone()]
    if it == 1: ref = res
    elif it > 1:
        print(it, [bool(torch.equal(x, y)) for x, y in zip(res, ref)])
        for nm, net, idx in (("ga.main", ga, 4), ("ga.alt", ga, 5), ("gb.main", gb, 6), ("gb.alt", gb, 7)):
            if not torch.equal(res[idx], ref[idx]):
                bad = []
                for name, shape, trainable, off in net.arena.specs:
                    if trainable:
                        n = 1
                        for d_ in shape: n *= d_
                        if not torch.equal(res[idx][off:off + n], ref[idx][off:off + n]):
                            bad.append((name, float((res[idx][off:off + n] - ref[idx][off:off + n]).abs().max() / (ref[idx][off:off + n].abs().max() + 1e-30))))
                print("   ", nm, len(bad), bad[:6], "...", bad[-3:])
