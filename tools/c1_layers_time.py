"""Times the one-channel 7x7 layers of a generator (head 64 -> 1, stem 1 -> 64) on (8, 512, 512, .): forward, forward + backward."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BASE = "automatic-sem-image-segmentation_amd"
E, LY, L = (importlib.import_module(f"{BASE}.{m}") for m in ("engine", "layers", "_lib"))
dev = torch.device("cuda:0")
arena = E.ParamArena(dev)
head = LY.Conv2D(arena, "h", 7, 64, 1, padding=("reflect", 3), use_bias=True, act="tanh")
stem = LY.Conv2D(arena, "s", 7, 1, 64, padding=("reflect", 3), use_bias=False)
arena.materialize(); arena["h/kernel"].normal_(0, 0.02); arena["s/kernel"].normal_(0, 0.02)
x64 = torch.randn((8, 512, 512, 64), device=dev); x1 = torch.randn((8, 512, 512, 1), device=dev)
lib = L.load()
for name, layer, xt in (("head", head, x64), ("stem", stem, x1)):
    def run():
        tape = E.Tape(); x = E.Act(xt, requires_grad=True); y = layer(tape, x)
        g, _ = y.grad_target(); g.t.fill_(0.01); tape.backward()
    for _ in range(3): run()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(10): run()
    b.record(); torch.cuda.synchronize(); print(name, "fwd+bwd", round(a.elapsed_time(b) / 10 * 1e3, 1), "us")
