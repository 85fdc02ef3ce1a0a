"""CPU-side checks: the C-ABI library exports every declared symbol, host helpers match the reference-derived
golden vectors, the data-parallel plumbing works under gloo (world_size 2)."""
import ctypes
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

BASE = "automatic-sem-image-segmentation_amd"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = importlib.import_module(BASE + "._lib")
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    header = open(os.path.join(REPO, "include", "semseg_hip.h")).read()
    declared = set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", header))
    declared -= {"ss_status"}
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/semseg_hip.h but not exported"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.ss_version() >= 100


def test_tile_and_stitch_match_reference_vectors(golden_dir):
    HF = importlib.import_module(BASE + ".HelperFunctions")
    z = np.load(os.path.join(golden_dir, "helper_tiling.npz"))
    for name in ("a", "b", "c"):
        h, w, th, tw = (int(v) for v in z[f"tile_{name}/shape"])
        img = z[f"tile_{name}/img"]
        tiles = HF.tile_image(img, tw, th, min_overlap=2)
        np.testing.assert_array_equal(tiles, z[f"tile_{name}/tiles"])
        for mode in (0, 1, 2):
            st = HF.stitch_image(tiles, w, h, min_overlap=2, manage_overlap_mode=mode)
            np.testing.assert_array_equal(st, z[f"tile_{name}/stitched{mode}"])


def test_adam_alpha_and_lr_schedules():
    CG = importlib.import_module(BASE + ".CycleGAN")
    wf = CG.CycleGAN.__new__(CG.CycleGAN)
    wf.learning_rate, wf.epochs, wf.decay_epoch = 2e-4, 50, 37
    assert wf.linear_decay(0) == 2e-4 and wf.linear_decay(36) == 2e-4
    assert abs(wf.linear_decay(37) - 2e-4) < 1e-12 and abs(wf.linear_decay(49) - 2e-4 * (1 - 12 / 13)) < 1e-12
    UN = importlib.import_module(BASE + ".UNet_Segmentation")
    u = UN.UNet.__new__(UN.UNet)
    assert u.step_decay(8, 1e-3) == 1e-3 and u.step_decay(9, 1e-3) == 5e-4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    L = importlib.import_module(BASE + "._lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.SemsegHipError):
        L.load()


_WORKER = r'''
import os, sys, importlib
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
D = importlib.import_module("automatic-sem-image-segmentation_amd.dist")
D.init_from_env("gloo")
r, w = D.rank(), D.world_size()
assert w == 2
class Arena: pass
class Net: pass
net = Net(); net.arena = Arena()
net.arena.params = torch.full((10,), float(r)); net.arena.state = torch.full((4,), float(r))
net.arena.grads = torch.arange(40_000_000 // 1000, dtype=torch.float32) * (r + 1)
D.broadcast_params([net])
assert float(net.arena.params.abs().max()) == 0.0 and float(net.arena.state.abs().max()) == 0.0
D.all_reduce_flat(net.arena.grads, bucket_elems=7000)       # several buckets, ragged tail
assert torch.equal(net.arena.grads, torch.arange(40_000, dtype=torch.float32) * 3)
m = D.mean_scalars(np.array([1.0 + r, 10.0 * r]))
assert np.allclose(m, [1.5, 5.0])
print("RANK_OK", r, flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o


_OVERLAP_WORKER = r"""
import os, sys, importlib
import torch
sys.path.insert(0, sys.argv[1])
B = "automatic-sem-image-segmentation_amd"
D = importlib.import_module(B + ".dist"); E = importlib.import_module(B + ".engine")
D.init_from_env("gloo")
r, w = D.rank(), D.world_size()
E.ParamArena.BUCKET_ELEMS = 64
# dst[bucket] += src[bucket] goes through libsemseg_hip.so on a GPU; here (CPU, gloo) the same contract in torch
E.ParamArena._merge_bucket = lambda self, dst, src, b: dst[b["start"]:b["end"]].add_(src[b["start"]:b["end"]])
class Net: pass
net = Net(); net.arena = arena = E.ParamArena(torch.device("cpu"))
names = [f"layer{i}/kernel" for i in range(12)]           # creation order = forward order; 12 x 32 floats -> 6 buckets of 2 layers
for n in names:
    arena.declare(n, (32,))
arena.materialize()
assert len(arena.buckets) == 6
D.enable_overlap([net])
launched = []
inner = arena.grad_hook
def hook(flat):
    launched.append((phase[0], int(flat.data_ptr() - arena.merge_into.data_ptr()) // 4, flat.numel()))
    return inner(flat)
arena.grad_hook = hook
# two chains (CycleGAN._train_step_dual): both use every variable once; chain A writes `grads`, chain B the alternate buffer
arena._alt()
for n in names: arena.note_use([n])
for n in names: arena.note_use([n])
arena.merge_into, arena.merge_from = arena.grads, arena.grads_alt
arena.begin_backward()
phase = ["chain_a"]
for i in reversed(range(12)):                               # backward of chain A: reverse layer order
    arena.grad(names[i]).fill_((r + 1) * (i + 1)); arena.note_done([names[i]])
assert not launched, "no bucket is final before the second chain has written its share"
arena.swap_grads()
phase = ["chain_b"]
fired_after = {}
for i in reversed(range(12)):
    arena.grad(names[i]).fill_(100.0 * (r + 1)); arena.note_done([names[i]])
    fired_after[i] = len(launched)
arena.swap_grads()
phase = ["after"]
# the LAST layers' buckets were launched while chain B's backward still had the first layers to go
assert fired_after[10] == 1 and fired_after[6] == 3 and fired_after[0] == 6, fired_after
assert [l[0] for l in launched] == ["chain_b"] * 6 and [l[1] for l in launched] == [320, 256, 192, 128, 64, 0], launched
arena.merge_alt_grads()                                     # nothing left to merge: must not add chain B's share twice
arena.merge_into = arena.merge_from = None
D.all_reduce_grads([net])                                   # waits for the launched buckets; none left to send
assert len(launched) == 6
for i in range(12):                                         # sum over ranks of (chain A + chain B)
    want = (1 + 2) * (i + 1) + 100.0 * (1 + 2)
    assert torch.equal(arena.grad(names[i]), torch.full((32,), want)), (i, arena.grad(names[i])[:2], want)
print("RANK_OK", r, flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


def test_generator_buckets_are_exchanged_during_the_second_chains_backward_gloo_world2(tmp_path):
    """VERDICT r4 (missing 3): in the dual-chain CycleGAN step a generator's gradient bucket is final when BOTH chains have run their
    last op on it; the chain that gets there second merges the bucket and launches its all-reduce at once (engine.ParamArena.note_done /
    _fire), i.e. before that chain's backward returns -- here with the real arena / bucket / hook code over gloo, the two chains
    replayed by hand."""
    script = tmp_path / "worker.py"
    script.write_text(_OVERLAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29735", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=180)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o


_SOLO_WORKER = r"""
import os, sys, importlib
import torch
sys.path.insert(0, sys.argv[1])
D = importlib.import_module("automatic-sem-image-segmentation_amd.dist")
SP = importlib.import_module("automatic-sem-image-segmentation_amd.StartProcess")
D.init_from_env("gloo")
r = D.rank()
class Arena: pass
class Net: pass
calls = []
def fake_step(self):
    # what WGAN.simulate_masks -> load_model -> create_model does on the ONE rank that runs a file-producing step: build a model and
    # broadcast its parameters (WassersteinGAN.py create_model); plus the other collectives a model may issue
    assert D.rank() == 0 and D.world_size() == 1
    net = Net(); net.arena = Arena()
    net.arena.params = torch.ones(8); net.arena.state = torch.ones(2); net.arena.grads = torch.ones(8)
    D.broadcast_params([net])
    D.all_reduce_flat(net.arena.grads)
    assert float(D.mean_scalars([3.0])[0]) == 3.0
    calls.append(1)
for key in ("2", "4", "6b"):
    setattr(SP.Workflow, "step_" + key, fake_step)
wf = SP.Workflow(SP.WorkflowOptions(ROOT_DIR=sys.argv[2]))
wf.run(steps=["2", "4", "6b"])
assert D.world_size() == 2 and len(calls) == (3 if r == 0 else 0)
t = torch.tensor([float(r + 1)]); torch.distributed.all_reduce(t); assert float(t) == 3.0      # the group is still in step
print("RANK_OK", r, flush=True)
torch.distributed.destroy_process_group()
"""


def test_rank0_only_workflow_steps_issue_no_collectives_gloo_world2(tmp_path):
    """steps 2 / 4 / 6b run on rank 0 alone while the other ranks wait in the barrier behind the step; a model built
    there must not broadcast (dist.solo).  Without the guard rank 0's broadcast meets rank 1's barrier and the run hangs."""
    script = tmp_path / "worker.py"
    script.write_text(_SOLO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, str(tmp_path)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    try:
        outs = [p.communicate(timeout=180)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o


def test_workflow_options_directories_follow_root_dir():
    SP = importlib.import_module(BASE + ".StartProcess")
    o = SP.WorkflowOptions(ROOT_DIR="/tmp/ss_a")
    o.set("OUTPUT_DIR_UNET", "/x/y")
    o.set("ROOT_DIR", "/tmp/ss_b")          # --set ROOT_DIR=...: the defaulted directories move with it, the named one stays
    assert o.INPUT_DIR_IMAGES == "/tmp/ss_b/Input_Images" and o.OUTPUT_DIR_CYCLEGAN == "/tmp/ss_b/Output_Masks_CycleGAN"
    assert o.OUTPUT_DIR_UNET == "/x/y"
    import dataclasses
    o2 = SP.WorkflowOptions(**dataclasses.asdict(o))          # what --spawn hands to the child process
    assert o2.INPUT_DIR_IMAGES == o.INPUT_DIR_IMAGES and o2.OUTPUT_DIR_UNET == "/x/y"


def test_worker_defaults_follow_the_cgroup_cpu_quota(monkeypatch):
    """A container on a many-core host: os.cpu_count() shows the host, the cgroup quota is what the processes get (16 CPUs of 256 on
    the MI355X boxes).  usable_cores() = min(affinity, quota); the worker pools take one less."""
    import builtins
    import io
    HF = importlib.import_module(BASE + ".HelperFunctions")
    real_open = builtins.open
    state = {"text": "300000 100000\n"}

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            if state["text"] is None:
                raise FileNotFoundError(path)
            return io.StringIO(state["text"])
        if str(path).startswith("/sys/fs/cgroup/cpu/"):
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)

    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert HF.usable_cores() == 3 and HF.default_workers() == 2
    state["text"] = "max 100000\n"
    assert HF.usable_cores() == 64 and HF.default_workers() == 32 and HF.default_workers(16) == 16
    state["text"] = None
    assert HF.usable_cores() == 64
    state["text"] = "50000 100000\n"          # half a CPU: still one process
    assert HF.usable_cores() == 1 and HF.default_workers() == 1


def test_activation_storage_option_reaches_both_trainers(tmp_path):
    """ACTIVATION_STORAGE (not in the reference) is handed to the CycleGAN and MultiResUNet trainers; the default is fp32 storage."""
    SP = importlib.import_module(BASE + ".StartProcess")
    wf = SP.Workflow(SP.WorkflowOptions(ROOT_DIR=str(tmp_path)))
    assert wf._cyclegan().activation_storage == "f32" and wf._unet().activation_storage == "f32"
    wf.o.set("ACTIVATION_STORAGE", "f16")
    assert wf._cyclegan().activation_storage == "f16" and wf._unet().activation_storage == "f16"


def test_connectivity_and_loader_match_reference_vectors(golden_dir, tmp_path):
    from PIL import Image
    HF = importlib.import_module(BASE + ".HelperFunctions")
    z = np.load(os.path.join(golden_dir, "helper_tiling.npz"))
    for i in range(4):
        np.testing.assert_array_equal(HF.eight_to_four_connected(z[f"conn_{i}/in"].copy()), z[f"conn_{i}/out"])
    Image.fromarray(z["load/src"]).save(str(tmp_path / "a.tif"))
    np.testing.assert_array_equal(HF.load_and_preprocess_images(str(tmp_path), normalization_range=(-1, 1)), z["load/m11"])
    np.testing.assert_array_equal(HF.load_and_preprocess_images(str(tmp_path), normalization_range=(0, 1),
                                                                contrast_optimization_range=(0.5, 99.5)), z["load/unet"])
    np.testing.assert_array_equal(HF.load_and_preprocess_images(str(tmp_path), normalization_range=(0, 1), threshold_value=0.5),
                                  z["load/mask"])


def test_otsu_threshold_is_the_variance_maximiser():
    HF = importlib.import_module(BASE + ".HelperFunctions")
    rng = np.random.default_rng(0)
    img = np.concatenate([rng.normal(60, 10, 4000), rng.normal(180, 15, 2000)]).clip(0, 255).astype(np.uint8).reshape(60, 100)
    t = HF.threshold_otsu(img)
    best, best_v = None, -1.0
    for c in range(int(img.min()), int(img.max())):       # brute force over every split value
        lo, hi = img[img <= c], img[img > c]
        v = lo.size * hi.size * (lo.mean() - hi.mean()) ** 2
        if v > best_v:
            best, best_v = c, v
    assert t == best and 60 < t < 180
    lab = HF.segment(img, -1, watershed_lines=False)
    assert set(np.unique(lab)) <= {0, 255}
    lab_ws = HF.segment(img, -1, watershed_lines=True)
    assert set(np.unique(lab_ws)) <= {0, 255} and np.all(lab_ws <= lab)      # splitting only removes pixels


def test_three_way_bf16_split_is_exact():
    """The arithmetic claim behind the x6 contraction (csrc/conv_mfma_x6.hip, common.h ss_split3x2), restated in numpy:
    h = bf16(v), m = bf16(v - h), l = bf16(v - h - m) with round-to-nearest-even reproduces every finite fp32 v EXACTLY as
    h + m + l, and the three cross terms the kernel drops (m*l, l*m, l*l) are below 2^-24 of |a*b|."""
    rng = np.random.default_rng(0)

    def bf16(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
        return (u.astype(np.uint32) << 16).view(np.float32)

    def split(v):
        h = bf16(v)
        r1 = (v - h).astype(np.float32)
        m = bf16(r1)
        r2 = (r1 - m).astype(np.float32)
        return h, m, bf16(r2)

    v = np.concatenate([rng.standard_normal(400000), rng.standard_normal(200000) * 1e-20, rng.random(200000) * 1e20,
                        np.array([1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 65504.0, 2.0 ** -100])]).astype(np.float32)
    h, m, l = split(v)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), v.astype(np.float64))
    a, b = v[:300000], v[300000:600000]
    (ah, am, al), (bh, bm, bl) = split(a), split(b)
    f = lambda x: x.astype(np.float64)
    six = f(ah) * f(bh) + f(ah) * f(bm) + f(am) * f(bh) + f(ah) * f(bl) + f(al) * f(bh) + f(am) * f(bm)
    exact = f(a) * f(b)
    assert float(np.max(np.abs(six - exact) / np.abs(exact))) <= 2.0 ** -24


def test_keras_archive_round_trip(tmp_path):
    """model.save('model.keras') / load: a Keras-3 style zip (config.json, metadata.json, model.weights.h5 with
    <model>/layers/<keras layer name>/vars/<i> groups and the Adam states) round-trips every variable and the optimizer state
    bit for bit (CycleGAN.py:221,228; UNet_Segmentation.py:287,303).  HDF5 through h5py in-process or the stand-alone converter."""
    import importlib
    import zipfile

    import numpy as np
    import torch

    base = "automatic-sem-image-segmentation_amd"
    K = importlib.import_module(base + ".keras_io")
    if not (importlib.util.find_spec("h5py") or os.path.exists(K._h5py_python())):
        pytest.skip("no h5py anywhere")
    N, UN, CG, OPT = (importlib.import_module(f"{base}.{m}") for m in ("nets", "UNet_Segmentation", "CycleGAN", "optim"))
    # UNet
    net = N.MultiResUNet(16, device="cpu", seed=3)
    for name in net.variable_names:
        if name.endswith(("moving_mean", "beta")):
            net.arena[name].uniform_(-1, 1)
    model = UN.UNetModel(net, 9.0, OPT.Adam(1e-3))
    model.optimizer.iterations = 7
    net.arena.m.uniform_(-1, 1); net.arena.v.uniform_(0, 1)
    path = str(tmp_path / "model.keras")
    model.save(path)
    with zipfile.ZipFile(path) as z:
        assert sorted(z.namelist()) == ["config.json", "metadata.json", "model.weights.h5"]
    _, _, arrays = K.read_archive(path)
    assert "layers/conv2d/vars/0" in arrays and "layers/batch_normalization/vars/2" in arrays and "layers/conv2d_transpose/vars/1" in arrays
    assert arrays["layers/conv2d_transpose/vars/0"].shape[:2] == (2, 2) and "optimizer/vars/0" in arrays
    back = UN.UNetModel.load(path, device="cpu")
    for a, b in zip(net.get_weights(), back.net.get_weights()):
        assert np.array_equal(a, b)
    assert back.optimizer.iterations == 7
    for name, shape, trainable, off in net.arena.specs:          # (the 16-byte alignment gaps between variables are not state)
        if trainable:
            n_ = int(np.prod(shape))
            assert torch.equal(back.net.arena.m[off:off + n_], net.arena.m[off:off + n_]), name
            assert torch.equal(back.net.arena.v[off:off + n_], net.arena.v[off:off + n_]), name
    # CycleGAN: four networks under their attribute names, layer-name counters continue across them
    kw = dict(filters=4, device="cpu")
    cg = CG.CycleGanModel(N.ResnetGenerator(seed=1, **kw), N.ResnetGenerator(seed=2, **kw), N.PatchDiscriminator(filters=8, device="cpu", seed=3),
                          N.PatchDiscriminator(filters=8, device="cpu", seed=4))
    cg.compile(OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5), OPT.Adam(2e-4, beta_1=0.5))
    cg.gen_b_optimizer.iterations = 3
    cg.gen_b_optimizer.learning_rate = 1.25e-4          # e.g. saved part-way through the linear decay (CycleGAN.py:310-317)
    path = str(tmp_path / "cg.keras")
    cg.save(path)
    _, cfg, arrays = K.read_archive(path)
    # every network is its own Keras container: the per-class layer-name counters restart -- gen_b starts at 'conv2d' again
    assert "gen_a/layers/conv2d/vars/0" in arrays and "gen_b/layers/conv2d/vars/0" in arrays and cfg["filters"] == 4
    assert not any(k.startswith("gen_b/layers/conv2d_25/") for k in arrays)
    # optimizer variables in Keras' tracking order: iterations, learning_rate, momentums..., velocities...
    n_train = sum(1 for s_ in cg.gen_a.arena.specs if s_[2])
    assert sum(1 for k in arrays if k.startswith("gen_a_optimizer/vars/")) == 2 + 2 * n_train
    assert float(arrays["gen_b_optimizer/vars/1"]) == pytest.approx(1.25e-4) and arrays["gen_a_optimizer/vars/2"].shape == cg.gen_a.arena.specs[0][1]
    back = CG.CycleGanModel.load(path, device="cpu")
    for nm in ("gen_a", "gen_b", "disc_a", "disc_b"):
        for a, b in zip(getattr(cg, nm).get_weights(), getattr(back, nm).get_weights()):
            assert np.array_equal(a, b)
    assert back.gen_b_optimizer.iterations == 3
    # a model saved part-way through the linear decay resumes with ITS learning rate and betas, not the workflow defaults
    assert back.gen_b_optimizer.learning_rate == pytest.approx(1.25e-4) and back.gen_a_optimizer.learning_rate == pytest.approx(2e-4)
    assert back.disc_a_optimizer.beta_1 == 0.5
    # archives of this file's first version (one counter across the networks, interleaved Adam slots, no learning_rate) still load
    legacy = {}
    counters = K.NameCounters()
    for nm in ("gen_a", "gen_b", "disc_a", "disc_b"):
        legacy.update(K.net_arrays(getattr(cg, nm), nm + "/", counters))
        net = getattr(cg, nm)
        legacy[f"{nm}_optimizer/vars/0"] = np.asarray(5, dtype=np.int64)
        m_, v_ = net.arena.m.numpy(), net.arena.v.numpy()
        i = 1
        for _, shape, trainable, off in net.arena.specs:
            if trainable:
                n_ = int(np.prod(shape))
                legacy[f"{nm}_optimizer/vars/{i}"], legacy[f"{nm}_optimizer/vars/{i + 1}"] = m_[off:off + n_].reshape(shape), v_[off:off + n_].reshape(shape)
                i += 2
    assert "gen_b/layers/conv2d_25/vars/0" in legacy
    old_path = str(tmp_path / "old.keras")
    K.write_archive(old_path, legacy, "CycleGanModel", {k: v for k, v in cg._config().items() if k != "optimizers"})
    old = CG.CycleGanModel.load(old_path, device="cpu")
    for nm in ("gen_a", "gen_b", "disc_a", "disc_b"):
        for a, b in zip(getattr(cg, nm).get_weights(), getattr(old, nm).get_weights()):
            assert np.array_equal(a, b)
    assert old.gen_a_optimizer.iterations == 5 and old.gen_a_optimizer.learning_rate == pytest.approx(2e-4)


def test_keras_archive_falls_back_to_npz_without_hdf5(tmp_path, monkeypatch):
    """a box without h5py (and without the helper interpreter) must not lose a finished training at the final save():
    the archive's arrays + config go to '<path>.npz' with a warning, and load() of the '.keras' path finds them."""
    import importlib

    import numpy as np

    base = "automatic-sem-image-segmentation_amd"
    K = importlib.import_module(base + ".keras_io")
    N, UN, OPT = (importlib.import_module(f"{base}.{m}") for m in ("nets", "UNet_Segmentation", "optim"))

    def no_h5(arrays, h5_path):
        raise K.KerasIOError("writing model.weights.h5 needs h5py (test: pretend there is none)")
    monkeypatch.setattr(K, "_write_h5", no_h5)
    monkeypatch.setenv("SS_H5PY_PYTHON", str(tmp_path / "no-such-python"))
    net = N.MultiResUNet(16, device="cpu", seed=3)
    model = UN.UNetModel(net, 9.0, OPT.Adam(5e-4))
    model.optimizer.iterations = 4
    path = str(tmp_path / "model.keras")
    with pytest.warns(UserWarning, match="model.keras.npz"):
        model.save(path)
    assert not os.path.exists(path) and os.path.exists(path + ".npz")
    back = UN.UNetModel.load(path, device="cpu")
    for a, b in zip(net.get_weights(), back.net.get_weights()):
        assert np.array_equal(a, b)
    assert back.optimizer.iterations == 4 and back.optimizer.learning_rate == pytest.approx(5e-4)
    if not importlib.util.find_spec("h5py"):
        with pytest.warns(UserWarning, match="written as"):
            K.warn_if_no_hdf5("test")


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: no module of the package, bench.py outside its `cpu_baseline` leg, or __graft_entry__ outside
    build() / smoke() may import it (a product path that routes through the CPU restatement would void every parity claim)."""
    import ast
    pkg = os.path.join(REPO, BASE)
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{fn} imports the oracle"
            if isinstance(node, ast.Call) and getattr(node.func, "attr", getattr(node.func, "id", "")) in ("import_module", "__import__"):
                for a in node.args:
                    assert not (isinstance(a, ast.Constant) and isinstance(a.value, str) and a.value.split(".")[0] == "oracle"), fn
    # bench.py: only inside cpu_baseline()
    src = open(os.path.join(REPO, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom))
                    and any((getattr(a, "name", "") or "").startswith("oracle") for a in getattr(n, "names", [])) or
                    (isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle"))]
            if uses:
                assert node.name == "cpu_baseline", f"bench.py::{node.name} imports the oracle"
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and ((getattr(n, "module", "") or "").startswith("oracle")
                                                                                 or any(a.name.startswith("oracle") for a in n.names))]
    assert not top, "bench.py imports the oracle at module level"


def test_prefetch_keeps_order_contents_and_errors():
    """HelperFunctions.prefetch (the on-demand loaders of both trainers): results in key order whatever finishes first, every key once,
    a worker's exception surfaces at its position, depth 0 = the plain loop."""
    import threading
    import time
    HF = importlib.import_module(BASE + ".HelperFunctions")
    seen = []
    lock = threading.Lock()

    def fetch(k):
        time.sleep(0.002 * ((7 * k) % 5))          # out-of-order completion
        with lock:
            seen.append(k)
        if k == 13:
            raise ValueError("boom")
        return k * k

    out = []
    with pytest.raises(ValueError):
        for v in HF.prefetch(fetch, range(20), depth=4, workers=3):
            out.append(v)
    assert out == [k * k for k in range(13)]
    assert sorted(set(seen)) == sorted(seen) and max(seen) <= 13 + 4          # never more than `depth` ahead
    assert list(HF.prefetch(lambda k: -k, range(5), depth=0)) == [0, -1, -2, -3, -4]
    assert list(HF.prefetch(lambda k: k, [], depth=4)) == []


def test_job_pool_survives_a_killed_worker_process():
    """multiprocessing.Pool loses a task whose worker dies and blocks forever; HelperFunctions.JobPool (concurrent.futures,
    spawn context) notices the broken pool and runs what is unfinished in the calling process."""
    import warnings
    HF = importlib.import_module(BASE + ".HelperFunctions")
    pool = HF.JobPool(HF._jobpool_selftest_job, 2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for job in (1, 2, "die", 3, 4, 5):
            pool.submit(job, in_flight=4)
        res = pool.close()
    assert sorted(r for r in res if r != "survived") == [2, 4, 6, 8, 10] and res.count("survived") == 1, res
    assert any("worker processes lost" in str(x.message) for x in w)
    inline = HF.JobPool(HF._jobpool_selftest_job, 1)          # workers <= 1: no processes at all
    inline.submit("die")
    assert inline.close() == ["survived"]
