"""GPU parity of every C-ABI op (through the layer/tape engine) against the oracle ops on the same seeded
inputs.  fp32 tolerances: forward |d| <= 1e-4*max|ref| (+1e-5 abs), gradients rel-L2 <= 1e-4."""
import os
import zlib
import importlib

import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _mods():
    base = "automatic-sem-image-segmentation_amd"
    return (importlib.import_module(base + ".engine"), importlib.import_module(base + ".layers"),
            importlib.import_module(base + "._lib"))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_close(got, ref, what, rtol=1e-4, atol=0.0):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    tol = rtol * max(float(np.abs(ref).max()), 1e-6) + 1e-6 + atol
    err = float(np.abs(got - ref).max())
    assert err <= tol, f"{what}: max|d|={err:.3e} tol={tol:.3e} relL2={rel_l2(got, ref):.3e}"


CONV_CASES = [
    # name, k, cin, cout, stride, padding, bias, act, transposed, n, h, w
    ("res3x3_reflect", 3, 32, 32, 1, ("reflect", 1), False, None, False, 2, 16, 16),
    ("res3x3_reflect_big", 3, 64, 128, 1, ("reflect", 1), False, None, False, 1, 20, 12),
    ("c7_in", 7, 1, 8, 1, ("reflect", 3), False, None, False, 2, 16, 16),
    ("c7_out_tanh", 7, 8, 1, 1, ("reflect", 3), True, "tanh", False, 2, 16, 16),
    ("down_s2_same", 3, 8, 16, 2, "same", False, None, False, 2, 16, 16),
    ("down_s2_same_odd", 3, 8, 16, 2, "same", False, None, False, 1, 15, 17),
    ("disc_in", 4, 1, 8, 2, "valid", True, "lrelu", False, 2, 32, 32),
    ("disc_down", 4, 8, 16, 2, "valid", False, None, False, 2, 15, 15),
    ("disc_out", 4, 16, 1, 1, "valid", True, None, False, 2, 6, 6),
    ("up_T3", 3, 16, 8, 2, "same", False, None, True, 2, 8, 8),
    ("up_T2_bias", 2, 13, 8, 2, "same", True, None, True, 2, 8, 8),
    ("unet_3x3_odd", 3, 13, 17, 1, "same", False, None, False, 2, 16, 16),
    ("unet_1x1", 1, 25, 16, 1, "same", False, None, False, 2, 16, 16),
    ("unet_first", 3, 1, 4, 1, "same", False, None, False, 2, 16, 16),
    ("wide", 3, 40, 200, 1, "same", False, None, False, 1, 12, 12),
    ("fast_paths_w40", 3, 32, 64, 1, ("reflect", 1), False, None, False, 2, 12, 40),
    ("fast_paths_s2", 3, 32, 48, 2, "same", False, None, False, 2, 64, 72),
    ("fast_paths_T", 3, 64, 32, 2, "same", False, None, True, 1, 32, 36),
    # Winograd F(2x2,3x3) path: 3x3 stride 1, >= 64 channels, >= 1024 output tiles
    ("wino_reflect", 3, 64, 64, 1, ("reflect", 1), False, None, False, 2, 48, 48),
    ("wino_reflect_unaligned", 3, 64, 64, 1, ("reflect", 1), False, None, False, 2, 46, 50),   # not a multiple of 4: padded gradient + fold pass
    ("wino_same_bias_tanh", 3, 64, 96, 1, "same", True, "tanh", False, 2, 48, 50),   # smooth act: a kinked one flips masks at |y|~1e-6
    ("wino_same_odd", 3, 96, 64, 1, "same", False, None, False, 3, 47, 45),
    ("wino_valid", 3, 64, 64, 1, "valid", False, None, False, 2, 50, 50),
    # wide trunk shapes: pre-split-plane GEMMs (gemm_x6p persistent workgroups: tile counts that do not divide by 8 XCDs / 256 CUs),
    # weight gradient on K-major planes (gemm_tn_x3h: 256 | Cin, 128 | Cout, 32 | tiles) and its fallbacks
    ("wino_tn_256_384", 3, 256, 384, 1, ("reflect", 1), False, None, False, 2, 32, 32),          # 128 tiles, N tiles = 3
    ("wino_tn_512_128_same", 3, 512, 128, 1, "same", False, None, False, 3, 32, 32),             # 192 tiles
    ("wino_256_tiles_not_32", 3, 256, 256, 1, ("reflect", 1), False, None, False, 3, 24, 40),    # 180 tiles: in-kernel-split weight gradient
    ("wino_384_cin_not_256", 3, 384, 256, 1, "same", False, None, False, 1, 64, 64),
    # single-output-channel convs take the two-stage (1x1 MFMA GEMM + tap sum / tap scatter) path from 16 channels up
    ("c7_out_16_two_stage", 7, 16, 1, 1, ("reflect", 3), True, "tanh", False, 2, 16, 16),
    ("c7_in_16_two_stage_dgrad", 7, 1, 16, 1, ("reflect", 3), False, None, False, 2, 16, 16),
    ("disc_in_16_two_stage_dgrad", 4, 1, 16, 2, "valid", True, "lrelu", False, 2, 32, 32),
    ("disc_out_32_two_stage", 4, 32, 1, 1, "valid", True, None, False, 2, 7, 9),
    ("head_same_two_stage", 3, 24, 1, 1, "same", False, None, False, 2, 10, 10),
    # >= 16384 output pixels, stride 1, one channel on one side: LDS-tiled VALU kernels (conv_c1.hip), forward and data gradient
    ("c7_out_tiled", 7, 16, 1, 1, ("reflect", 3), True, "tanh", False, 1, 128, 130),
    ("c7_in_tiled", 7, 1, 32, 1, ("reflect", 3), False, None, False, 1, 130, 128),
    ("c4_out_tiled_valid", 4, 12, 1, 1, "valid", True, None, False, 2, 100, 96),
    ("c3_in_tiled_same_bias", 3, 1, 16, 1, "same", True, "lrelu", False, 2, 96, 100),
    # ... with >= 32 channels on the wide side: fp16 matrix-core kernels (x3h arithmetic), the data gradient of a reflection-padded
    # many -> 1 layer with the reflection's transpose folded into the one-channel im2col (no padded gradient, no fold pass)
    ("c7_in_mfma_64", 7, 1, 64, 1, ("reflect", 3), False, None, False, 1, 130, 128),
    ("c7_out_mfma_64_fold", 7, 64, 1, 1, ("reflect", 3), True, "tanh", False, 1, 128, 130),
    ("c7_out_mfma_32_fold_ragged", 7, 32, 1, 1, ("reflect", 3), False, None, False, 2, 100, 90),
    ("c4_out_mfma_valid_32", 4, 32, 1, 1, "valid", True, None, False, 2, 100, 96),
    ("c3_in_mfma_same_bias_96", 3, 1, 96, 1, "same", True, "lrelu", False, 2, 96, 100),
    # ... and with stride 2 (the discriminators' 4x4 stem, CycleGAN.py:388-396): the same kernel on a 2x larger halo tile
    ("disc_in_mfma_s2_64_valid", 4, 1, 64, 2, "valid", True, "lrelu", False, 2, 200, 190),
    ("c3_in_mfma_s2_32_same_odd", 3, 1, 32, 2, "same", False, None, False, 3, 131, 150),
    # >= 65536 pixels: their weight gradient on the fp16 matrix cores as well (wgrad_c1_x3h_kernel: K-major planes, transposing LDS reads)
    ("c7_out_wgrad_64", 7, 64, 1, 1, ("reflect", 3), True, "tanh", False, 1, 256, 260),
    ("c7_in_wgrad_64_ragged", 7, 1, 64, 1, ("reflect", 3), False, None, False, 1, 259, 256),
    ("c3_in_wgrad_32_same", 3, 1, 32, 1, "same", False, None, False, 2, 200, 180),
    ("c4_out_wgrad_valid_32", 4, 32, 1, 1, "valid", True, None, False, 1, 260, 256),
    # >= 65536 output pixels, stride 1, <= 64 channels: LDS-staged tile kernels (conv_tile.hip): forward / data gradient on the fp16
    # matrix cores with per-tile scales, weight gradient with fp32 MFMA (the MultiResUNet's 512x512 / 256x256 layers)
    ("tile_3x3_16_16", 3, 16, 16, 1, "same", False, None, False, 1, 256, 256),
    ("tile_3x3_odd_25_13_ragged", 3, 25, 13, 1, "same", False, None, False, 2, 200, 180),
    ("tile_1x1_25_51_bias_tanh", 1, 25, 51, 1, "same", True, "tanh", False, 1, 256, 272),
    ("tile_3x3_1_4", 3, 1, 4, 1, "same", False, None, False, 1, 256, 256),
    ("tile_3x3_64_17", 3, 64, 17, 1, "same", False, None, False, 1, 264, 256),
    ("tile_3x3_35_53", 3, 35, 53, 1, "same", False, None, False, 1, 256, 256),
    ("tile_3x3_reflect_8_8", 3, 8, 8, 1, ("reflect", 1), False, None, False, 1, 258, 254),
    ("tile_1x1_32_1_head", 1, 32, 1, 1, "same", False, None, False, 1, 256, 256),
    ("tile_1x1_64_105", 1, 64, 105, 1, "same", False, None, False, 1, 256, 256),
    ("tile_3x3_valid_8_32", 3, 8, 32, 1, "valid", False, None, False, 1, 300, 260),
]


def oracle_conv(x, w, b, k, stride, padding, act, transposed):
    if transposed:
        y = O.conv2d_transpose(x, w, b, stride)
    elif isinstance(padding, tuple):
        y = O.conv2d(O.reflection_pad(x, (2 * padding[1], 2 * padding[1])), w, b, stride, "valid")
    else:
        y = O.conv2d(x, w, b, stride, padding)
    if act == "tanh":
        y = torch.tanh(y)
    elif act == "lrelu":
        y = O.leaky_relu(y, 0.2)
    return y


@pytest.mark.parametrize("algo", ["auto", "direct", "mfma"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(case, algo):
    E, LY, L = _mods()
    name, k, cin, cout, stride, padding, bias, act, transposed, n, h, w = case
    if algo == "mfma" and (cout < 2):
        pytest.skip("Cout==1 heads are direct-kernel shapes")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)          # stable across processes (hash() is salted per process)
    arena = E.ParamArena(dev)
    layer = LY.Conv2D(arena, "c", k, cin, cout, stride=stride, padding=padding, use_bias=bias, act=act,
                      act_alpha=0.2, transposed=transposed,
                      algo={"auto": L.ALGO_AUTO, "direct": L.ALGO_DIRECT, "mfma": L.ALGO_MFMA}[algo])
    arena.materialize()
    wshape = (k, k, cout, cin) if transposed else (k, k, cin, cout)
    w_cpu = (torch.rand(wshape, generator=g) - 0.5) * 0.5
    b_cpu = torch.rand(cout, generator=g) - 0.5 if bias else None
    x_cpu = torch.rand((n, h, w, cin), generator=g) * 2 - 1
    arena["c/kernel"].copy_(w_cpu)
    if bias:
        arena["c/bias"].copy_(b_cpu)
    # oracle
    xr = x_cpu.clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    br = b_cpu.clone().requires_grad_(True) if bias else None
    yr = oracle_conv(xr, wr, br, k, stride, padding, act, transposed)
    gy = torch.rand(yr.shape, generator=g) - 0.5
    yr.backward(gy)
    # HIP
    tape = E.Tape()
    x = E.Act(x_cpu.to(dev), requires_grad=True)
    y = layer(tape, x)
    torch.cuda.synchronize()
    assert_close(y.dense().cpu(), yr.detach(), f"{name}/{algo} fwd")
    gt, acc = y.grad_target()
    gt.t.copy_(gy.to(dev))
    arena.zero_grad()
    tape.backward()
    torch.cuda.synchronize()
    assert_close(x.get_grad().dense().cpu(), xr.grad, f"{name}/{algo} dx", rtol=2e-4)
    assert_close(arena.grad("c/kernel").cpu(), wr.grad, f"{name}/{algo} dw", rtol=2e-4)
    if bias:
        # a bias gradient is a sum over every output pixel: for a one-channel layer it can land near zero, where a tolerance relative
        # to the value itself is smaller than fp32 summation noise -- floor it at 1e-7 of the sum of magnitudes
        assert_close(arena.grad("c/bias").cpu(), br.grad, f"{name}/{algo} db", rtol=2e-4, atol=1e-7 * float(gy.abs().sum()) / max(cout, 1))


@pytest.mark.parametrize("hw", [(12, 12), (192, 200)], ids=["small", "tile_kernels"])
def test_conv_into_and_from_channel_slices(hw):
    """Keras concatenate without copies: conv reads a slice and writes a slice of wider buffers (small maps: implicit-GEMM kernels;
    >= 65536 pixels: the LDS-staged tile kernels with strided, 16-byte-unaligned channel views)."""
    E, LY, L = _mods()
    H, W = hw
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    arena = E.ParamArena(dev)
    layer = LY.Conv2D(arena, "c", 3, 8, 13, padding="same")
    arena.materialize()
    w_cpu = torch.rand((3, 3, 8, 13), generator=g) - 0.5
    arena["c/kernel"].copy_(w_cpu)
    xin = torch.rand((2, H, W, 20), generator=g)
    xbuf = E.Act(xin.to(dev))
    ybuf = E.Act(torch.zeros((2, H, W, 30), device=dev))
    tape = E.Tape()
    y = layer(tape, xbuf.slice(4, 8), out=ybuf.slice(10, 13))
    xr = xin[..., 4:12].clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    yr = O.conv2d(xr, wr, None, 1, "same")
    assert_close(y.dense().cpu(), yr.detach(), "slice fwd")
    assert float(ybuf.t[..., :10].abs().max()) == 0.0 and float(ybuf.t[..., 23:].abs().max()) == 0.0
    gy = torch.rand(yr.shape, generator=g) - 0.5
    yr.backward(gy)
    gt, acc = y.grad_target()
    assert acc == 1
    gt.t[..., 10:23] = gy.to(dev)
    arena.zero_grad()
    tape.backward()
    assert_close(xbuf.slice(4, 8).get_grad().dense().cpu(), xr.grad, "slice dx", rtol=2e-4)
    assert_close(arena.grad("c/kernel").cpu(), wr.grad, "slice dw", rtol=2e-4)


NORM_CASES = [
    ("in_relu", "instance", 32, "relu", False, True, (2, 16, 16)),
    ("in_none_res", "instance", 64, None, True, True, (2, 8, 8)),
    ("in_lrelu", "instance", 16, "lrelu", False, True, (3, 15, 15)),
    ("bn_relu_noscale", "batch", 13, "relu", False, False, (2, 16, 16)),
    ("bn_relu_res", "batch", 25, "relu", True, True, (2, 16, 16)),
    ("bn_plain", "batch", 4, None, False, True, (4, 32, 32)),
    ("bn_sigmoid", "batch", 1, "sigmoid", False, False, (2, 16, 16)),
    ("in_big", "instance", 512, "relu", False, True, (1, 64, 64)),
    # mid-size groups (> 1024 pixels, <= 8 M elements): the apply kernels finalize the statistics of their own channel block
    # (norm_fuse_fin) -- several groups (parameter gradients summed over them), odd widths (serial lane sums), residuals
    ("in_mid_g3", "instance", 256, "relu", False, True, (3, 40, 40)),
    ("in_mid_res", "instance", 128, None, True, True, (2, 48, 48)),
    ("in_mid_lrelu_noscale", "instance", 64, "lrelu", False, False, (4, 36, 36)),
    ("bn_mid_odd51", "batch", 51, "relu", False, True, (2, 64, 64)),
    ("bn_mid_17_res", "batch", 17, "relu", True, True, (1, 96, 96)),
    ("bn_mid_212", "batch", 212, "relu", False, True, (2, 40, 40)),
]


@pytest.mark.parametrize("case", NORM_CASES, ids=[c[0] for c in NORM_CASES])
def test_norm_fwd_bwd(case):
    E, LY, L = _mods()
    name, kind, c, act, use_res, scale, (n, h, w) = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(len(name))
    arena = E.ParamArena(dev)
    layer = LY.Norm(arena, "n", c, kind, scale=scale)
    arena.materialize()
    x_cpu = torch.rand((n, h, w, c), generator=g) * 3 - 1
    r_cpu = torch.rand((n, h, w, c), generator=g) - 0.5 if use_res else None
    gam = torch.rand(c, generator=g) + 0.5
    bet = torch.rand(c, generator=g) - 0.5
    if scale:
        arena["n/gamma"].copy_(gam)
    arena["n/beta"].copy_(bet)
    if kind == "batch":
        arena["n/moving_mean"].fill_(0.25)
        arena["n/moving_variance"].fill_(2.0)
    xr = x_cpu.clone().requires_grad_(True)
    rr = r_cpu.clone().requires_grad_(True) if use_res else None
    gr = gam.clone().requires_grad_(True)
    br = bet.clone().requires_grad_(True)
    if kind == "instance":
        z = O.instance_norm(xr, gr if scale else torch.ones(c), br)
        nmm = nmv = None
    else:
        z, nmm, nmv = O.batch_norm(xr, gr if scale else None, br, torch.full((c,), 0.25), torch.full((c,), 2.0), True)
    if use_res:
        z = z + rr
    yr = {"relu": torch.relu, "lrelu": lambda t: O.leaky_relu(t, 0.2), "sigmoid": torch.sigmoid, None: lambda t: t}[act](z)
    gy = torch.rand(yr.shape, generator=g) - 0.5
    yr.backward(gy)

    tape = E.Tape()
    x = E.Act(x_cpu.to(dev))
    res = E.Act(r_cpu.to(dev)) if use_res else None
    y = layer(tape, x, act=act, act_alpha=0.2, residual=res)
    assert_close(y.dense().cpu(), yr.detach(), f"{name} fwd", rtol=2e-4)
    if kind == "batch":
        assert_close(arena["n/moving_mean"].cpu(), nmm, f"{name} moving_mean")
        assert_close(arena["n/moving_variance"].cpu(), nmv, f"{name} moving_var")
    gt, _ = y.grad_target()
    gt.t.copy_(gy.to(dev))
    arena.zero_grad()
    tape.backward()
    assert_close(x.get_grad().dense().cpu(), xr.grad, f"{name} dx", rtol=5e-4)
    if use_res:
        assert_close(res.get_grad().dense().cpu(), rr.grad, f"{name} dres", rtol=2e-4)
    if scale:
        assert_close(arena.grad("n/gamma").cpu(), gr.grad, f"{name} dgamma", rtol=5e-4)
    assert_close(arena.grad("n/beta").cpu(), br.grad, f"{name} dbeta", rtol=5e-4)
    # inference-mode batch norm
    if kind == "batch":
        y2 = layer(E.Tape(enabled=False), x, act=act, act_alpha=0.2, residual=res, training=False)
        zi, _, _ = O.batch_norm(x_cpu, gam if scale else None, bet, nmm, nmv, False)
        if use_res:
            zi = zi + r_cpu
        yi = {"relu": torch.relu, "lrelu": lambda t: O.leaky_relu(t, 0.2), "sigmoid": torch.sigmoid, None: lambda t: t}[act](zi)
        assert_close(y2.dense().cpu(), yi, f"{name} infer", rtol=2e-4)


def test_norm_fused_finalize_agrees_with_the_finalize_kernels_and_is_bit_stable():
    """ss_config norm_fuse_fin: the apply kernels' own reduction of the statistics partials against the separate finalize launches
    (other chunk geometry, other fp64 summation order: equal to a few ulp), forward + backward, InstanceNorm over 3 groups and BatchNorm
    with moving statistics; two fused runs are bit-identical (fixed-order sums, no atomics)."""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    for kind, c, shp, use_res in (("instance", 256, (3, 40, 40), True), ("batch", 51, (2, 64, 64), False), ("batch", 64, (1, 128, 128), False)):
        n, h, w = shp
        x_cpu = torch.randn((n, h, w, c), generator=g) * 2 + 0.3
        r_cpu = torch.randn((n, h, w, c), generator=g) if use_res else None
        gy_cpu = torch.randn((n, h, w, c), generator=g)
        runs = []
        for fuse in (0, 3, 3):
            with L.config(norm_fuse_fin=fuse):
                arena = E.ParamArena(dev)
                layer = LY.Norm(arena, "n", c, kind)
                arena.materialize()
                arena["n/gamma"].copy_(torch.linspace(0.5, 1.5, c))
                arena["n/beta"].copy_(torch.linspace(-0.3, 0.3, c))
                if kind == "batch":
                    arena["n/moving_variance"].fill_(1.0)
                tape = E.Tape()
                x = E.Act(x_cpu.to(dev))
                res = E.Act(r_cpu.to(dev)) if use_res else None
                y = layer(tape, x, act=None if use_res else "relu", residual=res)
                gt, _ = y.grad_target()
                gt.t.copy_(gy_cpu.to(dev))
                arena.zero_grad()
                tape.backward()
                out = [y.dense().cpu(), x.get_grad().dense().cpu(), arena.grad("n/gamma").cpu().clone(), arena.grad("n/beta").cpu().clone()]
                if kind == "batch":
                    out += [arena["n/moving_mean"].cpu().clone(), arena["n/moving_variance"].cpu().clone()]
                runs.append(out)
        for a, b in zip(runs[1], runs[2]):
            assert torch.equal(a, b), "fused finalize is not bit-stable"
        for i, (a, b) in enumerate(zip(runs[0], runs[1])):
            assert float((a - b).abs().max()) <= 4e-6 * max(float(a.abs().max()), 1.0), (kind, c, i, float((a - b).abs().max()))


@pytest.mark.parametrize("n,hw,c,act,use_res", [(3, 128, 64, "relu", False), (2, 128, 256, None, True), (4, 64, 96, "lrelu", False), (2, 100, 64, "relu", False)])
def test_instance_norm_backward_in_one_pass_agrees_with_the_two_pass_form(n, hw, c, act, use_res):
    """ss_config norm_bwd_resident: the register-resident one-pass InstanceNorm backward (group-local barrier between statistics and
    apply) against the statistics + apply kernels and against float64: dx, the residual branch's gradient, dgamma, dbeta; two runs are
    bit-identical (partials added in piece order); no workgroup ever gave up at its barrier."""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 1000 + c)
    x_cpu = torch.randn((n, hw, hw, c), generator=g) * 1.5 + 0.2
    r_cpu = torch.randn((n, hw, hw, c), generator=g) if use_res else None
    gy_cpu = torch.randn((n, hw, hw, c), generator=g)
    gam, bet = torch.linspace(0.5, 1.5, c), torch.linspace(-0.3, 0.3, c)
    runs = []
    for mode in (0, 1, 1):
        with L.config(norm_bwd_resident=mode):
            arena = E.ParamArena(dev)
            layer = LY.Norm(arena, "n", c, "instance")
            arena.materialize()
            arena["n/gamma"].copy_(gam)
            arena["n/beta"].copy_(bet)
            tape = E.Tape()
            x = E.Act(x_cpu.to(dev))
            res = E.Act(r_cpu.to(dev)) if use_res else None
            y = layer(tape, x, act=act, act_alpha=0.2, residual=res)
            gt, _ = y.grad_target()
            gt.t.copy_(gy_cpu.to(dev))
            arena.zero_grad()
            lib = L.load()
            lib.ss_prof_reset(); lib.ss_prof_enable(1)
            tape.backward()
            torch.cuda.synchronize()
            lib.ss_prof_enable(0)
            assert ("norm_bwd_resident_kernel" in L.prof_summary()) == bool(mode), sorted(L.prof_summary())
            out = [x.get_grad().dense().cpu(), arena.grad("n/gamma").cpu().clone(), arena.grad("n/beta").cpu().clone()]
            if use_res:
                out.append(res.get_grad().dense().cpu())
            runs.append(out)
    assert L.load().ss_norm_resident_timeouts() == 0
    for a, b in zip(runs[1], runs[2]):
        assert torch.equal(a, b), "the one-pass backward is not bit-stable"
    # float64 truth
    xr = x_cpu.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    mu = xr.mean(dim=(1, 2), keepdim=True)
    var = ((xr - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    z = (xr - mu) / torch.sqrt(var + 1e-5) * gr + br
    if use_res:
        rr = r_cpu.double().requires_grad_(True)
        z = z + rr
    yr = {"relu": torch.relu, "lrelu": lambda t: torch.where(t > 0, t, 0.2 * t), None: lambda t: t}[act](z)
    yr.backward(gy_cpu.double())
    truth = [xr.grad, gr.grad, br.grad] + ([rr.grad] if use_res else [])
    for i, (a, b, t) in enumerate(zip(runs[0], runs[1], truth)):
        scale = max(float(t.abs().max()), 1e-30)
        e2, e1 = float((a.double() - t).abs().max()) / scale, float((b.double() - t).abs().max()) / scale
        print(f"tensor {i}: two-pass {e2:.2e} one-pass {e1:.2e} of max|truth|")
        assert e1 <= max(2.0 * e2, 2e-6), (i, e1, e2)


def test_maxpool_fwd_bwd():
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x_cpu = torch.randint(0, 4, (2, 8, 12, 5), generator=g).float()  # ties on purpose
    xr = x_cpu.clone().requires_grad_(True)
    yr = O.max_pool2x2(xr)
    gy = torch.rand(yr.shape, generator=g)
    yr.backward(gy)
    tape = E.Tape()
    x = E.Act(x_cpu.to(dev))
    y = LY.maxpool2x2(tape, x)
    assert torch.equal(y.dense().cpu(), yr.detach())
    gt, _ = y.grad_target()
    gt.t.copy_(gy.to(dev))
    tape.backward()
    assert torch.equal(x.get_grad().dense().cpu(), xr.grad)


def test_sync_batchnorm_two_phase_equals_whole_batch():
    """Data-parallel BatchNorm: per-rank raw sums, summed (= the all-reduce), then finish per rank with the global count
    must reproduce whole-batch BatchNorm on the concatenated batch -- forward, dx, dgamma/dbeta (summed over ranks)."""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    c, n, h, w = 24, 4, 16, 12
    x_all = (torch.rand((n, h, w, c), generator=g) * 3 - 1).to(dev)
    gy_all = (torch.rand((n, h, w, c), generator=g) - 0.5).to(dev)

    def run(batches, hook_factory):
        outs, dxs, grads, mms = [], [], [], []
        pending = {}
        for r, (xb, gb) in enumerate(batches):
            arena = E.ParamArena(dev)
            layer = LY.Norm(arena, "n", c, "batch")
            arena.materialize()
            arena["n/gamma"].copy_(torch.linspace(0.5, 1.5, c))
            arena["n/beta"].copy_(torch.linspace(-0.2, 0.2, c))
            arena["n/moving_variance"].fill_(1.0)
            LY.SYNC_BN = hook_factory(r) if hook_factory else None
            tape = E.Tape()
            xa = E.Act(xb.contiguous())
            y = layer(tape, xa, act="relu")
            gt, _ = y.grad_target()
            gt.t.copy_(gb)
            arena.zero_grad()
            tape.backward()
            outs.append(y.dense()); dxs.append(xa.get_grad().dense())
            grads.append((arena.grad("n/gamma").clone(), arena.grad("n/beta").clone())); mms.append(arena["n/moving_mean"].clone())
        LY.SYNC_BN = None
        return outs, dxs, grads, mms

    ref_y, ref_dx, ref_g, ref_mm = run([(x_all, gy_all)], None)
    # "two ranks": the hook adds the other half's raw sums, computed on the fly with the same kernels
    halves = [(x_all[:2], gy_all[:2]), (x_all[2:], gy_all[2:])]
    lib = L.load()
    import ctypes

    def hook_factory(rank):
        other_x, other_gy = halves[1 - rank]
        state = {"calls": 0, "mean": None, "rstd": None, "y": None}

        def hook(t):
            d = L.NormDesc(2, h, w, c, c, c, 0, 1, 1e-3, L.ACT_RELU, 0.0)
            ws = E.workspace(lib.ss_norm_workspace_bytes(ctypes.byref(d)), dev)
            osums = torch.empty_like(t)
            ox = other_x.contiguous()
            if state["calls"] == 0:      # forward statistics of the other rank
                L.check(lib.ss_norm_fwd_stats(ctypes.byref(d), ox.data_ptr(), osums.data_ptr(), ws.data_ptr(), ws.numel(), E._stream()), "s")
                t += osums
                # remember the other rank's forward (global stats) for its backward sums
                tot = t.clone()
                mean, rstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
                yo = torch.empty_like(ox)
                gam = torch.linspace(0.5, 1.5, c).to(dev); bet = torch.linspace(-0.2, 0.2, c).to(dev)
                L.check(lib.ss_norm_fwd_finish(ctypes.byref(d), ox.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, yo.data_ptr(),
                                               tot.data_ptr(), 4 * h * w, mean.data_ptr(), rstd.data_ptr(), None, None, 0.99, E._stream()), "f")
                state.update(mean=mean, rstd=rstd, y=yo)
            else:                        # backward statistics of the other rank
                og = other_gy.contiguous()
                L.check(lib.ss_norm_bwd_stats(ctypes.byref(d), og.data_ptr(), c, ox.data_ptr(), state["y"].data_ptr(), state["mean"].data_ptr(),
                                              state["rstd"].data_ptr(), osums.data_ptr(), ws.data_ptr(), ws.numel(), E._stream()), "bs")
                t += osums
            state["calls"] += 1
            return 2
        return hook

    ys, dxs, gs, mms = run(halves, hook_factory)
    assert_close(torch.cat(ys).cpu(), ref_y[0].cpu(), "syncbn y", rtol=1e-5)
    assert_close(torch.cat(dxs).cpu(), ref_dx[0].cpu(), "syncbn dx", rtol=1e-4)
    assert_close((gs[0][0] + gs[1][0]).cpu(), ref_g[0][0].cpu(), "syncbn dgamma", rtol=1e-4)
    assert_close((gs[0][1] + gs[1][1]).cpu(), ref_g[0][1].cpu(), "syncbn dbeta", rtol=1e-4)
    assert_close(mms[0].cpu(), ref_mm[0].cpu(), "syncbn moving_mean", rtol=1e-5)


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0].startswith("wino_")], ids=[c[0] for c in CONV_CASES if c[0].startswith("wino_")])
def test_conv_split_bf16_opt_in(case):
    """Opt-in SS_ALGO_BF16X3: Winograd GEMMs as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (16 mantissa bits per operand,
    amplified by the F(4x4,3x3) output transform): measured rel-L2 ~5e-5, max error ~1e-3 of max|ref| -> tolerance 2e-3.
    That is 10x looser than the fp32 parity bar, which is why this mode is NOT the default.  The weight gradient stays fp32."""
    E, LY, L = _mods()
    name, k, cin, cout, stride, padding, bias, act, transposed, n, h, w = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    arena = E.ParamArena(dev)
    layer = LY.Conv2D(arena, "c", k, cin, cout, stride=stride, padding=padding, use_bias=bias, act=act, act_alpha=0.2, algo=L.ALGO_BF16X3)
    arena.materialize()
    w_cpu = (torch.rand((k, k, cin, cout), generator=g, dtype=torch.float64) - 0.5) * 0.5
    b_cpu = torch.rand(cout, generator=g, dtype=torch.float64) - 0.5 if bias else None
    x_cpu = torch.rand((n, h, w, cin), generator=g, dtype=torch.float64) * 2 - 1
    arena["c/kernel"].copy_(w_cpu.float())
    if bias:
        arena["c/bias"].copy_(b_cpu.float())
    xr = x_cpu.clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    yr = oracle_conv(xr, wr, b_cpu, k, stride, padding, act, transposed)
    gy = torch.rand(yr.shape, generator=g, dtype=torch.float64) - 0.5
    yr.backward(gy)
    tape = E.Tape()
    x = E.Act(x_cpu.float().to(dev), requires_grad=True)
    y = layer(tape, x)
    assert_close(y.dense().cpu(), yr.detach().float(), f"{name} bf16x3 fwd", rtol=2e-3)
    assert rel_l2(y.dense().cpu().numpy(), yr.detach().numpy()) <= 2e-4
    gt, _ = y.grad_target()
    gt.t.copy_(gy.float().to(dev))
    arena.zero_grad()
    tape.backward()
    assert_close(x.get_grad().dense().cpu(), xr.grad.float(), f"{name} bf16x3 dx", rtol=2e-3)
    assert rel_l2(x.get_grad().dense().cpu().numpy(), xr.grad.numpy()) <= 2e-4
    assert_close(arena.grad("c/kernel").cpu(), wr.grad.float(), f"{name} bf16x3 dw", rtol=2e-4)


X6_CASES = [
    # name, k, cin, cout, stride, padding, bias, act, transposed, n, h, w     (reduction channels % 32 == 0, >= 1024 output pixels)
    ("x6_disc_4x4_s2", 4, 128, 256, 2, "valid", False, None, False, 2, 66, 66),
    ("x6_down_3x3_s2", 3, 64, 128, 2, "same", False, None, False, 2, 64, 64),
    ("x6_up_T3", 3, 128, 64, 2, "same", False, None, True, 2, 32, 32),
    ("x6_trunk_wino", 3, 256, 256, 1, ("reflect", 1), False, None, False, 2, 48, 48),
    ("x6_3x3_same_bias_tanh", 3, 96, 40, 1, "same", True, "tanh", False, 1, 40, 36),
]


@pytest.mark.parametrize("case", X6_CASES, ids=[c[0] for c in X6_CASES])
def test_conv_x6_is_fp32_grade(case):
    """SS_ALGO_X6 (what AUTO picks for these shapes): fp32 operands split EXACTLY into three bf16 pieces, six piece products on
    the bf16 matrix cores, fp32 accumulation -- or, for the Winograd case with 128 / 256 / 512 channels (x6_trunk_wino), "x3h": two
    fp16 pieces with per-tile power-of-two scales, three products (csrc/gemm_x6p.hip).  Claim under test: the result is as close to the fp64 oracle as the
    v_mfma_f32_32x32x2_f32 path (SS_ALGO_MFMA) -- same 1e-4 parity bar, and rel-L2 error within 1.5x of the fp32-MFMA
    path's own rounding error (both are ~1e-7 .. 1e-6), for the output, the data gradient and the weight gradient.  With SS_X3H
    (default) the same kernels carry two fp16 pieces with one power-of-two scale per operand tensor (per tile for the Winograd
    forward / data-gradient GEMMs) and three products."""
    E, LY, L = _mods()
    name, k, cin, cout, stride, padding, bias, act, transposed, n, h, w = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    w_cpu = (torch.rand((k, k, cout, cin) if transposed else (k, k, cin, cout), generator=g, dtype=torch.float64) - 0.5) * 0.2
    b_cpu = torch.rand(cout, generator=g, dtype=torch.float64) - 0.5 if bias else None
    x_cpu = torch.rand((n, h, w, cin), generator=g, dtype=torch.float64) * 2 - 1
    xr = x_cpu.clone().requires_grad_(True)
    wr = w_cpu.clone().requires_grad_(True)
    yr = oracle_conv(xr, wr, b_cpu, k, stride, padding, act, transposed)
    gy = torch.rand(yr.shape, generator=g, dtype=torch.float64) - 0.5
    yr.backward(gy)
    errs = {}
    for algo in (L.ALGO_MFMA, L.ALGO_X6):
        arena = E.ParamArena(dev)
        layer = LY.Conv2D(arena, "c", k, cin, cout, stride=stride, padding=padding, use_bias=bias, act=act, act_alpha=0.2,
                          transposed=transposed, algo=algo)
        arena.materialize()
        arena["c/kernel"].copy_(w_cpu.float())
        if bias:
            arena["c/bias"].copy_(b_cpu.float())
        tape = E.Tape()
        x = E.Act(x_cpu.float().to(dev), requires_grad=True)
        y = layer(tape, x)
        gt, _ = y.grad_target()
        gt.t.copy_(gy.float().to(dev))
        arena.zero_grad()
        tape.backward()
        got_y, got_dx, got_dw = y.dense().cpu(), x.get_grad().dense().cpu(), arena.grad("c/kernel").cpu()
        assert_close(got_y, yr.detach(), f"{name}/{algo} fwd", rtol=1e-4)
        assert_close(got_dx, xr.grad, f"{name}/{algo} dx", rtol=2e-4)
        assert_close(got_dw, wr.grad, f"{name}/{algo} dw", rtol=2e-4)
        errs[algo] = (rel_l2(got_y.numpy(), yr.detach().numpy()), rel_l2(got_dx.numpy(), xr.grad.numpy()),
                      rel_l2(got_dw.numpy(), wr.grad.numpy()))
    print(f"{name}: rel-L2 vs fp64  fp32-MFMA y={errs[L.ALGO_MFMA][0]:.2e} dx={errs[L.ALGO_MFMA][1]:.2e} dw={errs[L.ALGO_MFMA][2]:.2e}   "
          f"x6/x3h y={errs[L.ALGO_X6][0]:.2e} dx={errs[L.ALGO_X6][1]:.2e} dw={errs[L.ALGO_X6][2]:.2e}")
    for i in (0, 1, 2):
        assert errs[L.ALGO_X6][i] <= 1.5 * errs[L.ALGO_MFMA][i] + 2e-7, errs


_X6P_FORCED = [c for c in CONV_CASES if c[0].startswith("wino_")] + [c for c in X6_CASES if c[0] == "x6_trunk_wino"]


@pytest.mark.parametrize("case", _X6P_FORCED, ids=[c[0] for c in _X6P_FORCED])
def test_x6p_forced_on_small_shapes(case):
    """The pre-split-plane GEMM (csrc/gemm_x6p.hip) is only chosen for launches of >= 1024 workgroups (tests/test_fullsize_gpu.py
    runs it at the real trunk shape); x6p = 2 (SS_X6P=force) sends the small Winograd cases of this file through it too: ragged M
    (tiles not a multiple of 256) and N (channels not a multiple of 128) tile edges -- the same bodies and tolerances as the default
    route's tests."""
    E, LY, L = _mods()
    with L.config(x6p=2):
        if case[0] == "x6_trunk_wino":
            test_conv_x6_is_fp32_grade(case)
        else:
            test_conv_fwd_bwd(case, "auto")


@pytest.mark.parametrize("k2", [-24, -9, 13])
def test_x3h_power_of_two_invariance(k2):
    """x3h (fp16 two-piece operands of the Winograd GEMMs) picks one power-of-two scale per tile from the data, so scaling the
    input -- or the weights -- by 2^k must scale the result by exactly 2^k, bit for bit: gradients of any magnitude (1e-7 ... 1e4)
    get the same relative accuracy as O(1) activations.  (An fp16 path WITHOUT the dynamic scales fails this at 2^-24.)"""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    cin = cout = 256
    w_cpu = (torch.rand((3, 3, cin, cout), generator=g) - 0.5) * 0.2
    x_cpu = torch.rand((2, 48, 48, cin), generator=g) * 2 - 1
    outs = []
    for sx, sw in ((0, 0), (k2, 0), (0, k2)):
        arena = E.ParamArena(dev)
        layer = LY.Conv2D(arena, "c", 3, cin, cout, stride=1, padding=("reflect", 1), use_bias=False)
        arena.materialize()
        arena["c/kernel"].copy_(w_cpu * (2.0 ** sw))
        x = E.Act((x_cpu * (2.0 ** sx)).to(dev), requires_grad=True)
        tape = E.Tape()
        y = layer(tape, x)
        gt, _ = y.grad_target()
        gt.t.copy_(torch.ones_like(gt.t) * 0.25)
        arena.zero_grad()
        tape.backward()
        outs.append((y.dense().cpu(), x.get_grad().dense().cpu(), arena.grad("c/kernel").cpu().clone()))
    y0, dx0, dw0 = outs[0]
    assert torch.isfinite(y0).all() and float(y0.abs().max()) > 0
    assert torch.equal(outs[1][0], y0 * (2.0 ** k2)), "forward is not exactly homogeneous in the input"
    assert torch.equal(outs[2][0], y0 * (2.0 ** k2)), "forward is not exactly homogeneous in the weights"
    assert torch.equal(outs[2][1], dx0 * (2.0 ** k2)), "data gradient is not exactly homogeneous in the weights"
    assert torch.equal(outs[1][2], dw0 * (2.0 ** k2)), "weight gradient is not exactly homogeneous in the input"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 32, 32, 256, 256), (8, 32, 32, 512, 256), (1, 64, 64, 256, 128)], ids=["n4_32_256", "n8_32_512to256", "n1_64_256to128"])
def test_winograd_weight_gradient_from_the_forward_passs_saved_operand(shape):
    """ss_conv_desc::saved_operand: the Winograd x3h forward keeps its transformed input planes (per-tile scales), the weight gradient
    contracts them with the transformed dy (rows scaled by the tiles' inverse factors) instead of transforming x again.  Both routes
    are the x3h arithmetic: each must match the fp64 gradient as closely as the other (fp32-grade), and they must agree with each
    other far below that distance; the forward output and dx do not depend on the switch at all."""
    import ctypes
    E, LY, L = _mods()
    n, h, w, cin, cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    wt = torch.empty((3, 3, cin, cout)).uniform_(-0.05, 0.05, generator=g)
    # heavy-tailed input: a few tiles two orders of magnitude above the rest (per-tile scales vs one scale per tensor)
    xt = torch.randn((n, h, w, cin), generator=g)
    xt[:, : h // 4, : w // 4] *= 60.0
    dyt = torch.randn((n, h, w, cout), generator=g)
    res = {}
    lib = L.load()
    for save in (False, True):
        LY.SAVE_OPERAND = save
        try:
            with L.config(x6p=2):
                arena = E.ParamArena(dev)
                conv = LY.Conv2D(arena, "c", 3, cin, cout, padding=("reflect", 1), use_bias=False)
                arena.materialize()
                arena["c/kernel"].copy_(wt)
                tape = E.Tape()
                x = E.Act(xt.to(dev), requires_grad=True)
                y = conv(tape, x)
                if save:
                    assert lib.ss_conv2d_saved_operand_bytes(ctypes.byref(conv.desc(x, y))) > 0, "this shape must take the saved-operand path"
                gt, _ = y.grad_target()
                gt.t.copy_(dyt.to(dev))
                arena.zero_grad()
                tape.backward()
                torch.cuda.synchronize()
                res[save] = (y.dense().clone(), x.get_grad().dense().clone(), arena.grad("c/kernel").clone())
        finally:
            LY.SAVE_OPERAND = True
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    # fp64 reference of dw: correlation of the reflect-padded input with dy
    xp = torch.nn.functional.pad(xt.double().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")
    dyp = dyt.double().permute(0, 3, 1, 2)
    ref = torch.empty((3, 3, cin, cout), dtype=torch.float64)
    for a in range(3):
        for b in range(3):
            ref[a, b] = torch.einsum("nchw,nkhw->ck", xp[:, :, a:a + h, b:b + w], dyp)
    scale = ref.abs().max()
    e_off = float((res[False][2].double().cpu() - ref).abs().max() / scale)
    e_on = float((res[True][2].double().cpu() - ref).abs().max() / scale)
    d_on_off = float((res[True][2] - res[False][2]).abs().max().cpu() / scale)
    assert e_off <= 2e-5 and e_on <= 2e-5, (e_off, e_on)
    assert e_on <= 3.0 * e_off + 1e-7, (e_on, e_off)
    assert d_on_off <= 2e-5, d_on_off


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 40, 256, 256), (5, 28, 28, 64, 320), (1, 64, 64, 256, 128), (3, 36, 36, 128, 192)],
                         ids=["ragged_M", "K64_N320", "n1", "N192"])
def test_plane_gemm_schedules_are_bit_identical(shape):
    """The x3h Winograd GEMMs exist in several SCHEDULES of the same arithmetic -- fragment reads interleaved with the MFMAs or in
    bursts (gemm_ilv), the opt-in ping-pong kernels (x6p_pp = 1 / 2 / 3), the persistent grid on a CU subset (gemm_cus) -- with the same
    MFMA order per accumulator: forward output, data gradient and weight gradient must not change by one bit (ragged M, N not a
    multiple of 128, K = 64 = two chunks, one sample)."""
    E, LY, L = _mods()
    n, h, w, cin, cout = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 100 + cin)
    wt = torch.empty((3, 3, cin, cout)).uniform_(-0.05, 0.05, generator=g)
    xt = torch.randn((n, h, w, cin), generator=g)
    dyt = torch.randn((n, h, w, cout), generator=g)

    def run(**cfg):
        with L.config(x6p=2, **cfg):
            arena = E.ParamArena(dev)
            conv = LY.Conv2D(arena, "c", 3, cin, cout, padding=("reflect", 1) if h % 8 == 0 else "same", use_bias=False)
            arena.materialize()
            arena["c/kernel"].copy_(wt)
            tape = E.Tape()
            x = E.Act(xt.to(dev), requires_grad=True)
            y = conv(tape, x)
            gt, _ = y.grad_target()
            gt.t.copy_(dyt.to(dev))
            arena.zero_grad()
            tape.backward()
            torch.cuda.synchronize()
            return y.dense().clone(), x.get_grad().dense().clone(), arena.grad("c/kernel").clone()

    base = run(gemm_ilv=0, x6p_pp=0)
    for cfg in (dict(gemm_ilv=1), dict(x6p_pp=1), dict(x6p_pp=2), dict(x6p_pp=3), dict(gemm_cus=64), dict(gemm_ilv=1, gemm_cus=8)):
        got = run(**cfg)
        for a, b, what in zip(base, got, ("y", "dx", "dw")):
            assert torch.equal(a, b), (cfg, what, float((a - b).abs().max()))


@pytest.mark.gpu
def test_probe_mfma_reports_a_real_data_ceiling_below_the_zero_operand_rate():
    """ss_probe_mfma (bench.py roofline.real_data_ceiling): a register-only fp16 MFMA stream; on random operands the chip clocks lower
    than on zeros (power budget), both within the physical range of the device."""
    import ctypes
    E, LY, L = _mods()
    lib = L.load()
    sc = torch.empty(65600, dtype=torch.uint8, device="cuda:0")
    out = {}
    for rnd in (0, 1):
        tf, mhz = ctypes.c_double(0.0), ctypes.c_double(0.0)
        assert lib.ss_probe_mfma(rnd, sc.data_ptr(), sc.numel(), None, ctypes.byref(tf), ctypes.byref(mhz)) == 0
        out[rnd] = (tf.value, mhz.value)
    assert 500.0 < out[1][0] <= out[0][0] * 1.02 and out[0][0] < 2700.0, out
    assert 800.0 < out[1][1] < 2600.0 and 800.0 < out[0][1] < 2600.0, out
    assert lib.ss_probe_mfma(1, sc.data_ptr(), 100, None, ctypes.byref(tf), ctypes.byref(mhz)) != 0          # scratch too small: refused


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("up_T3_128_64", 3, 128, 64, 2, "same", True, 8, 128, 128, "fwd"), ("up_T3_256_128_ragged", 3, 256, 128, 2, "same", True, 10, 72, 90, "fwd"),
                                  ("down_s2_64_128_dgrad", 3, 64, 128, 2, "same", False, 8, 256, 256, "dgrad"),
                                  ("down_s2_odd_dgrad", 3, 128, 256, 2, "same", False, 3, 131, 125, "dgrad"), ("small_up_T3", 3, 64, 64, 2, "same", True, 1, 36, 40, "fwd"),
                                  ("disc_4x4_valid_odd_dgrad", 4, 128, 256, 2, "valid", False, 6, 255, 255, "dgrad"),
                                  ("disc_4x4_valid_even_dgrad", 4, 256, 512, 2, "valid", False, 16, 126, 126, "dgrad"),
                                  ("down_s2_dgrad_accumulated", 3, 64, 128, 2, "same", False, 2, 96, 80, "dgrad_acc")],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("split", [1, 0], ids=["two_phases_per_wave_pair", "four_phases_per_wave"])
def test_fused_subpixel_phases_agree_with_one_launch_per_phase(case, split):
    """gconv_phases_fused_kernel (conv_phase.hip: the four sub-pixel phases of a stride-2 data gradient / transposed convolution in one
    workgroup per input tile) against one gather launch per phase: the same x3h pieces and products in another K order, so the two
    agree to fp32 rounding (rel-L2 <= 2e-6, max |d| <= 1e-5 max|ref|) -- transposed forward, strided data gradients, odd output sizes
    (class grids that differ by one between the phases), the discriminators' 4x4 layers (16 taps), ragged tile edges.  Both forms of the
    kernel: two phases per wave pair with the interleaved weight stream (`phases_split`, the default) and all four phases in every wave."""
    E, LY, L = _mods()
    name, k, cin, cout, stride, padding, transposed, n, h, w, which = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    wt = torch.empty((k, k, cout, cin) if transposed else (k, k, cin, cout)).uniform_(-0.05, 0.05, generator=g)
    xt = torch.randn((n, h, w, cin), generator=g)

    def run(fused):
        with L.config(phases_fused=fused, phases_split=split):
            arena = E.ParamArena(dev)
            conv = LY.Conv2D(arena, "c", k, cin, cout, stride=stride, padding=padding, use_bias=False, transposed=transposed)
            arena.materialize()
            arena["c/kernel"].copy_(wt)
            tape = E.Tape()
            tape.param_grads = False
            x = E.Act(xt.to(dev), requires_grad=(which != "fwd"))
            y = conv(tape, x)
            if which == "fwd":
                torch.cuda.synchronize()
                return y.dense().clone()
            gt, _ = y.grad_target()
            gt.t.copy_(torch.randn(tuple(gt.t.shape), generator=torch.Generator().manual_seed(5)).to(dev))
            tape.backward()
            if which == "dgrad_acc":          # a second consumer of x: its data gradient ACCUMULATES into the first one's (generic epilogue)
                tape2 = E.Tape()
                tape2.param_grads = False
                y2 = conv(tape2, x)
                g2, _ = y2.grad_target()
                g2.t.copy_(torch.randn(tuple(g2.t.shape), generator=torch.Generator().manual_seed(6)).to(dev))
                tape2.backward()
            torch.cuda.synchronize()
            return x.get_grad().dense().clone()

    ref, got = run(0), run(1)
    assert float(ref.abs().max()) > 0
    assert not torch.equal(ref, got), "the fused kernel did not take this problem (another K order cannot give the same bits)"
    assert rel_l2(got.cpu(), ref.cpu()) <= 2e-6
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.gpu
def test_fused_subpixel_phases_vs_oracle():
    """The same kernel against the fp32 CPU oracle (Conv2DTranspose 3x3 stride 2 with bias and leaky ReLU, 128 -> 64 channels, 8 x 128 x 128):
    forward within the layer tolerance of this file (1e-4 of max|ref|)."""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    wt = torch.empty((3, 3, 64, 128)).uniform_(-0.05, 0.05, generator=g)
    bt = torch.empty(64).uniform_(-0.1, 0.1, generator=g)
    xt = torch.randn((8, 128, 128, 128), generator=g)
    arena = E.ParamArena(dev)
    conv = LY.Conv2D(arena, "c", 3, 128, 64, stride=2, padding="same", use_bias=True, transposed=True, act="lrelu", act_alpha=0.2)
    arena.materialize()
    arena["c/kernel"].copy_(wt)
    arena["c/bias"].copy_(bt)
    y = conv(E.Tape(enabled=False), E.Act(xt.to(dev), requires_grad=False)).dense().cpu()
    ref = torch.nn.functional.leaky_relu(O.conv2d_transpose(xt, wt, bt, 2), 0.2)
    assert_close(y, ref, "fused sub-pixel transposed convolution")


@pytest.mark.parametrize("shape", [(4, 64, 64, 4), (1, 128, 128, 256), (2, 96, 96, 17)], ids=["in_c4_g4", "in_c256", "bn_c17"])
def test_norm_statistics_are_exact_on_constant_channels(shape):
    """A constant input image stays constant per channel through the whole generator: every InstanceNorm then sees groups of identical
    values, beta starts at 0 and the ReLU behind it turns the SIGN of a rounding error in the mean into a whole channel's mask (the
    reference-generated CycleGAN vectors contain such tiles: 31 -> 5 127 flipped Adam steps when the mean of 4 096 identical values was
    summed as x, 2x, 3x, ...).  The fused-finalize FORWARD statistics (opt-in, norm_fuse_fin bit 0) are sums of x - x[first pixel]: mean = x and variance = 0 exactly."""
    E, LY, L = _mods()
    dev = torch.device("cuda:0")
    n, h, w, c = shape
    kind = "batch" if c == 17 else "instance"
    vals = torch.linspace(-1.37, 2.11, c) * 0.7312345
    x_cpu = vals.expand(n, h, w, c).contiguous()
    arena = E.ParamArena(dev)
    layer = LY.Norm(arena, "n", c, kind)
    arena.materialize()
    arena["n/gamma"].fill_(1.0)
    arena["n/beta"].zero_()
    with L.config(norm_fuse_fin=3):          # forward fusion is opt-in (bit 0)
        tape = E.Tape()
        x = E.Act(x_cpu.to(dev))
        y = layer(tape, x, act="relu")
        assert float(y.dense().abs().max()) == 0.0, "relu(gamma * (x - mean) * rstd + 0) of a constant channel must be exactly 0"
        gt, _ = y.grad_target()
        gt.t.copy_(torch.randn((n, h, w, c), generator=torch.Generator().manual_seed(1)).to(dev))
        arena.zero_grad()
        tape.backward()
        assert float(x.get_grad().dense().abs().max()) == 0.0 and float(arena.grad("n/gamma").abs().max()) == 0.0      # relu'(0) = 0: nothing flows


@pytest.mark.parametrize("case", [("down_3x3", 3, 64, 128, False, (4, 128, 256)), ("patchgan_4x4", 4, 64, 128, False, (8, 128, 128)),
                                  ("up_T3x3", 3, 128, 64, True, (4, 64, 128)), ("patchgan_valid_ragged", 4, 64, 128, False, (8, 130, 126))],
                         ids=lambda c: c[0])
def test_staged_weight_gradient_vs_float64_and_the_gather_kernel(case):
    """conv_wgrad_stage.hip (stride-2 layers: operands staged once per spatial tile, taps from LDS with transposing reads) against a
    float64 convolution's weight gradient and against wgrad_x6_kernel (ss_config wgrad_stage = 0: the same x3h arithmetic in another
    summation order); two runs are bit-identical (fixed-order reduction of the splits' partials)."""
    E, LY, L = _mods()
    name, k, cin, cout, tr, (n, h, w) = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    arena = E.ParamArena(dev)
    pad = "valid" if name.endswith("ragged") else "same"          # 'valid' PatchGAN layers: 64 x 62 output pixels, tiles hang over the grid
    conv = LY.Conv2D(arena, "c", k, cin, cout, stride=2, padding=pad, transposed=tr)
    arena.materialize()
    arena["c/kernel"].normal_(0, 0.05)
    x_cpu = torch.randn((n, h, w, cin), generator=g) * 0.8
    oh, ow = conv.out_hw(h, w)
    gy_cpu = torch.randn((n, oh, ow, cout), generator=g) * 0.2
    grads = {}
    for mode in (1, 1, 0):
        with L.config(wgrad_stage=mode):
            tape = E.Tape()
            x = E.Act(x_cpu.to(dev), requires_grad=True)
            y = conv(tape, x)
            gt, _ = y.grad_target()
            gt.t.copy_(gy_cpu.to(dev))
            y.grad_init = True
            arena.zero_grad()
            tape.backward()
            torch.cuda.synchronize()
            grads.setdefault(mode, []).append(arena.grad("c/kernel").cpu().clone())
    assert torch.equal(grads[1][0], grads[1][1]), "staged weight gradient is not bit-stable"
    # float64 truth through torch's CPU convolutions (Keras layouts: Conv2D (kh, kw, cin, cout), Conv2DTranspose (kh, kw, cout, cin))
    wk = arena["c/kernel"].cpu().double()
    xc = x_cpu.double().permute(0, 3, 1, 2)
    if not tr:
        wt = wk.permute(3, 2, 0, 1).clone().requires_grad_(True)
        pt, pl = (LY.same_pad(h, k, 2)[0], LY.same_pad(w, k, 2)[0]) if pad == "same" else (0, 0)
        yy = torch.nn.functional.conv2d(torch.nn.functional.pad(xc, (pl, k, pt, k)), wt, stride=2)[:, :, :oh, :ow]
    else:
        wt = wk.permute(3, 2, 0, 1).clone().requires_grad_(True)          # conv_transpose2d weight: (in, out, kh, kw)
        yy = torch.nn.functional.conv_transpose2d(xc, wt, stride=2, padding=1, output_padding=1)
    (yy * gy_cpu.double().permute(0, 3, 1, 2)).sum().backward()
    truth = wt.grad.permute(2, 3, 1, 0) if not tr else wt.grad.permute(2, 3, 1, 0)
    ref = float(truth.abs().max())
    e_stage = float((grads[1][0].double() - truth).abs().max()) / ref
    e_x6 = float((grads[0][0].double() - truth).abs().max()) / ref
    assert e_stage <= 1.5 * e_x6 + 2e-7 and e_stage <= 2e-6, (name, e_stage, e_x6)
