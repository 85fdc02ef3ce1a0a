"""CycleGAN workflow, model and image buffer with the reference's class / attribute / metric names
(Releases/Version 1.2.0/CycleGAN.py), executing on libsemseg_hip.so.

Reference surface mirrored here (file:line in the reference):
* ``CycleGAN`` ctor + attribute defaults ............ CycleGAN.py:20-114   (StartProcess.py:91-102 overrides)
* ``create_model`` / ``start_training`` .............. CycleGAN.py:116-222
* ``generator_loss_fn`` / ``discriminator_loss_fn`` .. CycleGAN.py:301-308  (LSGAN, label smoothing)
* ``linear_decay`` .................................. CycleGAN.py:310-317
* ``DataLoader`` .................................... CycleGAN.py:454-479
* ``CycleGanModel`` (+ ``compile``, ``train_step``) .. CycleGAN.py:512-710  (torch-backend semantics)
* ``ImagePool`` ..................................... CycleGAN.py:908-964

Behavioural quirks kept on purpose (SURVEY.md "Three facts"):
* the pools are built in ``__init__`` with the constructor default ``batch_size = 2`` and never see the
  later ``cycle_gan.batch_size = N`` assignment, so ``query`` only ever uses the first two images of a batch;
* both generator losses are back-propagated into BOTH generators (``retain_graph`` double backward without
  zeroing, CycleGAN.py:664-665): implemented as ONE backward of (L_a + L_b) -- same gradient, 6 generator
  backward traversals instead of 8.
"""
import ctypes
import json
import os
import random
import time

import numpy as np
import torch

from . import HelperFunctions
from . import _lib as L
from . import dist as D
from . import layers as LY
from . import losses
from .engine import Act, Tape, _stream, cat_batch
from .nets import PatchDiscriminator, ResnetGenerator
from .optim import Adam

METRIC_NAMES = ("d_a", "d_b", "d_fake_a", "d_fake_b", "d_real_a", "d_real_b", "g_a", "g_b",
                "g_adv_a", "g_adv_b", "g_cyc_a", "g_cyc_b", "g_id_a", "g_id_b")


class ImagePool:
    """History buffer of generated images (CycleGAN.py:908-964), device resident.  The DECISIONS are taken exactly as the reference
    takes them -- same control flow, same consumption of python's ``random`` stream: one ``uniform`` (+ one ``randint`` on a swap)
    per image once the buffer is full, the loop bound frozen at the constructor's ``batch_size`` -- and the copies they imply run as
    ONE launch per query (``ss_pool_query``, include/semseg_hip.h) on a single [pool_size, H, W, C] buffer."""

    def __init__(self, batch_size, pool_size=50, rng=random):
        self.pool_size = pool_size
        self.batch_size = batch_size
        self.rng = rng
        if self.pool_size > 0:
            self.num_imgs = 0
            self._buf = None          # [pool_size, H, W, C], allocated at the first query

    @property
    def images(self):
        """The stored images, slot by slot (the reference keeps a python list of [1,H,W,C] tensors)."""
        return [] if self._buf is None else [self._buf[i:i + 1] for i in range(self.num_imgs)]

    def query(self, images):
        """images: NHWC device tensor.  Returns a new [k,H,W,C] tensor, k = min(self.batch_size, N)."""
        if self.pool_size == 0:
            return images
        assert images.is_contiguous()
        if self._buf is None:
            self._buf = torch.empty((self.pool_size,) + tuple(images.shape[1:]), dtype=images.dtype, device=images.device)
        assert tuple(images.shape[1:]) == tuple(self._buf.shape[1:]) and images.dtype == self._buf.dtype, "one buffer holds images of one shape"
        modes, slots = [], []
        for index in range(0, self.batch_size):
            if index >= images.shape[0]:
                break  # short batch: the reference's documented intent (CycleGAN.py:945-948)
            if self.num_imgs < self.pool_size:
                modes.append(L.POOL_FILL); slots.append(self.num_imgs)
                self.num_imgs += 1
            else:
                p = self.rng.uniform(0, 1)
                if p > 0.5:
                    modes.append(L.POOL_SWAP); slots.append(self.rng.randint(0, self.pool_size - 1))
                else:
                    modes.append(L.POOL_PASS); slots.append(0)
        k = len(modes)
        out = torch.empty((k,) + tuple(images.shape[1:]), dtype=images.dtype, device=images.device)
        lib = L.load()
        per = images[0].numel() * images.element_size()
        # one launch per query; a query whose images name the same slot twice is issued image by image (the second swap must see
        # the first one's store)
        named = [s_ for m, s_ in zip(modes, slots) if m != L.POOL_PASS]
        groups = [(0, k)] if len(set(named)) == len(named) and k <= L.POOL_MAX_QUERY else [(i, i + 1) for i in range(k)]
        for a, b in groups:
            n = b - a
            marr = (ctypes.c_int32 * n)(*modes[a:b])
            sarr = (ctypes.c_int32 * n)(*slots[a:b])
            L.check(lib.ss_pool_query(self._buf.data_ptr(), images[a:].data_ptr(), out[a:].data_ptr(), per, n, marr, sarr, self.pool_size,
                                      _stream()), "ss_pool_query")
        return out


class CycleGanModel:
    """Drop-in for the reference's ``CycleGanModel(keras.Model)``: same constructor arguments, ``compile``
    keywords, ``gen_a/gen_b/disc_a/disc_b`` attributes and ``train_step`` metric keys."""

    def __init__(self, generator_a, generator_b, discriminator_a, discriminator_b, image_pool_a=None,
                 image_pool_b=None, lambda_cycle_a=10.0, lambda_cycle_b=10.0, lambda_identity_a=0.5,
                 lambda_identity_b=0.5, **kwargs):
        self.gen_a, self.gen_b = generator_a, generator_b
        self.disc_a, self.disc_b = discriminator_a, discriminator_b
        self.lambda_cycle_a, self.lambda_cycle_b = lambda_cycle_a, lambda_cycle_b
        self.lambda_identity_a, self.lambda_identity_b = lambda_identity_a, lambda_identity_b
        self.use_identity_loss = lambda_identity_a > 0 or lambda_identity_b > 0
        # translation + identity pass of each generator as ONE pass over the concatenated batch (same maths per sample)
        self.batch_generator_passes = os.environ.get("SS_BATCH_G_PASSES", "1") != "0"
        # run the two independent chains of each phase on two HIP streams (see _train_step_dual); SS_DUAL_STREAM=0 disables
        self.dual_stream = {"0": False, "force": "force"}.get(os.environ.get("SS_DUAL_STREAM", "1"), True)
        self.dual_gemm_cus = 192          # CUs the persistent GEMMs of one chain occupy while two chains run (0: all; 224 before round 6: 179.0 -> 178.2 ms per step)
        # weight gradients of the generator chains on two further streams: measured SLOWER (164.8 vs 157.2 ms for the CycleGAN step -- four
        # chains already share the chip and the weight-gradient GEMMs take whole CUs); the MultiResUNet step, one chain, gains 6 % from it
        self.wgrad_side_streams = os.environ.get("SS_WGRAD_STREAMS", "0") == "1"
        self.refresh_side_streams = os.environ.get("SS_REFRESH_STREAMS", "1") != "0"
        env = os.environ.get("SS_CG_STREAMS")
        self.stream_indices = [int(v) for v in env.split(",")] if env else None
        # (switched off automatically when several ranks share one GPU -- dist.ranks_share_device(); "force" overrides, for tests)
        self.gen_a_optimizer = self.gen_b_optimizer = self.disc_a_optimizer = self.disc_b_optimizer = None
        self.image_pool_a = image_pool_a if image_pool_a is not None else ImagePool(1, 0)
        self.image_pool_b = image_pool_b if image_pool_b is not None else ImagePool(1, 0)
        self.label_smoothing_factor = 0.0
        self.device = generator_a.device
        # mixed precision: the networks' activation storage type (float32 by default) and a static loss scale for float16 storage
        # (gradients below 6e-8 vanish in fp16): every loss gradient is multiplied by it, the optimizer steps divide it out
        self.act_dtype = generator_a.act_dtype
        self.loss_scale = 1024.0 if self.act_dtype == torch.float16 else 1.0
        # 14 running means (keras.metrics.Mean, CycleGAN.py:547-560): device sums, one host read per query
        self._scalars = torch.zeros(16, dtype=torch.float32, device=self.device)
        self._bce3 = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._sums = np.zeros(14, dtype=np.float64)
        self._count = 0
        self.sync_metrics = True
        self.use_binary_crossentropy_a = False   # cycle/identity loss A = BinaryCrossentropy (generator A ends in a sigmoid)
        self.built = True

    def compile(self, gen_a_optimizer, gen_b_optimizer, disc_x_optimizer, disc_y_optimizer, disc_loss_fn=None,
                gen_loss_fn=None, cycle_loss_fn_a=None, cycle_loss_fn_b=None, identity_loss_fn_a=None,
                identity_loss_fn_b=None, label_smoothing_factor=0.0, **kwargs):
        """The loss callables of the reference are fixed functions here (LSGAN MSE with label smoothing,
        MAE cycle / identity); they are accepted for signature compatibility."""
        self.gen_a_optimizer, self.gen_b_optimizer = gen_a_optimizer, gen_b_optimizer
        self.disc_a_optimizer, self.disc_b_optimizer = disc_x_optimizer, disc_y_optimizer
        self.label_smoothing_factor = label_smoothing_factor

    @property
    def metrics_names(self):
        return list(METRIC_NAMES)

    def reset_metrics(self):
        self._sums[:] = 0.0
        self._count = 0

    def _slot(self, i):
        return self._scalars[i:i + 1]

    def train_step(self, batch_data):
        """One optimisation step of G_a, G_b, D_a, D_b (CycleGAN.py:615-710).  batch_data = (real_a, real_b):
        NHWC float32 arrays / tensors in [-1, 1].  Returns {metric: running mean}."""
        from .engine import convert
        real_a, real_b = (convert(self._to_act(t), self.act_dtype) for t in batch_data)
        ga, gb, da, db = self.gen_a, self.gen_b, self.disc_a, self.disc_b
        ls = self.label_smoothing_factor
        one = 1.0 - ls + ls / 2.0
        zero = ls / 2.0
        world = D.world_size()
        # weight-derived operands of the new weight version, before any concurrent chain starts: each network's ~100 small launches on
        # a stream of its own with one event per layer (engine.ParamArena.refresh_derived), so that the chains wait for the layer
        # they are about to run, not for all four networks (SS_REFRESH_STREAMS=0: in front of the step, on this stream)
        rs = (None,) * 4
        if self.refresh_side_streams and self.dual_stream and real_a.device.type == "cuda" and not D.ranks_share_device():
            from .engine import side_streams
            rs = side_streams(real_a.device, 10)[6:10]
        for net, side in zip((ga, gb, da, db), rs):
            net.arena.refresh_derived(side)

        if (self.dual_stream and self.use_identity_loss and self.batch_generator_passes and not self.use_binary_crossentropy_a
                and (self.dual_stream == "force" or not D.ranks_share_device())):
            # Two kernel chains share the GPU: the persistent Winograd GEMMs then take 224 of the 256 CUs -- a GEMM workgroup holds a
            # CU's whole LDS, so with all 256 occupied the other chain's kernels only start when the GEMM ends; with 32 CUs left its
            # bandwidth-bound kernels run beside it (measured: 197.8 -> 195.9 ms per step, same box).  SS_GEMM_CUS overrides.
            cus = self.dual_gemm_cus if not os.environ.get("SS_GEMM_CUS") else None
            if cus:
                # straight through the C ABI: this key changes no cached answer (L.config_set would bump the epoch all host-side caches key on)
                lib = L.load()
                prev = int(lib.ss_config_get(b"gemm_cus"))
                if prev:          # configured by the caller (L.config_set / L.config): theirs wins, and stays
                    return self._train_step_dual(real_a, real_b, one, zero, world)
                lib.ss_config_set(b"gemm_cus", cus)
                try:
                    return self._train_step_dual(real_a, real_b, one, zero, world)
                finally:
                    lib.ss_config_set(b"gemm_cus", prev)
            return self._train_step_dual(real_a, real_b, one, zero, world)

        # ---- generators -------------------------------------------------------------------------------
        tape = Tape()
        if self.use_identity_loss and self.batch_generator_passes:
            # the translation and the identity pass of a generator use the same weights on independent samples (InstanceNorm
            # is per sample): one pass over the concatenated batch -- larger GEMMs, one weight transform instead of two
            n = real_a.n
            fake_b, same_b = LY.batch_split(tape, ga(Act(cat_batch([real_a.t, real_b.t]), requires_grad=False), True, tape), [n, real_b.n])
            fake_a, same_a = LY.batch_split(tape, gb(Act(cat_batch([real_b.t, real_a.t]), requires_grad=False), True, tape), [real_b.n, n])
            cycled_a = gb(fake_b, True, tape)
            cycled_b = ga(fake_a, True, tape)
        else:
            fake_b = ga(real_a, True, tape)
            fake_a = gb(real_b, True, tape)
            cycled_a = gb(fake_b, True, tape)
            cycled_b = ga(fake_a, True, tape)
            if self.use_identity_loss:
                same_a = gb(real_a, True, tape)
                same_b = ga(real_b, True, tape)
        tape.param_grads = False            # discriminators only route gradients in this phase
        disc_fake_a = da(fake_a, True, tape)
        disc_fake_b = db(fake_b, True, tape)
        tape.param_grads = True
        # slots: 0 adv_a 1 adv_b 2 cyc_a 3 cyc_b 4 id_a 5 id_b | 6 d_real_a 7 d_fake_a 8 d_real_b 9 d_fake_b
        losses.mse_const(disc_fake_b, one, 1.0 * self.loss_scale, self._slot(0))
        losses.mse_const(disc_fake_a, one, 1.0 * self.loss_scale, self._slot(1))
        if self.use_binary_crossentropy_a:
            losses.weighted_bce(real_b, cycled_b, 1.0, self.lambda_cycle_a * self.loss_scale, self._bce3)
            L.check(L.load().ss_copy(self._bce3.data_ptr(), 1, self._slot(2).data_ptr(), 1, 1, 1, _stream()), "ss_copy")
        else:
            losses.mae(real_b, cycled_b, (self.lambda_cycle_a) * self.loss_scale, self._slot(2))
        losses.mae(real_a, cycled_a, (self.lambda_cycle_b) * self.loss_scale, self._slot(3))
        if self.use_identity_loss:
            losses.mae(real_b, same_b, (self.lambda_cycle_a * self.lambda_identity_a) * self.loss_scale, self._slot(4))
            losses.mae(real_a, same_a, (self.lambda_cycle_b * self.lambda_identity_b) * self.loss_scale, self._slot(5))
        ga.zero_grad()
        gb.zero_grad()
        D.begin_backward([ga, gb, da, db])
        tape.backward()                      # d(L_a + L_b)/d theta for both generators in one traversal
        D.all_reduce_grads([ga, gb])
        self.gen_a_optimizer.apply(ga, 1.0 / (world * self.loss_scale))
        self.gen_b_optimizer.apply(gb, 1.0 / (world * self.loss_scale))

        # ---- discriminators ---------------------------------------------------------------------------
        tape = Tape()
        pooled_a = self.image_pool_a.query(fake_a.t)          # detached copy of the generated batch
        pooled_b = self.image_pool_b.query(fake_b.t)
        if self.batch_generator_passes:                       # real + pooled-fake batch of a discriminator in one pass
            disc_real_a, disc_fake_a2 = LY.batch_split(
                tape, da(Act(cat_batch([real_a.t, pooled_a]), requires_grad=False), True, tape), [real_a.n, pooled_a.shape[0]])
            disc_real_b, disc_fake_b2 = LY.batch_split(
                tape, db(Act(cat_batch([real_b.t, pooled_b]), requires_grad=False), True, tape), [real_b.n, pooled_b.shape[0]])
        else:
            disc_real_a = da(real_a, True, tape)
            disc_fake_a2 = da(Act(pooled_a, requires_grad=False), True, tape)
            disc_real_b = db(real_b, True, tape)
            disc_fake_b2 = db(Act(pooled_b, requires_grad=False), True, tape)
        losses.mse_const(disc_real_a, one, 0.5 * self.loss_scale, self._slot(6))
        losses.mse_const(disc_fake_a2, zero, 0.5 * self.loss_scale, self._slot(7))
        losses.mse_const(disc_real_b, one, 0.5 * self.loss_scale, self._slot(8))
        losses.mse_const(disc_fake_b2, zero, 0.5 * self.loss_scale, self._slot(9))
        da.zero_grad()
        db.zero_grad()
        D.begin_backward([ga, gb, da, db])
        tape.backward()
        D.all_reduce_grads([da, db])
        self.disc_a_optimizer.apply(da, 1.0 / (world * self.loss_scale))
        self.disc_b_optimizer.apply(db, 1.0 / (world * self.loss_scale))

        return self._update_metrics()

    def _train_step_dual(self, real_a, real_b, one, zero, world):
        try:
            return self._train_step_dual_body(real_a, real_b, one, zero, world)
        finally:
            # whatever happened (a kernel error in one of the chains), leave the arenas in their single-chain state
            for net in (self.gen_a, self.gen_b):
                net.arena.merge_into = net.arena.merge_from = None

    def _train_step_dual_body(self, real_a, real_b, one, zero, world):
        """The same step as two CONCURRENT kernel chains on two HIP streams.  In the generator phase the A->B->A chain
        (G_a on [real_a; real_b], G_b on fake_b, D_b on fake_b and their losses) and the B->A->B chain share nothing but
        read-only weights and inputs; each has its own tape, scratch buffer and gradient buffer (both chains produce gradients
        of BOTH generators, so chain B accumulates into the alternate buffer and the two are added after the join).  The
        discriminator phase splits into the D_a and the D_b chain.  Same arithmetic per chain as the single-stream step; the
        only difference is one extra fp32 addition per generator gradient element (g_A + g_B instead of accumulating in place)."""
        from .engine import side_streams
        ga, gb, da, db = self.gen_a, self.gen_b, self.disc_a, self.disc_b
        cur = torch.cuda.current_stream()
        s1, s2, s3, s4, s5, s6 = side_streams(real_a.device, 6)
        if self.stream_indices is not None:          # SS_CG_STREAMS="a,b,da,db": which of engine.side_streams the four chains take
            pool = side_streams(real_a.device, max(self.stream_indices) + 1)
            s1, s2, s3, s4 = (pool[i] for i in self.stream_indices)
        n_a, n_b = real_a.n, real_b.n
        # SS_OVERLAP_D=0: discriminator chains only after the generator backward passes (two phases); default: they start as soon
        # as the generator FORWARD passes have produced the fakes and run beside the generator backward passes.  Nothing they
        # write is read there: the generator backward goes through the discriminators with weight gradients switched off, all
        # four optimizer steps are applied at the end of the step (every gradient is taken at the pre-update weights, as in the
        # reference's single GradientTape, CycleGAN.py:500-560).
        overlap_d = os.environ.get("SS_OVERLAP_D", "1") != "0"

        ga.zero_grad()
        gb.zero_grad()
        da.zero_grad()
        db.zero_grad()
        ga.arena.zero_grad_alt()
        gb.arena.zero_grad_alt()
        # Data parallel: a gradient bucket of a generator is final once BOTH chains have run their last op on it.  The chain that gets
        # there second adds the other chain's share of that bucket to the main buffer and launches its exchange behind itself
        # (engine.ParamArena.note_done / _fire) -- the up-sampling and trunk buckets travel while the stems are still in backward.
        for net in (ga, gb):
            net.arena.merge_into, net.arena.merge_from = net.arena.grads, net.arena._alt()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        tape_a, tape_b = Tape(), Tape()
        if self.wgrad_side_streams:          # weight gradients of each chain on a stream of their own (engine.Tape.wgrad_stream)
            tape_a.wgrad_stream, tape_b.wgrad_stream = s5, s6
        with torch.cuda.stream(s1):
            fake_b, same_b = LY.batch_split(tape_a, ga(Act(cat_batch([real_a.t, real_b.t]), requires_grad=False), True, tape_a), [n_a, n_b])
            cycled_a = gb(fake_b, True, tape_a)
            tape_a.param_grads = False
            disc_fake_b = db(fake_b, True, tape_a)
            tape_a.param_grads = True
            losses.mse_const(disc_fake_b, one, 1.0 * self.loss_scale, self._slot(0))
            losses.mae(real_a, cycled_a, (self.lambda_cycle_b) * self.loss_scale, self._slot(3))
            losses.mae(real_b, same_b, (self.lambda_cycle_a * self.lambda_identity_a) * self.loss_scale, self._slot(4))
            fwd_a_done = s1.record_event()
        with torch.cuda.stream(s2):
            fake_a, same_a = LY.batch_split(tape_b, gb(Act(cat_batch([real_b.t, real_a.t]), requires_grad=False), True, tape_b), [n_b, n_a])
            cycled_b = ga(fake_a, True, tape_b)
            tape_b.param_grads = False
            disc_fake_a = da(fake_a, True, tape_b)
            tape_b.param_grads = True
            losses.mse_const(disc_fake_a, one, 1.0 * self.loss_scale, self._slot(1))
            losses.mae(real_b, cycled_b, (self.lambda_cycle_a) * self.loss_scale, self._slot(2))
            losses.mae(real_a, same_a, (self.lambda_cycle_b * self.lambda_identity_b) * self.loss_scale, self._slot(5))
            fwd_b_done = s2.record_event()

        def generator_backward():
            with torch.cuda.stream(s1):
                tape_a.backward()
            ga.arena.swap_grads()
            gb.arena.swap_grads()
            try:
                with torch.cuda.stream(s2):
                    tape_b.backward()
            finally:
                ga.arena.swap_grads()
                gb.arena.swap_grads()

        def discriminator_chains():
            """Forward + backward of the two discriminator losses on s3 / s4 (entered with the fakes visible on ``cur``)."""
            pooled_a = self.image_pool_a.query(fake_a.t)
            pooled_b = self.image_pool_b.query(fake_b.t)
            s3.wait_stream(cur)
            s4.wait_stream(cur)
            tape_da, tape_db = Tape(), Tape()
            with torch.cuda.stream(s3):
                disc_real_a, disc_fake_a2 = LY.batch_split(
                    tape_da, da(Act(cat_batch([real_a.t, pooled_a]), requires_grad=False), True, tape_da), [n_a, pooled_a.shape[0]])
                losses.mse_const(disc_real_a, one, 0.5 * self.loss_scale, self._slot(6))
                losses.mse_const(disc_fake_a2, zero, 0.5 * self.loss_scale, self._slot(7))
            with torch.cuda.stream(s4):
                disc_real_b, disc_fake_b2 = LY.batch_split(
                    tape_db, db(Act(cat_batch([real_b.t, pooled_b]), requires_grad=False), True, tape_db), [n_b, pooled_b.shape[0]])
                losses.mse_const(disc_real_b, one, 0.5 * self.loss_scale, self._slot(8))
                losses.mse_const(disc_fake_b2, zero, 0.5 * self.loss_scale, self._slot(9))
            D.begin_backward([da, db])
            with torch.cuda.stream(s3):
                tape_da.backward()
            with torch.cuda.stream(s4):
                tape_db.backward()
            return pooled_a, pooled_b      # alive until the caller has joined s3 / s4

        keep = None
        D.begin_backward([ga, gb])
        if overlap_d:
            cur.wait_event(fwd_a_done)
            cur.wait_event(fwd_b_done)
            keep = discriminator_chains()         # issued first: they are short and start while the host still issues what follows
        generator_backward()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        ga.arena.merge_alt_grads()
        gb.arena.merge_alt_grads()
        # the generator gradient exchange (the bulk of the step's bytes) runs behind the discriminator phase, which touches neither
        # the generators' weights nor their gradients; the generator optimizer steps are applied after it
        for net in (ga, gb):
            net.arena.merge_into = net.arena.merge_from = None
        gen_works = D.begin_all_reduce_grads([ga, gb])          # what backward has not launched (buckets with unused variables)
        if os.environ.get("SS_DEFER_G_ALLREDUCE", "1") == "0":
            D.finish_all_reduce_grads(gen_works)
            gen_works = []
        if not overlap_d:
            keep = discriminator_chains()
        cur.wait_stream(s3)
        cur.wait_stream(s4)
        D.all_reduce_grads([da, db])
        self.disc_a_optimizer.apply(da, 1.0 / (world * self.loss_scale))
        self.disc_b_optimizer.apply(db, 1.0 / (world * self.loss_scale))
        D.finish_all_reduce_grads(gen_works)
        self.gen_a_optimizer.apply(ga, 1.0 / (world * self.loss_scale))
        self.gen_b_optimizer.apply(gb, 1.0 / (world * self.loss_scale))
        del keep
        del tape_a, tape_b
        return self._update_metrics()

    # ---- helpers --------------------------------------------------------------------------------------
    def _to_act(self, t):
        if isinstance(t, Act):
            return t
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
        from .engine import convert
        return convert(Act(t.to(self.device, dtype=torch.float32).contiguous(), requires_grad=False), self.act_dtype)

    def _update_metrics(self):
        if not self.sync_metrics:
            self._count += 1
            return {}
        # one device->host read per step; the values are THIS rank's (means over its shard of the batch): the cross-rank exchange is
        # not part of the step -- `global_metrics()` does it once per logging interval (the training loop: once per epoch)
        s = self._scalars.cpu().numpy().astype(np.float64)
        adv_a, adv_b = s[0], s[1]
        cyc_a, cyc_b = s[2] * self.lambda_cycle_a, s[3] * self.lambda_cycle_b
        id_a = s[4] * self.lambda_cycle_a * self.lambda_identity_a if self.use_identity_loss else 0.0
        id_b = s[5] * self.lambda_cycle_b * self.lambda_identity_b if self.use_identity_loss else 0.0
        vals = dict(d_a=(s[6] + s[7]) * 0.5, d_b=(s[8] + s[9]) * 0.5, d_fake_a=s[7], d_fake_b=s[9],
                    d_real_a=s[6], d_real_b=s[8], g_a=adv_a + cyc_a + id_a, g_b=adv_b + cyc_b + id_b,
                    g_adv_a=adv_a, g_adv_b=adv_b, g_cyc_a=cyc_a, g_cyc_b=cyc_b, g_id_a=id_a, g_id_b=id_b)
        self._count += 1
        self._sums += np.array([vals[k] for k in METRIC_NAMES])
        return {k: float(self._sums[i] / self._count) for i, k in enumerate(METRIC_NAMES)}

    def global_metrics(self):
        """The running means since reset_metrics() averaged over the ranks (equal shards: the mean of the ranks' means is the mean over
        the global batches) -- ONE small all-reduce per call; train_step itself exchanges nothing but gradients.  Single process: the
        running means as they are."""
        if self._count == 0:
            return {}
        m = D.mean_scalars(self._sums / self._count)
        return {k: float(m[i]) for i, k in enumerate(METRIC_NAMES)}

    @staticmethod
    def to_numpy_array(x):
        if isinstance(x, Act):
            x = x.dense()
        return x.detach().float().cpu().numpy().copy()          # 16-bit activation storage: numpy has no bfloat16

    # ---- persistence (Keras variable order; see DESIGN.md for the .keras/HDF5 status) ----------------------
    def get_weights(self):
        return {nm: net.get_weights() for nm, net in (("gen_a", self.gen_a), ("gen_b", self.gen_b),
                                                      ("disc_a", self.disc_a), ("disc_b", self.disc_b))}

    _NETS = ("gen_a", "gen_b", "disc_a", "disc_b")
    _OPTS = dict(gen_a="gen_a_optimizer", gen_b="gen_b_optimizer", disc_a="disc_a_optimizer", disc_b="disc_b_optimizer")

    def _config(self):
        return dict(filters=self.gen_a.filters, nd=self.gen_a.nd, nr=self.gen_a.nr, nu=self.gen_a.nu,
                    use_skip_connection=self.gen_a.use_skip_connection, use_resize_convolution=self.gen_a.use_resize_convolution,
                    sigmoid_output_a=self.gen_a.sigmoid_output,
                    disc_filters=self.disc_a.filters, disc_nd=self.disc_a.nd, gaussian_noise_value=self.disc_a.gaussian_noise_value,
                    lambda_cycle_a=self.lambda_cycle_a, lambda_cycle_b=self.lambda_cycle_b,
                    lambda_identity_a=self.lambda_identity_a, lambda_identity_b=self.lambda_identity_b,
                    optimizers={self._OPTS[nm]: dict(learning_rate=float(o.learning_rate), beta_1=float(o.beta_1), beta_2=float(o.beta_2))
                                for nm in self._NETS for o in [getattr(self, self._OPTS[nm])] if o is not None})

    def wait_saved(self, reraise=True):
        """Join a ``save(..., background=True)`` still being written.  reraise = True: re-raises what its writer raised; False: a
        failed background write (an intermediate checkpoint: disk full, converter hiccup) is reported as a warning and returned --
        it must not cost the caller the save it is about to make."""
        t = getattr(self, "_saver", None)
        if t is None:
            return None
        t.join()
        self._saver = None
        err, self._saver_error = getattr(self, "_saver_error", None), None
        if err is not None:
            if reraise:
                raise err
            import warnings
            warnings.warn(f"a background checkpoint could not be written ({err!r}); training state is unaffected", stacklevel=2)
        return err

    def save(self, path, background=False):
        """``model.save('…/model.keras')`` (CycleGAN.py:203-204,221): a Keras-3 archive (zip of config.json, metadata.json and
        model.weights.h5 with the four networks under gen_a/ gen_b/ disc_a/ disc_b/ and the four Adam states; keras_io.py).  A path
        ending in ``.npz`` writes the plain-numpy form instead (variable names of this framework).
        background = True (the per-epoch checkpoints of ``start_training``, CycleGAN.py:203-205): the weights and optimizer slots are
        copied to the host NOW (1.2 GB: what the file will hold is this moment's state), the archive is written by a thread while the
        next epoch trains; the next save / ``wait_saved`` joins it (a failure of THAT earlier write is warned about, not raised: this
        save still happens)."""
        self.wait_saved(reraise=False)
        if path.endswith(".npz"):
            arrays = {}
            for nm in self._NETS:
                net = getattr(self, nm)
                for name, w in zip(net.variable_names, net.get_weights()):
                    arrays[f"{nm}/{name}"] = w
            arrays["__config__"] = np.array(json.dumps(self._config()))
            np.savez(path, **arrays)
            return
        from . import keras_io as K
        arrays = {}
        for nm in self._NETS:
            arrays.update(K.net_arrays(getattr(self, nm), nm + "/"))
        for nm in self._NETS:
            arrays.update(K.optimizer_arrays(getattr(self, self._OPTS[nm]), getattr(self, nm), self._OPTS[nm] + "/"))
        if not background:
            K.write_archive(path, arrays, "CycleGanModel", self._config())
            return
        import threading
        cfg = self._config()

        def write():
            try:
                K.write_archive(path, arrays, "CycleGanModel", cfg)
            except BaseException as e:          # noqa: BLE001 -- surfaced by wait_saved() on the training thread
                self._saver_error = e

        self._saver = threading.Thread(target=write, name="checkpoint-writer")
        self._saver.start()

    @classmethod
    def load(cls, path, device=None):
        """Counterpart of ``keras.models.load_model(.../model.keras)`` (CycleGAN.py:228-231); restores the Adam states when the archive
        holds them (the reference never resumes training, CycleGAN.py has no ``initial_epoch``)."""
        device = device if device is not None else D.local_device()
        if path.endswith(".npz"):
            z = np.load(path)
            cfg = json.loads(str(z["__config__"]))
            arrays = None
        else:
            from . import keras_io as K
            _, cfg, arrays = K.read_archive(path)
        gk = dict(filters=cfg["filters"], num_downsampling_blocks=cfg["nd"], num_residual_blocks=cfg["nr"],
                  num_upsample_blocks=cfg["nu"], device=device, use_skip_connection=cfg.get("use_skip_connection", False),
                  use_resize_convolution=cfg.get("use_resize_convolution", False))
        dk = dict(filters=cfg["disc_filters"], num_downsampling_blocks=cfg["disc_nd"], device=device,
                  gaussian_noise_value=cfg.get("gaussian_noise_value", 0.0))
        nets = dict(gen_a=ResnetGenerator(sigmoid_output=cfg.get("sigmoid_output_a", False), **gk), gen_b=ResnetGenerator(**gk),
                    disc_a=PatchDiscriminator(**dk), disc_b=PatchDiscriminator(**dk))
        model = cls(nets["gen_a"], nets["gen_b"], nets["disc_a"], nets["disc_b"], lambda_cycle_a=cfg["lambda_cycle_a"],
                    lambda_cycle_b=cfg["lambda_cycle_b"], lambda_identity_a=cfg["lambda_identity_a"],
                    lambda_identity_b=cfg["lambda_identity_b"])
        model.use_binary_crossentropy_a = cfg.get("sigmoid_output_a", False)
        if arrays is None:
            for nm, net in nets.items():
                net.set_weights([z[f"{nm}/{name}"] for name in net.variable_names])
            return model
        legacy = K.NameCounters()
        for nm in cls._NETS:
            K.load_net_arrays(nets[nm], nm + "/", arrays, legacy)
        if any(k.startswith("gen_a_optimizer/") for k in arrays):
            # the optimizers as they were: learning rate / betas recorded by save() (an archive written part-way through the linear
            # decay resumes at ITS step size); archives without the record fall back to the workflow defaults (CycleGAN.py:168-171)
            oc = cfg.get("optimizers", {})
            mk = lambda nm: Adam(oc.get(nm, {}).get("learning_rate", 2e-4), beta_1=oc.get(nm, {}).get("beta_1", 0.5),
                                 beta_2=oc.get(nm, {}).get("beta_2", 0.999))
            model.compile(*(mk(cls._OPTS[nm]) for nm in cls._NETS))
            for nm in cls._NETS:
                K.load_optimizer_arrays(getattr(model, cls._OPTS[nm]), nets[nm], cls._OPTS[nm] + "/", arrays)
        return model


class DataLoader:
    """Unpaired A/B batches: floor length, contiguous slices, independent per-epoch shuffles (CycleGAN.py:454-479)."""

    def __init__(self, train_a, train_b, batch_size=1, use_dataloader=False, scale_for_binary_crossentropy=False,
                 invert_images=False, **kwargs):
        self.batch_size = batch_size
        self.train_a, self.train_b = train_a, train_b
        self.use_dataloader = use_dataloader
        self.scale_for_binary_crossentropy = scale_for_binary_crossentropy
        self.invert_images = invert_images

    def __len__(self):
        return int(min(len(self.train_a), len(self.train_b)) / float(self.batch_size))

    def __getitem__(self, idx):
        a = self.train_a[idx * self.batch_size:(idx + 1) * self.batch_size]
        b = self.train_b[idx * self.batch_size:(idx + 1) * self.batch_size]
        if self.use_dataloader:
            a = CycleGAN.load_images(a, False, invert=self.invert_images)
            b = CycleGAN.load_images(b, self.scale_for_binary_crossentropy)
        return np.asarray(a), np.asarray(b)

    def on_epoch_end(self):
        np.random.shuffle(self.train_a)
        np.random.shuffle(self.train_b)


class GANMonitor:
    """Per-epoch preview sheets, the role of the reference's ``GANMonitor`` callback (CycleGAN.py:810-905): for the first
    ``num_img`` test images of each domain one row  [input | translated | cycled back | input dimmed + outline of the thresholded
    translation (A-B-A) or of the input mask (B-A-B)]  written as ``A-B-A_Epoch_#####.tif`` / ``B-A-B_Epoch_#####.tif``.
    Every panel is min-max scaled to uint8, the outline is mask XOR its 2x eroded self, overlay brightness 0.7.
    Deviation: the reference normalises its test arrays IN PLACE while drawing, so from the first sheet on it feeds [0, 255]
    images to the generators; here the test images are left untouched."""

    def __init__(self, test_a, test_b, output_dir, num_img=2):
        self.test_a, self.test_b, self.output_dir, self.num_img = test_a, test_b, output_dir, num_img

    @staticmethod
    def _u8(img):
        img = np.asarray(img, np.float32)
        lo, hi = float(img.min()), float(img.max())
        return np.zeros(img.shape, np.uint8) if hi <= lo else ((img - lo) / (hi - lo) * 255.0).astype(np.uint8)

    def _sheet(self, first, gen_1, gen_2, outline_from_input):
        from scipy import ndimage
        n = min(self.num_img, len(first))
        if n == 0:
            return None
        h, w = first.shape[1], first.shape[2]
        sheet = np.zeros((n * h, 4 * w, 3), np.uint8)
        for i in range(n):
            x = np.ascontiguousarray(first[i:i + 1], dtype=np.float32)
            y_act = gen_1(torch.from_numpy(x).to(gen_1.arena.device), False)
            y = CycleGanModel.to_numpy_array(y_act)
            z = CycleGanModel.to_numpy_array(gen_2(y_act.dense(), False))
            panels = [self._u8(x[0, :, :, 0]), self._u8(y[0, :, :, 0]), self._u8(z[0, :, :, 0])]
            mask = (panels[0] if outline_from_input else panels[1]) > 127
            mask ^= ndimage.binary_erosion(mask, iterations=2)
            grey = panels[1] if outline_from_input else panels[0]
            for k, pnl in enumerate(panels):
                sheet[i * h:(i + 1) * h, k * w:(k + 1) * w, :] = pnl[:, :, None]
            dim = (grey * 0.7).astype(np.uint8)
            sheet[i * h:(i + 1) * h, 3 * w:, :] = dim[:, :, None]
            sheet[i * h:(i + 1) * h, 3 * w:, 0] = np.maximum(dim, mask.astype(np.uint8) * 255)
        return sheet

    def on_epoch_end(self, model, epoch):
        from PIL import Image
        os.makedirs(self.output_dir, exist_ok=True)
        for tag, first, g1, g2, from_input in (("A-B-A", self.test_a, model.gen_a, model.gen_b, False),
                                               ("B-A-B", self.test_b, model.gen_b, model.gen_a, True)):
            if first is None or len(first) == 0:
                continue
            sheet = self._sheet(first, g1, g2, from_input)
            if sheet is not None:
                Image.fromarray(sheet).save(os.path.join(self.output_dir, '{}_Epoch_{:05d}.tif'.format(tag, epoch + 1)))


class CycleGAN:
    """Workflow object with the reference's attribute names and defaults (CycleGAN.py:21-114)."""

    def __init__(self, root_dir='./', image_shape=(384, 384, 1), allow_memory_growth=True, use_gpus_no=(0,)):
        self.batch_size = 2
        self.epochs = 50
        self.learning_rate = 2e-4
        self.use_data_loader = False
        self.filters = 32
        self.num_downsampling_blocks_gen = 3
        self.num_residual_blocks_gen = 9
        self.num_upsampling_blocks_gen = 3
        self.num_downsampling_blocks_disc = 2
        self.allow_memory_growth = allow_memory_growth
        self.use_gpus_no = use_gpus_no
        # not in the reference: 'f32' | 'bf16' | 'f16' activation storage for training (fp32 master weights, fp32 statistics and
        # accumulation either way).  Saved models are fp32 and reload as fp32 networks.
        self.activation_storage = os.environ.get("SS_ACT_DTYPE", "f32")
        self.lambda_cycle_a = 10
        self.lambda_cycle_b = 10
        self.use_binary_crossentropy = False
        self.use_linear_decay = True
        self.decay_epoch = int(0.75 * self.epochs)
        # Keras' LearningRateScheduler acts on model.optimizer, which is none of the four Adams,
        # so the reference's linear decay never reaches them.  Default reproduces that (constant 2e-4).
        self.apply_lr_decay_to_adams = False
        self.lambda_identity_a = 0.5
        self.lambda_identity_b = 0.5
        self.use_skip_connection = True
        self.use_resize_convolution = False
        self.label_smoothing_factor = 0.0
        self.gaussian_noise_value = 0.15
        self.invert_images = False
        self.image_pool_size = 50
        self.gen_a = self.gen_b = self.disc_a = self.disc_b = None
        self.model = None
        self.data = None
        self.root_dir = root_dir
        self.model_dir = os.path.join(self.root_dir, '2_CycleGAN', 'Models')
        self.image_shape = image_shape
        self.prefix = time.strftime('%Y-%m-%d_%H-%M-%S', time.localtime())
        # pools are created HERE, with the constructor default batch_size (= 2) -- reference quirk, see module doc
        self.image_pool_a = ImagePool(batch_size=self.batch_size, pool_size=self.image_pool_size)
        self.image_pool_b = ImagePool(batch_size=self.batch_size, pool_size=self.image_pool_size)
        self.device = D.local_device()
        self.seed = 0
        from . import HelperFunctions
        data = os.path.join(self.root_dir, '2_CycleGAN', 'data')
        self.train_a = HelperFunctions.get_image_file_paths_from_directory(os.path.join(data, 'trainA'), missing_ok=True)
        self.test_a = HelperFunctions.get_image_file_paths_from_directory(os.path.join(data, 'testA'), missing_ok=True)
        self.train_b = HelperFunctions.get_image_file_paths_from_directory(os.path.join(data, 'trainB'), missing_ok=True)
        self.test_b = HelperFunctions.get_image_file_paths_from_directory(os.path.join(data, 'testB'), missing_ok=True)

    def create_model(self):
        assert not (self.use_binary_crossentropy and (self.lambda_identity_a > 0 or self.lambda_identity_b > 0)), \
            'binary crossentropy cannot be used with identity mapping (CycleGAN.py:71)'
        ch = self.image_shape[-1] if len(self.image_shape) == 3 else 1
        kw = dict(filters=self.filters, num_downsampling_blocks=self.num_downsampling_blocks_gen,
                  num_residual_blocks=self.num_residual_blocks_gen, num_upsample_blocks=self.num_upsampling_blocks_gen,
                  channels=ch, device=self.device, use_skip_connection=self.use_skip_connection,
                  use_resize_convolution=self.use_resize_convolution, act_dtype=self.activation_storage)
        self.gen_a = ResnetGenerator(seed=self.seed + 1, sigmoid_output=self.use_binary_crossentropy, **kw)
        self.gen_b = ResnetGenerator(seed=self.seed + 2, **kw)
        self.disc_a = PatchDiscriminator(filters=2 * self.filters, num_downsampling_blocks=self.num_downsampling_blocks_disc,
                                         channels=ch, padding="valid", device=self.device, seed=self.seed + 3,
                                         gaussian_noise_value=self.gaussian_noise_value, act_dtype=self.activation_storage)
        self.disc_b = PatchDiscriminator(filters=2 * self.filters, num_downsampling_blocks=self.num_downsampling_blocks_disc,
                                         channels=ch, padding="valid", device=self.device, seed=self.seed + 4,
                                         gaussian_noise_value=self.gaussian_noise_value, act_dtype=self.activation_storage)
        D.broadcast_params([self.gen_a, self.gen_b, self.disc_a, self.disc_b])
        D.enable_overlap([self.gen_a, self.gen_b, self.disc_a, self.disc_b])
        if D.world_size() > 1:
            # per-rank image buffers draw from per-rank streams (the reference's single process has one module-level stream)
            self.image_pool_a.rng = random.Random(1000003 * (self.seed + 1) + 2 * D.rank())
            self.image_pool_b.rng = random.Random(1000003 * (self.seed + 1) + 2 * D.rank() + 1)
        model = CycleGanModel(generator_a=self.gen_a, generator_b=self.gen_b, discriminator_a=self.disc_a,
                              discriminator_b=self.disc_b, image_pool_a=self.image_pool_a, image_pool_b=self.image_pool_b,
                              lambda_cycle_a=self.lambda_cycle_a, lambda_cycle_b=self.lambda_cycle_b,
                              lambda_identity_a=self.lambda_identity_a, lambda_identity_b=self.lambda_identity_b)
        model.compile(gen_a_optimizer=Adam(learning_rate=self.learning_rate, beta_1=0.5),
                      gen_b_optimizer=Adam(learning_rate=self.learning_rate, beta_1=0.5),
                      disc_x_optimizer=Adam(learning_rate=self.learning_rate, beta_1=0.5),
                      disc_y_optimizer=Adam(learning_rate=self.learning_rate, beta_1=0.5),
                      gen_loss_fn=self.generator_loss_fn, disc_loss_fn=self.discriminator_loss_fn,
                      label_smoothing_factor=self.label_smoothing_factor)
        model.use_binary_crossentropy_a = self.use_binary_crossentropy      # CycleGAN.py:117-121
        return model

    # LSGAN losses with label smoothing (CycleGAN.py:301-308).  train_step calls the same kernel with the same targets; these
    # methods give the reference's callable surface: NHWC device tensors (or Acts) in, python floats out.
    @staticmethod
    def _mse_to(t, target):
        a = t if isinstance(t, Act) else Act(t.to(dtype=torch.float32).contiguous(), requires_grad=False)
        slot = torch.zeros(1, dtype=torch.float32, device=a.device)
        losses.mse_const(a, target, 1.0, slot, want_grad=False)
        return float(slot.item())

    def generator_loss_fn(self, fake):
        ls = self.label_smoothing_factor
        return self._mse_to(fake, 1.0 - ls + ls / 2)

    def discriminator_loss_fn(self, real, fake):
        ls = self.label_smoothing_factor
        real_loss = self._mse_to(real, 1.0 - ls + ls / 2)
        fake_loss = self._mse_to(fake, ls / 2)
        return (real_loss + fake_loss) * 0.5, real_loss, fake_loss

    def linear_decay(self, epoch, current_lr=None):
        if epoch < self.decay_epoch:
            return self.learning_rate
        decay = (1 - ((epoch - self.decay_epoch) / float(self.epochs - self.decay_epoch))) ** 1
        return self.learning_rate * decay

    @staticmethod
    def load_images(image_list, scale_for_binary_crossentropy=False, invert=False):
        from . import HelperFunctions
        r = (0, 1) if scale_for_binary_crossentropy else (-1, 1)
        images = HelperFunctions.load_and_preprocess_images(input_dir_or_filelist=image_list, threshold_value=None,
                                                            normalization_range=r, output_channels=1,
                                                            contrast_optimization_range=None)
        if invert:
            images *= -1.0
        return images

    def run_inference(self, files, output_directory, source_domain, model=None, tile_images=False, min_overlap=2,
                      manage_overlap_mode=2, use_gpu=False):
        """CycleGAN.py:224-286: translate every image of ``files`` with generator A (source 'A') or B, whole-image or
        tiled+stitched, min-max to uint8, save under the input file name.  Always runs on the MI355X (``use_gpu`` is
        accepted for signature compatibility; there is no CPU path)."""
        from PIL import Image
        from . import HelperFunctions
        if self.model is None:
            if model is None:
                latest = sorted(os.listdir(self.model_dir))[-1]     # timestamp-named directories (CycleGAN.py:102,228)
                self.model = CycleGanModel.load(os.path.join(self.model_dir, latest, 'model.keras'), self.device)
            elif isinstance(model, str):
                self.model = CycleGanModel.load(model, self.device)
            else:
                self.model = model
        gen = self.model.gen_a if 'a' in source_domain.lower() else self.model.gen_b
        input_files = HelperFunctions.load_and_preprocess_images(files, normalization_range=(-1, 1))
        file_names = HelperFunctions.get_image_file_paths_from_directory(files) if isinstance(files, str) and os.path.isdir(files) \
            else ([files] if isinstance(files, str) else list(files))
        os.makedirs(output_directory, exist_ok=True)
        for i in range(input_files.shape[0]):
            input_file = input_files[i]
            if 'a' in source_domain.lower() and self.invert_images:
                input_file *= -1
            if tile_images:
                tiles = np.asarray(HelperFunctions.tile_image(input_file, self.image_shape[0], self.image_shape[1], min_overlap=min_overlap))
                pred = np.asarray([CycleGanModel.to_numpy_array(gen(torch.from_numpy(t[None]).to(self.device), training=False))[0] for t in tiles])
                img = HelperFunctions.stitch_image(pred, input_file.shape[1], input_file.shape[0], min_overlap=min_overlap,
                                                   manage_overlap_mode=manage_overlap_mode)
            else:
                # the nets are shape-agnostic: no rebuild + set_weights as in CycleGAN.py:243-251
                img = CycleGanModel.to_numpy_array(gen(torch.from_numpy(np.ascontiguousarray(input_file[None])).to(self.device), training=False))[0].copy()
            img = img[:, :, 0]
            if 'b' in source_domain.lower() and self.invert_images:
                img *= -1
            img -= np.min(img)
            img /= np.max(img)
            img *= 255
            img = img.astype(np.uint8)
            Image.fromarray(img).save(os.path.join(output_directory, os.path.split(file_names[i])[-1]))

    def start_training(self):
        """Epoch loop equivalent to ``model.fit(DataLoader, epochs, callbacks)`` (CycleGAN.py:182-222): per-epoch metric
        reset, CSV log (';'), per-epoch weight checkpoint, final ``model`` file.  Under torch.distributed each rank
        takes a contiguous slice of every global batch."""
        os.makedirs(os.path.join(self.model_dir, self.prefix), exist_ok=True)
        from . import keras_io
        keras_io.warn_if_no_hdf5('CycleGAN.start_training')
        self.decay_epoch = int(0.75 * self.epochs)
        if not self.use_data_loader:
            self.train_a = self.load_images(self.train_a, False, invert=self.invert_images)
            self.train_b = self.load_images(self.train_b, self.use_binary_crossentropy)
        self.data = DataLoader(self.train_a, self.train_b, batch_size=self.batch_size, use_dataloader=self.use_data_loader,
                               scale_for_binary_crossentropy=self.use_binary_crossentropy, invert_images=self.invert_images)
        self.model = self.create_model()
        plotter = None
        if D.rank() == 0 and len(self.test_a) and len(self.test_b):
            test_a = self.load_images(self.test_a[:2], False, invert=self.invert_images)
            test_b = self.load_images(self.test_b[:2], self.use_binary_crossentropy)
            plotter = GANMonitor(test_a, test_b, os.path.join(self.root_dir, '2_CycleGAN', 'images', self.prefix), num_img=2)
        log_path = os.path.join(self.model_dir, self.prefix, 'training_log.csv')
        rank, world = D.rank(), D.world_size()
        D.check_batch_divisible(self.batch_size, world, 'CycleGAN.batch_size')
        for epoch in range(self.epochs):
            if self.use_linear_decay and self.apply_lr_decay_to_adams:
                lr = self.linear_decay(epoch)
                for opt in (self.model.gen_a_optimizer, self.model.gen_b_optimizer, self.model.disc_a_optimizer,
                            self.model.disc_b_optimizer):
                    opt.learning_rate = lr
            self.model.reset_metrics()
            logs = {}
            order = list(range(len(self.data)))
            np.random.shuffle(order)     # Keras fit(shuffle=True) shuffles the batch order of a Sequence (K-list 10)
            # batches decoded ahead of the train steps when they come from disk (HelperFunctions.prefetch: same order, same contents)
            ahead = HelperFunctions.PREFETCH_DEPTH if self.use_data_loader else 0
            for a, b in HelperFunctions.prefetch(self.data.__getitem__, order, depth=ahead):
                per = len(a) // world
                logs = self.model.train_step((a[rank * per:(rank + 1) * per], b[rank * per:(rank + 1) * per]))
            if world > 1:
                logs = self.model.global_metrics()          # the epoch's one metrics exchange (the steps return rank-local values)
            self.data.on_epoch_end()
            if rank == 0:
                new = not os.path.exists(log_path)
                with open(log_path, 'a') as f:
                    if new:
                        f.write(';'.join(['epoch'] + sorted(logs)) + '\n')
                    f.write(';'.join([str(epoch)] + [repr(logs[k]) for k in sorted(logs)]) + '\n')
                self.model.save(os.path.join(self.model_dir, self.prefix, 'checkpoints_{:03d}.keras'.format(epoch + 1)), background=True)
                if plotter is not None:
                    plotter.on_epoch_end(self.model, epoch)
        if rank == 0:
            self.model.save(os.path.join(self.model_dir, self.prefix, 'model.keras'))
        return self.model
