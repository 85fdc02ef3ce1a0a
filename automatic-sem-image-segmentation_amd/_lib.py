"""ctypes binding of libsemseg_hip.so (include/semseg_hip.h).  Fails loudly when the library is absent:
there is no CPU or eager-PyTorch fallback for the hot path."""
import ctypes
import os

# Kernel arguments in device memory instead of host-coherent memory: the workloads here are chains of thousands of 7 - 25 us kernels, and
# each dispatch otherwise reads its arguments over the host link (measured on MI355X / ROCm 7.2, same box: MultiResUNet step 35.7 -> 34.0 ms,
# per-GPU batch-1 step 43.8 -> 42.8 / 42.5 -> 41.1 ms).  Read by the HIP runtime when it initialises, i.e. at the first device use; a value
# the caller has set wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_HERE, "libsemseg_hip.so")      # SS_LIB_PATH: diagnostic builds only

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
c_vp, c_sz = ctypes.c_void_p, ctypes.c_size_t


class _SizedDesc(ctypes.Structure):
    """First field = sizeof(struct) (checked by the library), second = ss_dtype of the activations; positional constructor
    arguments start at the third field."""

    def __init__(self, *args, dtype=0, **kw):
        super().__init__(ctypes.sizeof(type(self)), dtype, *args, **kw)


class WCacheEntry(ctypes.Structure):
    _fields_ = [("tag", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("bytes", ctypes.c_uint64)]


class WCache(ctypes.Structure):
    """ss_wcache: host-side directory of one layer's weight-derived operands; `base` is a caller-owned device buffer."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("count", c_i32), ("fills", c_i32), ("fill_only", c_i32), ("base", c_vp),
                ("bytes", ctypes.c_uint64), ("used", ctypes.c_uint64), ("entry", WCacheEntry * 32)]

    def __init__(self, base=None, nbytes=0):
        super().__init__(ctypes.sizeof(WCache), 0, 0, 0, base, nbytes, 0)


class ConvDesc(_SizedDesc):
    _fields_ = [("struct_size", ctypes.c_uint32), ("dtype", c_i32)] + [(n, c_i32) for n in ("n", "ih", "iw", "cin", "in_cstride", "oh", "ow", "cout", "out_cstride",
                                     "kh", "kw", "stride", "pad_top", "pad_left", "pad_mode", "transposed", "act")] + \
               [("act_alpha", c_f32), ("algo", c_i32),
                # optional x3h slots (include/semseg_hip.h): device uint32 with the bit pattern of max|x| / max|dy| + "already computed" flags
                ("x_amax", c_vp), ("dy_amax", c_vp), ("x_amax_valid", c_i32), ("dy_amax_valid", c_i32),
                # optional weight cache of the layer (ss_wcache: transformed / split weights kept across calls)
                ("w_cache", c_vp),
                # optional output statistics (sum y, sum y^2 per sample and channel) written by the forward epilogue
                ("y_stats", c_vp),
                # optional fused input normalisation: x is the pre-norm tensor, normalised in the operand load (in_norm_groups == 0: off)
                ("in_norm_mean", c_vp), ("in_norm_rstd", c_vp), ("in_norm_gamma", c_vp), ("in_norm_beta", c_vp),
                ("in_norm_groups", c_i32), ("in_norm_act", c_i32), ("in_norm_alpha", c_f32), ("in_norm_reserved", c_i32),
                # optional buffer that carries the transformed input operand from the forward to the weight-gradient pass
                ("saved_operand", c_vp)]


class NormDesc(_SizedDesc):
    _fields_ = [("struct_size", ctypes.c_uint32), ("dtype", c_i32)] + [(n, c_i32) for n in ("n", "h", "w", "c", "x_cstride", "y_cstride", "res_cstride", "groups")] + \
               [("eps", c_f32), ("act", c_i32), ("act_alpha", c_f32),
                # optional slots the norm raises to max|y| (forward) / max|dx| (backward) while writing the tensor (x3h scales)
                ("y_amax", c_vp), ("dx_amax", c_vp),
                # optional input statistics from the producing convolution's epilogue (ss_conv_desc.y_stats)
                ("x_stats", c_vp), ("x_stats_chunks", c_i32), ("reserved0", c_i32)]


class ProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("launches", c_i64), ("total_ms", ctypes.c_double), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


PAD_ZERO, PAD_REFLECT = 0, 1
POOL_PASS, POOL_FILL, POOL_SWAP, POOL_MAX_QUERY = 0, 1, 2, 16          # ss_pool_query modes
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2      # ss_dtype: storage type of activation tensors


def dtype_of(t):
    """ss_dtype of a torch tensor / torch dtype (activations: float32, bfloat16 or float16)."""
    import torch
    dt = t if isinstance(t, torch.dtype) else t.dtype
    try:
        return {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16}[dt]
    except KeyError:
        raise SemsegHipError(f"activation dtype {dt} is not one of float32 / bfloat16 / float16") from None


def torch_dtype(name):
    """'f32' | 'bf16' | 'f16' (or a torch dtype) -> torch dtype."""
    import torch
    if isinstance(name, torch.dtype):
        return name
    return {"f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
            "f16": torch.float16, "fp16": torch.float16, "float16": torch.float16}[str(name).lower()]

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PASS_FWD, PASS_BWD_DATA, PASS_BWD_WEIGHT = 0, 1, 2
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_BF16X3, ALGO_X6 = 0, 1, 2, 3, 4


def default_algo():
    """SS_PRECISION=bf16x3 opts in to split-bf16 matrix-core arithmetic for the Winograd GEMMs (default: exact fp32)."""
    return ALGO_BF16X3 if os.environ.get("SS_PRECISION", "f32").lower() == "bf16x3" else ALGO_AUTO

# name -> (restype, argtypes); the complete export list of include/semseg_hip.h
SIGNATURES = {
    "ss_version": (c_i32, []),
    "ss_status_string": (ctypes.c_char_p, [c_i32]),
    "ss_last_error": (ctypes.c_char_p, []),
    "ss_config_set": (c_i32, [ctypes.c_char_p, c_i64]),
    "ss_config_get": (c_i64, [ctypes.c_char_p]),
    "ss_config_key": (ctypes.c_char_p, [c_i32]),
    "ss_prof_enable": (c_i32, [c_i32]),
    "ss_prof_reset": (c_i32, []),
    "ss_prof_count": (c_i32, []),
    "ss_prof_get": (c_i32, [c_i32, ctypes.POINTER(ProfEntry)]),
    "ss_conv2d_workspace_bytes": (c_sz, [ctypes.POINTER(ConvDesc), c_i32]),
    "ss_conv2d_uses_amax": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "ss_conv2d_stats_chunks": (c_i32, [ctypes.POINTER(ConvDesc)]),
    "ss_conv2d_fuses_in_norm": (c_i32, [ctypes.POINTER(ConvDesc), c_i32]),
    "ss_norm_resident_timeouts": (c_i32, []),
    "ss_conv2d_saved_operand_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "ss_probe_mfma": (c_i32, [c_i32, c_vp, ctypes.c_size_t, c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ss_conv2d_wcache_bytes": (c_sz, [ctypes.POINTER(ConvDesc), c_i32]),
    "ss_wcache_invalidate": (None, [ctypes.POINTER(WCache)]),
    "ss_wprep_record_begin": (c_i32, []),
    "ss_wprep_record_end": (c_i32, [ctypes.POINTER(c_sz), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "ss_wprep_plan_write": (c_i32, [c_vp, c_sz]),
    "ss_wprep_run": (c_i32, [c_vp, c_vp, c_sz, c_vp]),
    "ss_wprep_run_part": (c_i32, [c_vp, c_vp, c_sz, c_i32, c_vp]),
    "ss_conv2d_fwd": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_conv2d_bwd_data": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "ss_conv2d_bwd_weight": (c_i32, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "ss_norm_reports_amax": (c_i32, [ctypes.POINTER(NormDesc)]),
    "ss_norm_workspace_bytes": (c_sz, [ctypes.POINTER(NormDesc)]),
    "ss_norm_fwd": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_sz, c_vp]),
    "ss_norm_apply": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ss_norm_infer": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "ss_norm_bwd": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "ss_norm_fwd_stats": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_norm_fwd_finish": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp]),
    "ss_norm_bwd_stats": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_norm_bwd_finish": (c_i32, [ctypes.POINTER(NormDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64,
                                   c_vp, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "ss_act_bwd": (c_i32, [c_i32, c_f32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_axpby": (c_i32, [c_f32, c_vp, c_i32, c_f32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_maxpool2x2_fwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ss_maxpool2x2_bwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ss_reflect_pad2d_fwd": (c_i32, [c_vp, c_i32, c_vp, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_reflect_pad2d_bwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_crop2d_fwd": (c_i32, [c_vp, c_i32, c_vp, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_crop2d_bwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_upsample2x_fwd": (c_i32, [c_vp, c_i32, c_vp, c_i32] + [c_i32] * 4 + [c_vp]),
    "ss_upsample2x_bwd": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 4 + [c_vp]),
    "ss_copy": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_act_bwd_t": (c_i32, [c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_axpby_t": (c_i32, [c_i32, c_f32, c_vp, c_i32, c_f32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_copy_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_pool_query": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_i32, c_vp]),
    "ss_mul_t": (c_i32, [c_i32, c_f32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_wgan_interpolate": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "ss_wgan_gp_grad": (c_i32, [c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp]),
    "ss_maxpool2x2_fwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ss_maxpool2x2_bwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "ss_reflect_pad2d_fwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_reflect_pad2d_bwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_crop2d_fwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_crop2d_bwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 8 + [c_vp]),
    "ss_upsample2x_fwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32] + [c_i32] * 4 + [c_vp]),
    "ss_upsample2x_bwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i32] + [c_i32] * 4 + [c_vp]),
    "ss_convert": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "ss_loss_mse_const_t": (c_i32, [c_i32, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_loss_mae_t": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_loss_weighted_bce_t": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_softmax_fwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_softmax_bwd_t": (c_i32, [c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp]),
    "ss_loss_weighted_bce_mc_t": (c_i32, [c_i32, c_vp, c_vp, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_fill": (c_i32, [c_vp, c_f32, c_i64, c_vp]),
    "ss_zero": (c_i32, [c_vp, c_sz, c_vp]),
    "ss_loss_workspace_bytes": (c_sz, [c_i64]),
    "ss_loss_mse_const": (c_i32, [c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_loss_mae": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_loss_weighted_bce": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "ss_adam_keras": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_f32, c_vp]),
    "ss_adam_keras_dev": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_f32, c_vp]),
}

_lib = None


class SemsegHipError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SemsegHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(make -C automatic-sem-image-segmentation_amd/csrc).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.ss_status_string(status).decode()
        detail = lib.ss_last_error().decode()
        raise SemsegHipError(f"{what} failed: {msg} ({status})" + (f": {detail}" if detail else ""))


CONFIG_EPOCH = 0      # bumped by config_set: host-side caches of pure functions of (descriptor, configuration) key on it


def config_set(key, value):
    """ss_config_set: explicit kernel-selection switches (include/semseg_hip.h)."""
    global CONFIG_EPOCH
    check(load().ss_config_set(key.encode(), int(value)), f"ss_config_set[{key}]")
    CONFIG_EPOCH += 1


def config_get(key):
    return int(load().ss_config_get(key.encode()))


class config:
    """``with config(x3h=0): ...`` -- temporarily override switches (tests / bench arithmetic modes)."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = config_get(k)
            config_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            config_set(k, v)
        return False


def prof_summary():
    """{kernel class: {launches, total_ms, avg_ms, flops, bytes}} of the instrumented launches since the last ss_prof_reset."""
    lib = load()
    out = {}
    for i in range(lib.ss_prof_count()):
        e = ProfEntry()
        check(lib.ss_prof_get(i, ctypes.byref(e)), "ss_prof_get")
        out[e.name.decode()] = dict(launches=int(e.launches), total_ms=float(e.total_ms), avg_ms=float(e.total_ms) / max(int(e.launches), 1),
                                    flops=float(e.flops), bytes=float(e.bytes))
    return out
