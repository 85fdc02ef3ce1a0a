/*
 * semseg_hip.h -- C ABI of libsemseg_hip.so: the MI355X (gfx950) kernels behind the
 * CycleGAN -> MultiResUNet training hot path of BAMresearch/automatic-sem-image-segmentation.
 *
 * The reference has no FFI: its "operator API" for this path is the set of Keras layers /
 * losses / optimizer calls listed below (Releases/Version 1.2.0/...), executed by the Keras
 * torch backend.  Each entry point names the reference call sites it replaces.
 *
 * Conventions
 *  - plain C, no C++/torch types; all device buffers are CALLER-OWNED, 16-byte aligned.  ACTIVATION tensors (x, y, dy, dx,
 *    residual: the `void*` arguments) are stored as the descriptor's / the `_t` entry point's ss_dtype -- fp32 (the reference's
 *    precision and the default), or bf16 / fp16 "mixed precision" storage (BASELINE configs 2 and 5); weights, their gradients,
 *    optimizer state, normalisation statistics, losses and every accumulation are always fp32.
 *  - activations are NHWC "views": pointer + (n,h,w,c) + cstride, where cstride >= c is the
 *    distance in elements between consecutive pixels (lets a layer write straight into a slice
 *    of a concatenated tensor: keras.layers.concatenate, UNet_Segmentation.py:469,542-551).
 *  - weights use the Keras variable layouts: Conv2D kernel (kh,kw,cin,cout), bias (cout);
 *    Conv2DTranspose kernel (kh,kw,cout,cin); norm gamma/beta (c).
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*) and allocates nothing;
 *    scratch comes from the caller (`ws`, size from the matching *_workspace_bytes()).
 *  - process-wide state the library DOES keep (a deviation from "no global state", on purpose): the
 *    kernel-selection table behind ss_config_set / ss_config_get (one table per process, NOT
 *    synchronised -- set it from one thread while no call is in flight) and the opt-in profiling
 *    recorder ss_prof_* (mutex-protected).  ss_last_error() is thread-local.  Nothing else persists
 *    between calls: derived weight operands live in CALLER-owned ss_wcache buffers.
 *  - the shared object exports exactly the entry points declared here (-fvisibility=hidden; the
 *    declarations below are inside a `visibility push(default)` region).
 *  - return value: SS_OK or a negative ss_status; ss_status_string() names it.
 */
#ifndef SEMSEG_HIP_H
#define SEMSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef enum ss_status {
    SS_OK = 0,
    SS_ERR_INVALID = -1,      /* bad descriptor / null pointer */
    SS_ERR_WORKSPACE = -2,    /* ws_bytes too small */
    SS_ERR_LAUNCH = -3,       /* HIP launch error (hipGetLastError) */
    SS_ERR_UNSUPPORTED = -4   /* combination not implemented */
} ss_status;

/* storage type of ACTIVATION tensors (x, y, dy, dx, residual): weights, their gradients, optimizer state, normalisation
 * statistics and every accumulation stay fp32 ("master" precision).  SURVEY 8(b) B2 dtype enum. */
typedef enum ss_dtype { SS_DTYPE_F32 = 0, SS_DTYPE_BF16 = 1, SS_DTYPE_F16 = 2 } ss_dtype;

enum { SS_PAD_ZERO = 0, SS_PAD_REFLECT = 1 };
enum { SS_ACT_NONE = 0, SS_ACT_RELU = 1, SS_ACT_LRELU = 2, SS_ACT_TANH = 3, SS_ACT_SIGMOID = 4 };
enum { SS_PASS_FWD = 0, SS_PASS_BWD_DATA = 1, SS_PASS_BWD_WEIGHT = 2 };
enum { SS_ALGO_AUTO = 0, SS_ALGO_DIRECT = 1, SS_ALGO_MFMA = 2,
       /* AUTO + opt-in "split-bf16" matrix-core arithmetic for the Winograd GEMMs: every fp32 operand is carried as
        * hi + lo bf16 planes and each product as hi*hi + hi*lo + lo*hi with fp32 accumulation (~2^-16 relative per
        * product; the default paths are exact fp32) */
       SS_ALGO_BF16X3 = 3,
       /* fp32-exact contraction on the bf16 matrix cores ("x6"): every fp32 operand is split EXACTLY into three bf16
        * pieces (h + m + l) and each product is formed as the six leading piece products with fp32 accumulation
        * (dropped terms <= 2^-26 relative, below one fp32 rounding).  AUTO uses it wherever the shape is eligible
        * (reduction channels % 32 == 0) unless the environment says SS_X6=0; SS_ALGO_MFMA never uses it
        * (v_mfma_f32_32x32x2_f32 only); SS_ALGO_X6 forces it on eligible shapes. */
       SS_ALGO_X6 = 4 };

int ss_version(void);
const char* ss_status_string(int status);
/* Human-readable detail behind the last non-OK status returned on the calling thread ("" if none): which check failed, the HIP
 * error string of a failed launch.  Thread-local; valid until the thread's next failing call. */
const char* ss_last_error(void);

/* Kernel-selection configuration: ONE explicit process-wide table instead of environment variables latched inside the library.
 * Keys (ss_config_key(i) enumerates them, NULL past the end): "x6" (16-bit matrix-core contraction with fp32-grade operand
 * splits; 0 = v_mfma_f32_32x32x2_f32 only), "x3h" (1 = two fp16 pieces / three products, 0 = exact three bf16 pieces / six
 * products), "x3h_direct", "x6p" (0 / 1 / 2 = off / auto / always), "winograd", "wino_r" (4 | 2), "wgrad_c1", "norm_fused_pix",
 * "tile_conv", measurement keys "gconv_fast", "gconv_nt512", "gconv_tile256".  Initial values: the matching SS_* environment
 * variables at load time, else the defaults.  Not synchronised: call while no other call is in flight.
 * ss_config_get returns INT64_MIN for an unknown key. */
int ss_config_set(const char* key, int64_t value);
int64_t ss_config_get(const char* key);
const char* ss_config_key(int index);

/* In-process kernel timing for bench.py's roofline leg: while enabled, the instrumented launchers (the contraction kernels:
 * gemm_x6p, gconv_x6, wgrad_x6, tile_conv, ...) bracket each kernel with HIP events recorded on the launch stream and count its
 * EXECUTED matrix-instruction FLOPs (every piece product of the x3h / x6 splits) and algorithmic bytes.  ss_prof_get synchronises
 * on the recorded events.  Profile on ONE stream: with two concurrent chains an interval also contains the neighbour's kernels. */
typedef struct ss_prof_entry { char name[64]; int64_t launches; double total_ms; double flops; double bytes; } ss_prof_entry;
int ss_prof_enable(int on);
int ss_prof_reset(void);
int ss_prof_count(void);
int ss_prof_get(int index, ss_prof_entry* out);

/* Measurement aid (bench.py `roofline.real_data_ceiling`; no product path calls it): TF/s and effective shader clock (MHz) of a
 * register-only stream of v_mfma_f32_32x32x16_f16 on every CU of device 0, on zero operands (random_operands == 0) or on random fp16
 * operands.  The chip clocks to its power budget: the second figure is the ceiling of any real-data kernel on THIS box
 * (profiles/r04_microbenchmarks.md: 0.61 - 0.66 of the nominal 2516.6 TF/s).  scratch: caller-owned device buffer of >= 65600
 * bytes.  Synchronises on `stream`. */
int ss_probe_mfma(int random_operands, void* scratch, size_t scratch_bytes, void* stream, double* tflops, double* mhz);

/* ------------------------------------------------------------------------------------------
 * 2-D convolution / transposed convolution.
 * Replaces keras.layers.Conv2D at CycleGAN.py:327,333,340,372,393,429/431,448 and
 * UNet_Segmentation.py:421; keras.layers.Conv2DTranspose at CycleGAN.py:353 and
 * UNet_Segmentation.py:542-551; and the ReflectionPadding2D that precedes the 'valid' convs
 * (CycleGAN.py:326,332,368,392) which is folded into the gather (never materialised in fwd).
 *
 * transposed == 0 : y[n,oy,ox,co] = bias[co] + sum x_pad[n, oy*stride+kh, ox*stride+kw, ci] * w[kh,kw,ci,co]
 *                   x_pad = x padded by pad_top/pad_left (and whatever is needed after) with zeros
 *                   (Keras 'same'/'valid') or by reflection (pad_mode == SS_PAD_REFLECT, stride 1 only).
 * transposed == 1 : Keras Conv2DTranspose(padding='same') with the torch-backend alignment:
 *                   y[n, iy*stride - pad_top + kh, ix*stride - pad_left + kw, co] += x[n,iy,ix,ci] * w[kh,kw,co,ci]
 *                   (pad_top = pad_left = torch `padding`; k=3,s=2 -> 1, output 2x; k=2,s=2 -> 0).
 * (ih,iw,cin) always describe x and (oh,ow,cout) always describe y of the FORWARD op.
 * `act` is applied to y in the forward epilogue (after bias); backward entry points take the
 * gradient w.r.t. the pre-activation output (use ss_act_bwd first).
 * ---------------------------------------------------------------------------------------- */
/* Weight cache of ONE convolution layer (optional).  The forward and data-gradient passes derive operands from the weights alone
 * (Winograd-transformed / transposed / 16-bit split planes, the weight's maximum); CycleGAN.py:612-633 runs every generator three
 * times forward and backward per optimizer step, so a layer's weights are otherwise transformed 4-6 times per step.  The caller
 * owns a HOST struct per layer plus a device buffer (`base`, `bytes` >= the sum of ss_conv2d_wcache_bytes over the passes it will
 * run); the library keeps the directory: each derived operand is a tagged entry, computed on first use (`fills` is incremented
 * -- a caller that shares the cache between streams orders them on it) and read afterwards, by ANY descriptor geometry of the layer
 * that needs the same operand.  An operand that does not fit is recomputed per call into the workspace, as without a cache.
 * The caller empties it with ss_wcache_invalidate whenever the weight
 * VALUES or the ss_config arithmetic switches change.  Results are bit-identical with and without a cache. */
typedef struct ss_wcache_entry { uint64_t tag, offset, bytes; } ss_wcache_entry;
typedef struct ss_wcache {
    uint32_t struct_size;            /* = sizeof(ss_wcache) */
    int32_t count;                   /* entries in use; 0 = empty */
    int32_t fills;                   /* incremented whenever an entry is computed */
    int32_t fill_only;               /* != 0: passes called with this cache only compute the entries they would keep (activation
                                      * pointers may be NULL, nothing else is launched): lets a caller refresh a layer's operands on
                                      * one stream right after the optimizer step, before concurrent streams use them */
    void* base;                      /* device buffer */
    uint64_t bytes, used;
    ss_wcache_entry entry[32];
} ss_wcache;
void ss_wcache_invalidate(ss_wcache* wc);

/* Batched refresh of the weight caches of MANY layers (optional).  Refreshing the caches layer by layer (fill_only calls) is ~100 small
 * launches per network and optimizer step -- weight maxima, tap-wise transposes, split planes, Winograd-transformed planes -- the same
 * kernels on the same pointers every step.  A caller records them once and replays the recording as one launch per kind of operand:
 *   ss_wprep_record_begin();                       the calling thread's fill_only calls now RECORD what they would launch
 *   ... the fill_only calls of every layer ...     (operands the recorder does not know are launched at once and make it `incomplete`)
 *   ss_wprep_record_end(&bytes, &n_jobs, &complete);
 *   ss_wprep_plan_write(host_buf, bytes);          serialise the plan; the caller copies the bytes to device memory verbatim
 *   ss_wprep_run(host_buf, dev_buf, bytes, stream) executes it: now (the recorded fills have not run yet) and -- if `complete` -- in
 *                                                  place of the per-layer refresh of every later step, as long as the weight arena, the
 *                                                  cache buffers, the descriptors that use them and the ss_config switches stay the same.
 * The caches' directories are left as the recording built them (do not invalidate a cache that a plan refreshes).  Results are
 * bit-identical to the per-layer refresh: the same kernel bodies run on the same operands.  Recording state is per thread. */
int ss_wprep_record_begin(void);
int ss_wprep_record_end(size_t* plan_bytes, int32_t* n_jobs, int32_t* complete);
int ss_wprep_plan_write(void* plan_host, size_t bytes);
int ss_wprep_run(const void* plan_host, const void* plan_dev, size_t bytes, void* stream);
/* The same in two parts, for a caller that lets the first layers start before the whole plan has run: part 0 = maxima (of every layer),
 * transposes and split planes (what the strided / transposed / 4x4 layers need), part 1 = the Winograd-transformed planes (the trunk).
 * Part 1 reads the maxima part 0 leaves: run part 0 first, on the same stream.  ss_wprep_run = part 0 followed by part 1. */
int ss_wprep_run_part(const void* plan_host, const void* plan_dev, size_t bytes, int part, void* stream);

/* An "amax slot" is SS_AMAX_SLOT_BYTES of device memory: 16 uint32 words, one per 256-byte line (word index 64 * i); the value it
 * holds is the MAXIMUM of those words.  Producers (ss_norm_fwd / ss_norm_bwd, the scan inside the convolution passes) raise the
 * words with one atomic per workgroup, spread over the lines -- thousands of atomics on a single address serialise in L2.
 * A slot handed to a producer must be zero (or hold a lower bound). */
#define SS_AMAX_SLOT_BYTES 4096

typedef struct ss_conv_desc {
    uint32_t struct_size;            /* = sizeof(ss_conv_desc): a caller built against another layout gets SS_ERR_INVALID, not a wild read */
    int32_t dtype;                   /* ss_dtype of x / y / dy / dx */
    int32_t n, ih, iw, cin, in_cstride;
    int32_t oh, ow, cout, out_cstride;
    int32_t kh, kw, stride;
    int32_t pad_top, pad_left;
    int32_t pad_mode;
    int32_t transposed;
    int32_t act;
    float act_alpha;
    int32_t algo;
    /* Optional (may be NULL / 0): caller-owned device SLOTS (SS_AMAX_SLOT_BYTES each, see below) for the bit pattern of max|x| and max|dy| -- the power-of-two
     * scales of the fp16 two-piece ("x3h") contraction.  A pass that needs a maximum (ss_conv2d_uses_amax) computes it INTO the
     * slot unless the matching *_valid flag says the slot already holds it, so that a tensor consumed by several passes (x:
     * forward + weight gradient, dy: data + weight gradient) is scanned once.  Without slots every pass scans for itself.
     * *_valid: 0 = contents unknown (the pass clears the slot, then scans into it), 1 = the slot holds the maximum (no scan),
     * 2 = the slot is ZERO on entry (caller vouches, e.g. fresh from a zeroed pool): the pass scans into it without clearing. */
    void* x_amax;
    void* dy_amax;
    int32_t x_amax_valid;
    int32_t dy_amax_valid;
    /* Optional (may be NULL): weight cache of the layer this descriptor belongs to, see ss_wcache below. */
    struct ss_wcache* w_cache;
    /* Optional (may be NULL), forward pass only: caller-owned fp32 buffer [n][ss_conv2d_stats_chunks(d)][cout][2] that receives
     * partial sums (sum y, sum y^2) of the OUTPUT per sample and channel, taken in the epilogue that writes y -- the statistics
     * pass of a following InstanceNorm / BatchNorm (ss_norm_desc::x_stats) then has nothing left to read ("fused IN + conv":
     * CycleGAN.py:327-329, 333-335).  Only written when ss_conv2d_stats_chunks(d) > 0. */
    void* y_stats;
    /* Optional (in_norm_groups == 0: off), forward and weight-gradient passes, every storage type: the second half of "fused
     * InstanceNorm + conv" (CycleGAN.py:327-333: Conv2D -> GroupNormalization -> relu -> pad -> Conv2D).  `x` is then the
     * PRE-normalisation tensor of a norm whose apply pass was skipped (ss_norm_fwd with y == NULL left only mean / rstd), and the
     * pass forms  act((x - mean[g,c]) * (rstd[g,c] * gamma[c]) + beta[c])  -- the arithmetic of ss_norm_fwd, bit for bit -- while
     * it loads its operand (16-bit storage: rounded to the stored type there, as the norm's own store would have); the normalised
     * tensor never exists in memory.  g = sample index when in_norm_groups == n, 0 when
     * in_norm_groups == 1; gamma may be NULL.  Only the passes for which ss_conv2d_fuses_in_norm(d, pass) != 0 take it (others
     * return SS_ERR_UNSUPPORTED; the caller then runs ss_norm_apply and a plain convolution).  x_amax, when given, refers to
     * the NORMALISED tensor: the forward pass leaves its maximum there (unless x_amax_valid), the weight gradient reads it. */
    const float* in_norm_mean;
    const float* in_norm_rstd;
    const float* in_norm_gamma;
    const float* in_norm_beta;
    int32_t in_norm_groups;
    int32_t in_norm_act;
    float in_norm_alpha;
    int32_t in_norm_reserved;
    /* Optional (may be NULL): caller-owned device buffer of ss_conv2d_saved_operand_bytes(d) bytes that carries the TRANSFORMED input
     * operand from the forward pass to the weight-gradient pass of the same call (same x, same descriptor): the forward pass writes
     * its Winograd-domain input planes (fp16 h / l pieces under per-tile scales) there instead of into its workspace, the
     * weight gradient contracts them with the transformed dy and does not transform x a second time (the reference keeps every
     * layer input alive for autograd, CycleGAN.py:615-690; 288 GB of HBM hold the transformed form as well).  A weight-gradient
     * call that is handed the buffer trusts its contents.  Ignored when ss_conv2d_saved_operand_bytes(d) == 0. */
    void* saved_operand;
} ss_conv_desc;

/* Bytes of ss_conv_desc::saved_operand this descriptor's forward pass fills and its weight-gradient pass reads (0: the path keeps
 * nothing -- then the field is ignored).  Today: fp32 storage, the Winograd x3h forward (F(4x4,3x3) on pre-split planes) together
 * with the pre-split-plane weight gradient.  Pure function of d (ignoring the pointer fields) and the ss_config table. */
size_t ss_conv2d_saved_operand_bytes(const ss_conv_desc* d);

/* != 0: `pass` (SS_PASS_FWD or SS_PASS_BWD_WEIGHT) of this descriptor applies ss_conv_desc::in_norm_* in its operand load (the
 * Winograd x3h forward transform and the Winograd weight gradient on pre-split planes, today).  Pure function of d (ignoring the
 * in_norm_* fields themselves) and the ss_config table. */
int ss_conv2d_fuses_in_norm(const ss_conv_desc* d, int pass);

/* Chunks per sample of the output statistics the FORWARD pass of `d` can emit (0: this descriptor's path cannot).  Today: fp32
 * storage, no fused activation, and one of -- the Winograd forward; the 1 -> C matrix-core kernel of the full-resolution stem
 * (one chunk per 8 x 64 output tile); the gather kernel gconv_x6v2 of the stride-2 / 4x4 layers (one chunk per 256 output pixels,
 * when those tiles do not straddle samples).  Pure function of d and the ss_config table; a forward call that was handed y_stats but
 * cannot take that kernel (workspace too small, misaligned operands) fails with SS_ERR_UNSUPPORTED instead of leaving it unwritten. */
int ss_conv2d_stats_chunks(const ss_conv_desc* d);
/* Upper bound of the device bytes the pass keeps in the layer's weight cache (0: nothing, e.g. the weight gradient). */
size_t ss_conv2d_wcache_bytes(const ss_conv_desc* d, int pass);

/* bit 0: this pass reads max|x| (and leaves it in d->x_amax when that is set), bit 1: likewise max|dy|.  Pure function of d. */
int ss_conv2d_uses_amax(const ss_conv_desc* d, int pass);

size_t ss_conv2d_workspace_bytes(const ss_conv_desc* d, int pass);
int ss_conv2d_fwd(const ss_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                  void* ws, size_t ws_bytes, void* stream);
/* dx (view with in_cstride) = d loss / d x ; overwritten (accumulate == 0) or added to (accumulate != 0) */
int ss_conv2d_bwd_data(const ss_conv_desc* d, const void* dy, const float* w, void* dx, int accumulate,
                       void* ws, size_t ws_bytes, void* stream);
/* dw (Keras layout, dense) and optional dbias; accumulate != 0 adds to the existing contents
 * (a net that is run several times per step, CycleGAN.py:621-633). */
int ss_conv2d_bwd_weight(const ss_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                         int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation.  One descriptor serves
 *   InstanceNorm = keras.layers.GroupNormalization(groups=-1, axis=3, epsilon=1e-5)
 *                  (CycleGAN.py:329,335,342,355,374): groups = n, per-sample statistics;
 *   BatchNorm    = keras.layers.BatchNormalization(axis=3[, scale=False]), momentum 0.99,
 *                  epsilon 1e-3 (UNet_Segmentation.py:422,448,470,473,494,502): groups = 1.
 * Forward:  z = (x - mean) * rstd * gamma + beta ;  y = act(z + residual)
 *           (biased variance, var = E[x^2] - E[x]^2 as keras.ops.moments on torch).
 * The fused activation / residual replace keras.layers.Activation / add / LeakyReLU that follow
 * the norm (CycleGAN.py:330,336,344,357,375; UNet_Segmentation.py:425,471-472,492-493).
 * ---------------------------------------------------------------------------------------- */
typedef struct ss_norm_desc {
    uint32_t struct_size;    /* = sizeof(ss_norm_desc) */
    int32_t dtype;           /* ss_dtype of x / y / residual / dy / dx / dres */
    int32_t n, h, w, c;
    int32_t x_cstride, y_cstride, res_cstride;
    int32_t groups;          /* n -> instance norm, 1 -> batch norm */
    float eps;
    int32_t act;
    float act_alpha;
    /* Optional (may be NULL), fp32 activations only: caller-owned amax slots (SS_AMAX_SLOT_BYTES), ZERO (or a lower bound) on entry, that
     * ss_norm_fwd / ss_norm_bwd raise to the bit pattern of max|y| / max|dx| while they write the tensor -- the power-of-two scale
     * the next convolution's x3h contraction needs (ss_conv_desc::x_amax / dy_amax with the *_valid flag set), without another
     * pass over the tensor.  dx_amax describes dx AFTER an accumulate_dx. */
    void* y_amax;
    void* dx_amax;
    /* Optional (may be NULL / 0), ss_norm_fwd only: partial sums (sum x, sum x^2) of the input as a convolution's epilogue wrote
     * them (ss_conv_desc::y_stats), [n][x_stats_chunks][c][2]; the forward then skips its own statistics pass (the one-launch
     * kernels for small groups ignore it). */
    const void* x_stats;
    int32_t x_stats_chunks;
    int32_t reserved0;
} ss_norm_desc;

size_t ss_norm_workspace_bytes(const ss_norm_desc* d);
/* Diagnostics of the one-pass InstanceNorm backward (ss_config norm_bwd_resident): the number of workgroups that gave up waiting at
 * their group-local barrier since the library was loaded -- 0 in every healthy run (a launch that counts here produced wrong
 * gradients; the bound on the wait exists so that a scheduling surprise can never hang the GPU).  Synchronises the device; -1 on error. */
int ss_norm_resident_timeouts(void);

/* 1: ss_norm_fwd / ss_norm_bwd of this descriptor raise y_amax / dx_amax (the two-pass kernels, any storage type: the
 * maximum of the STORED values; the one-launch kernels for small groups leave the slots untouched).  Pure function of d and the ss_config table. */
int ss_norm_reports_amax(const ss_norm_desc* d);
/* gamma may be NULL (scale=False); residual may be NULL.  mean/rstd: [groups*c] outputs kept for backward.
 * If moving_mean/moving_var are non-NULL (batch norm training) they are updated in place:
 * moving = moving*momentum + batch*(1-momentum), with the biased batch variance. */
/* y == NULL: statistics only (mean / rstd, moving update); the apply pass is left to the consumer (ss_conv_desc::in_norm_*) or to
 * a later ss_norm_apply. */
int ss_norm_fwd(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta,
                const void* residual, void* y, float* mean, float* rstd,
                float* moving_mean, float* moving_var, float momentum,
                void* ws, size_t ws_bytes, void* stream);
/* The apply pass of ss_norm_fwd alone, from given mean / rstd: y = act((x - mean) * rstd * gamma + beta + residual).  Raises
 * d->y_amax like ss_norm_fwd. */
int ss_norm_apply(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                  const float* mean, const float* rstd, void* stream);
/* inference-mode batch norm: statistics come from moving_mean / moving_var */
int ss_norm_infer(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta,
                  const float* moving_mean, const float* moving_var, const void* residual, void* y,
                  void* stream);
/* dy: gradient w.r.t. y.  y: forward output (needed when act != NONE).
 * dx (view with dx_cstride) receives d loss / d x.  dres (optional, view with d->res_cstride) receives the
 * gradient w.r.t. the residual input (= dy * act'(.)).  dgamma may be NULL.  Each accumulate_* flag != 0
 * adds into the destination instead of overwriting it.  `y` (the forward output, needed for the activation derivative) may be
 * NULL for relu / leaky-relu layers WITHOUT a residual input: the sign of the pre-activation is then recomputed from x, mean,
 * rstd, gamma and `beta` (one tensor read less in both backward passes); `beta` is only read in that case. */
int ss_norm_bwd(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                const float* gamma, const float* beta, const float* mean, const float* rstd,
                void* dx, int32_t dx_cstride, int accumulate_dx, void* dres, int accumulate_dres,
                float* dgamma, float* dbeta, int accumulate_params,
                void* ws, size_t ws_bytes, void* stream);

/* Two-phase forms for DATA-PARALLEL batch statistics (new: the reference has no working multi-GPU path).
 * `sums` = [groups*c*2] raw sums (fwd: sum x, sum x^2; bwd: sum g, sum g*xhat).  The caller all-reduces (SUM) them over
 * the ranks and passes the GLOBAL element count per (group, channel); with one rank they reproduce ss_norm_fwd/bwd.
 * In backward, dgamma/dbeta are built from the LOCAL sums (the gradient all-reduce sums them over ranks afterwards). */
int ss_norm_fwd_stats(const ss_norm_desc* d, const void* x, float* sums, void* ws, size_t ws_bytes, void* stream);
int ss_norm_fwd_finish(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                       const float* sums, int64_t total_count, float* mean, float* rstd,
                       float* moving_mean, float* moving_var, float momentum, void* stream);
int ss_norm_bwd_stats(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                      const float* mean, const float* rstd, float* sums, void* ws, size_t ws_bytes, void* stream);
int ss_norm_bwd_finish(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* global_sums, const float* local_sums, int64_t total_count,
                       void* dx, int32_t dx_cstride, int accumulate_dx, void* dres, int accumulate_dres,
                       float* dgamma, float* dbeta, int accumulate_params, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / pooling on NHWC views (rows = n*h*w pixels, c channels, explicit pixel strides).
 * ---------------------------------------------------------------------------------------- */
/* dx = dy * act'(.) expressed through the forward OUTPUT y (tanh, sigmoid, relu, lrelu):
 * keras.layers.Activation("tanh") CycleGAN.py:420, LeakyReLU(0.2) CycleGAN.py:433. */
int ss_act_bwd(int act, float act_alpha, const float* dy, int32_t dy_cstride, const float* y, int32_t y_cstride,
               float* dx, int32_t dx_cstride, int64_t rows, int32_t c, void* stream);
/* out = alpha*a + beta*b  (b may be NULL) -- gradient accumulation where a tensor has two consumers */
int ss_axpby(float alpha, const float* a, int32_t a_cstride, float beta, const float* b, int32_t b_cstride,
             float* out, int32_t out_cstride, int64_t rows, int32_t c, void* stream);
/* keras.layers.MaxPooling2D(pool_size=(2,2)) UNet_Segmentation.py:525,529,533,537 */
int ss_maxpool2x2_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride,
                      int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
/* dx (+)= routed dy (first maximum in row-major window order wins, as torch max_pool2d) */
int ss_maxpool2x2_bwd(const float* dy, int32_t dy_cstride, const float* x, int32_t x_cstride,
                      float* dx, int32_t dx_cstride, int accumulate,
                      int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
/* keras.ops.pad(mode="reflect") as a standalone op -- the pre-padding of tiles whose size is not a multiple of
 * 2^n_down (CycleGAN.py:365-367) or of 16 (UNet_Segmentation.py:520-522); (h,w) describe x.  bwd folds dy back onto x. */
int ss_reflect_pad2d_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                         int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream);
int ss_reflect_pad2d_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                         int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream);
/* keras.layers.Cropping2D (UNet_Segmentation.py:554): y = x[:, top:top+oh, left:left+ow]; (h,w) describe x */
int ss_crop2d_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                  int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream);
int ss_crop2d_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                  int32_t c, int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream);
/* keras.layers.UpSampling2D(size=(2,2)), nearest (CycleGAN.py:349, resize-convolution branch); (h,w) describe x */
int ss_upsample2x_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int ss_upsample2x_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                      int32_t c, void* stream);
int ss_copy(const float* src, int32_t src_cstride, float* dst, int32_t dst_cstride, int64_t rows, int32_t c, void* stream);
int ss_fill(float* dst, float value, int64_t count, void* stream);
/* `bytes` zero bytes at dst (gradient buffers of any storage type; hipMemsetAsync on the stream). */
int ss_zero(void* dst, size_t bytes, void* stream);

/* The same operations on bf16 / fp16 stored activations: leading ss_dtype argument, `void*` views.  The fp32 entry points above are
 * these with SS_DTYPE_F32. */
int ss_act_bwd_t(int32_t dtype, int act, float act_alpha, const void* dy, int32_t dy_cstride, const void* y, int32_t y_cstride,
                 void* dx, int32_t dx_cstride, int64_t rows, int32_t c, void* stream);
int ss_axpby_t(int32_t dtype, float alpha, const void* a, int32_t a_cstride, float beta, const void* b, int32_t b_cstride,
               void* out, int32_t out_cstride, int64_t rows, int32_t c, void* stream);
/* out = scale * a .* b on [rows][c] views: keras.layers.Dropout with an explicit keep mask b in {0, 1} and scale = 1 / (1 - rate)
 * (WassersteinGAN.py:566-567, 621), and the mask products of the gradient-penalty chain. */
int ss_mul_t(int32_t dtype, float scale, const void* a, int32_t a_cstride, const void* b, int32_t b_cstride, void* out, int32_t out_cstride,
             int64_t rows, int32_t c, void* stream);
/* Gradient penalty of WGAN_GP.gradient_penalty (WassersteinGAN.py:113-116) on g = d critic / d interpolated, [n][per_sample] dense:
 * norms[i] = sqrt(sum_j g_ij^2); gbar (optional) = d/dg of coef * sum_i (norms[i] - 1)^2, i.e. coef * 2 (norm_i - 1) / norm_i * g_ij
 * (0 where norm_i == 0).  Deterministic (fixed-order reduction, one workgroup per sample). */
/* interpolated[i] = real[i] + alpha[i] * (fake[i] - real[i]) for n dense samples of per_sample elements (WassersteinGAN.py:97-99). */
int ss_wgan_interpolate(const float* real, const float* fake, const float* alpha, float* out, int64_t n, int64_t per_sample, void* stream);
int ss_wgan_gp_grad(const float* g, int64_t n, int64_t per_sample, float coef, float* gbar, float* norms, void* stream);
int ss_copy_t(int32_t dtype, const void* src, int32_t src_cstride, void* dst, int32_t dst_cstride, int64_t rows, int32_t c, void* stream);

/* Image buffer of generated images, the device half of ImagePool.query (CycleGAN.py:927-964) in ONE launch.  The decisions stay on
 * the host, where the reference draws them from python's `random` (one uniform + one randint per image once the buffer is full):
 * for each of the k images of this query, mode[i] says what happens and slot[i] names the buffer slot concerned --
 *   SS_POOL_PASS  (0): out[i] = images[i]                                        (p <= 0.5: the current image is returned)
 *   SS_POOL_FILL  (1): pool[slot[i]] = images[i]; out[i] = images[i]             (the buffer is not full yet)
 *   SS_POOL_SWAP  (2): out[i] = pool[slot[i]]; pool[slot[i]] = images[i]         (p > 0.5: an old image is returned, the new one kept)
 * pool: [pool_size][bytes_per_image], images: [>= k][bytes_per_image], out: [k][bytes_per_image]; k <= SS_POOL_MAX_QUERY; two
 * images of one query must not name the same slot (the caller splits such a query: the second swap must see the first one's
 * store).  Byte copies: bit-exact for any storage type. */
#define SS_POOL_PASS 0
#define SS_POOL_FILL 1
#define SS_POOL_SWAP 2
#define SS_POOL_MAX_QUERY 16
int ss_pool_query(void* pool, const void* images, void* out, int64_t bytes_per_image, int32_t k, const int32_t* mode,
                  const int32_t* slot, int32_t pool_size, void* stream);
int ss_maxpool2x2_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride,
                        int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int ss_maxpool2x2_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, const void* x, int32_t x_cstride,
                        void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int ss_reflect_pad2d_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w,
                           int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream);
int ss_reflect_pad2d_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n,
                           int32_t h, int32_t w, int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream);
int ss_crop2d_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                    int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream);
int ss_crop2d_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h,
                    int32_t w, int32_t c, int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream);
int ss_upsample2x_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w,
                        int32_t c, void* stream);
int ss_upsample2x_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h,
                        int32_t w, int32_t c, void* stream);
/* dst (dst_dtype view) = src (src_dtype view): the boundary between fp32 and 16-bit stored tensors (network inputs / outputs,
 * and the fp32 staging of the convolution paths that have no native 16-bit kernel yet) */
int ss_convert(const void* src, int32_t src_dtype, int32_t src_cstride, void* dst, int32_t dst_dtype, int32_t dst_cstride,
               int64_t rows, int32_t c, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses (Keras 'sum_over_batch_size' = mean over every element).  Each writes the scalar loss to
 * *loss_out (device memory) and, if grad != NULL, grad = grad_scale * d loss / d pred.
 * ws: ss_loss_workspace_bytes(count).
 * ---------------------------------------------------------------------------------------- */
size_t ss_loss_workspace_bytes(int64_t count);
/* mean((target - pred)^2): keras.losses.MeanSquaredError against ones/zeros, CycleGAN.py:301-308 */
int ss_loss_mse_const(const float* pred, int64_t count, float target, float grad_scale,
                      float* loss_out, float* grad, void* ws, size_t ws_bytes, void* stream);
/* mean(|truth - pred|): keras.losses.MeanAbsoluteError, CycleGAN.py:103-106,644-650 */
int ss_loss_mae(const float* truth, const float* pred, int64_t count, float grad_scale,
                float* loss_out, float* grad, void* ws, size_t ws_bytes, void* stream);
/* class-weighted BCE + metrics, UNet_Segmentation.py:379-384,395.  out3 = {loss, mae, binary acc@0.5} */
int ss_loss_weighted_bce(const float* truth, const float* pred, int64_t count, float weighting, float grad_scale,
                         float* out3, float* grad, void* ws, size_t ws_bytes, void* stream);

/* pred / truth / grad stored as `dtype` (loss value and metrics: fp32 device scalars) */
int ss_loss_mse_const_t(int32_t dtype, const void* pred, int64_t count, float target, float grad_scale,
                        float* loss_out, void* grad, void* ws, size_t ws_bytes, void* stream);
int ss_loss_mae_t(int32_t dtype, const void* truth, const void* pred, int64_t count, float grad_scale,
                  float* loss_out, void* grad, void* ws, size_t ws_bytes, void* stream);
/* Multi-class head of the MultiResUNet (UNet_Segmentation.py:558-560): softmax over the c channels of each of `rows` pixels and its
 * backward dx = y .* (dy - <dy, y>); and the reference's loss closure on a c-channel output (UNet_Segmentation.py:379-384, dense
 * [rows][c] tensors): per pixel the channel-mean BCE, broadcast over the channels and weighted with truth * (weighting - 1) + 1,
 * mean over everything; out3 = {loss, mae, categorical accuracy}; grad (optional) = d (grad_scale * loss) / d pred. */
int ss_softmax_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int64_t rows, int32_t c, void* stream);
int ss_softmax_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, const void* y, int32_t y_cstride, void* dx, int32_t dx_cstride,
                     int64_t rows, int32_t c, void* stream);
int ss_loss_weighted_bce_mc_t(int32_t dtype, const void* truth, const void* pred, int64_t rows, int32_t c, float weighting, float grad_scale,
                              float* out3, void* grad, void* ws, size_t ws_bytes, void* stream);
int ss_loss_weighted_bce_t(int32_t dtype, const void* truth, const void* pred, int64_t count, float weighting, float grad_scale,
                           float* out3, void* grad, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * keras.optimizers.Adam as applied at CycleGAN.py:668-669,690-692 and by Model.fit for the UNet
 * (UNet_Segmentation.py:393): fused single pass over a flat parameter arena, the Keras update forms
 *   m += (g - m) * (1 - beta_1) ;  v += (g*g - v) * (1 - beta_2) ;  p -= alpha * m / (sqrt(v) + epsilon)
 *   alpha = lr*sqrt(1-beta_2^t)/(1-beta_1^t) is computed by the caller (t = iterations + 1).
 * The hyper-parameters are DOUBLES: Keras forms (1 - beta) in double precision and casts the result to fp32 (1 - 0.999 ->
 * fp32(0.001); 1.f - fp32(0.999) would be 1.3e-5 off).  grad_scale multiplies g first (1/world_size after a sum all-reduce).
 * ---------------------------------------------------------------------------------------- */
int ss_adam_keras(float* p, const float* g, float* m, float* v, int64_t count,
                  double alpha, double beta_1, double beta_2, double epsilon, float grad_scale, void* stream);
/* The same update with the step size read from DEVICE memory when the kernel runs (*alpha_dev = fp32(alpha), written by the caller
 * before the launch is replayed): what a captured hipGraph of a train step needs, since alpha changes with every iteration. */
int ss_adam_keras_dev(float* p, const float* g, float* m, float* v, int64_t count,
                      const float* alpha_dev, double beta_1, double beta_2, double epsilon, float grad_scale, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SEMSEG_HIP_H */
