// Shared helpers for libsemseg_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/semseg_hip.h"
struct SsTuning;
void ss_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define SS_LAUNCH_CHECK()                                                                   \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            ss_set_error("%s:%d: HIP launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return SS_ERR_LAUNCH;                                                           \
        }                                                                                   \
    } while (0)

static inline size_t ss_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Activation with a RUN-TIME (wave-uniform) code.  The piecewise-linear ones (none / relu / leaky relu) -- every hot path -- are one
// branch-free form, v > 0 ? v : neg(v) with neg chosen by uniform selects: as a `switch` this function cost ~14 scalar instructions and
// several taken branches per ELEMENT wherever it was inlined into an unrolled epilogue or a norm loop.  Same values
// bit for bit (relu of a NaN is 0, leaky relu multiplies, none returns v).
__device__ __forceinline__ float ss_apply_act(float v, int act, float alpha) {
    if (act == SS_ACT_TANH) return tanhf(v);
    if (act == SS_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    const float lin = act == SS_ACT_LRELU ? alpha * v : v;
    const float neg = act == SS_ACT_RELU ? 0.f : lin;
    return v > 0.f ? v : neg;
}

// The same for the piecewise-linear activations only (none / relu / leaky relu), with the case analysis done ONCE by the caller:
// relu = (act == SS_ACT_RELU), slope = (act == SS_ACT_LRELU ? alpha : 1).  Bit for bit ss_apply_act's values; streaming kernels
// choose between a loop instantiated with this form and one with the generic form OUTSIDE their loops -- ss_apply_act inside an
// unrolled loop leaves the tanh / sigmoid tests as scalar branches per ELEMENT (norm_apply_kernel: 156 branches in the loop body).
__device__ __forceinline__ float ss_act_pwl(float v, bool relu, float slope) {
    const float neg = relu ? 0.f : v * slope;
    return v > 0.f ? v : neg;
}
__device__ __forceinline__ float ss_act_grad_pwl(float y, bool relu, float slope) {
    const float neg = relu ? 0.f : slope;
    return y > 0.f ? 1.f : neg;
}
__device__ __forceinline__ bool ss_act_is_pwl(int act) { return act != SS_ACT_TANH && act != SS_ACT_SIGMOID; }

// derivative of the activation expressed through the forward OUTPUT y
__device__ __forceinline__ float ss_act_grad_from_out(float y, int act, float alpha) {
    if (act == SS_ACT_TANH) return 1.f - y * y;
    if (act == SS_ACT_SIGMOID) return y * (1.f - y);
    const float neg = act == SS_ACT_RELU ? 0.f : (act == SS_ACT_LRELU ? alpha : 1.f);
    return y > 0.f ? 1.f : neg;
}

// ---- 4 x 4 transpose inside every group of four adjacent lanes (DPP quad permutes): in[r] of lane l = M[r][l]  ->  out[k] of lane l =
// M[l][k].  Used by the GEMM epilogues: the 32 x 32 MFMA C/D layout gives a lane ONE column and four consecutive rows per register
// quad; after the transpose a lane holds four consecutive columns of ONE row, i.e. one 16-byte store, with no trip through LDS
// (the LDS-transposed epilogue of gemm_x6p cost ~10 000 cycles per 256 x 128 tile: 16 dependent write -> read -> store rounds).
__device__ __forceinline__ f32x4 ss_quad_transpose(float a0, float a1, float a2, float a3, bool odd, bool hi) {
    // quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), [2,3,0,1] = 0x4E (lane ^ 2)
    const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, true));
    const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a1), 0xB1, 0xF, 0xF, true));
    const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a2), 0xB1, 0xF, 0xF, true));
    const float x3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a3), 0xB1, 0xF, 0xF, true));
    const float b0 = odd ? x1 : a0, b1 = odd ? a1 : x0, b2 = odd ? x3 : a2, b3 = odd ? a3 : x2;
    const float y0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b0), 0x4E, 0xF, 0xF, true));
    const float y1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b1), 0x4E, 0xF, 0xF, true));
    const float y2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b2), 0x4E, 0xF, 0xF, true));
    const float y3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b3), 0x4E, 0xF, 0xF, true));
    return f32x4{hi ? y2 : b0, hi ? y3 : b1, hi ? b2 : y0, hi ? b3 : y1};
}

// ---- exact 3-way bf16 split of fp32 values (x6 contraction, conv_mfma_x6.hip; also emitted by the Winograd weight transform)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float ss_sub1(float a, float b) {      // plain v_sub_f32: keeps the SLP vectoriser from forming v_pk_add_f32,
    float r;                                                     // which is slow next to an MFMA stream (MI355X_MICROARCH.md)
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// two fp32 values -> packed (h, m, l) bf16 pairs, round-to-nearest-even at every step (v_cvt_pk_bf16_f32 + v_sub_f32)
__device__ __forceinline__ void ss_split3x2(f32x2 v, unsigned int& h, unsigned int& m, unsigned int& l) {
    const bf16x2 hb = __builtin_convertvector(v, bf16x2);
    const f32x2 hf = __builtin_convertvector(hb, f32x2);
    const f32x2 r1 = {ss_sub1(v[0], hf[0]), ss_sub1(v[1], hf[1])};
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const f32x2 mf = __builtin_convertvector(mb, f32x2);
    const f32x2 r2 = {ss_sub1(r1[0], mf[0]), ss_sub1(r1[1], mf[1])};
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned int, hb);
    m = __builtin_bit_cast(unsigned int, mb);
    l = __builtin_bit_cast(unsigned int, lb);
}

// x3h scales: exponent e of a tensor maximum (max = f * 2^e, f in [0.5, 1)); 14 for an all-zero tensor; clamped from below so
// that the scale 2^(14-e) stays finite for subnormal maxima (the results of such tensors underflow in fp32 as well)
// A caller-owned amax slot (ss_conv_desc::x_amax / dy_amax, ss_norm_desc::y_amax / dx_amax) is SS_AMAX_STRIPES words, one per
// 256-byte line: the producers' per-workgroup atomics spread over that many L2 lines (thousands of atomics on ONE address cost
// ~10 ns each: 20 us for a 2048-workgroup launch), readers take the maximum of the stripes.
#define SS_AMAX_STRIPES 16
#define SS_AMAX_STRIDE 64          // words between stripes
__device__ __forceinline__ unsigned int ss_amax_load(const unsigned int* __restrict__ p, int stripes) {
    unsigned int m = p[0];
    for (int i = 1; i < stripes; ++i) m = max(m, p[i * SS_AMAX_STRIDE]);
    return m;
}

// block maximum of a non-negative per-thread value -> a caller's striped slot (bit pattern; non-negative floats order like unsigned
// ints, so the atomic max is order-independent).  One atomic per block, spread over the stripes.  ALL threads of the block call it.
__device__ __forceinline__ void ss_block_amax_to_slot(float m, unsigned int* __restrict__ slot) {
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float ss_bam_wm[16];
    if ((threadIdx.x & 63) == 0) ss_bam_wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = ss_bam_wm[0];
        for (int i = 1; i < (int)((blockDim.x + 63) >> 6); ++i) b = fmaxf(b, ss_bam_wm[i]);
        atomicMax(slot + (blockIdx.x % SS_AMAX_STRIPES) * SS_AMAX_STRIDE, __float_as_uint(b));
    }
}

__device__ __forceinline__ int ss_amax_exp(float amax) {
    int e = 14;
    if (amax > 0.f) (void)frexpf(amax, &e);
    return e < -100 ? -100 : e;
}

// Two fp32 values (already scaled) -> packed fp16 pairs: x = h + l (ss_split_h2), or x = h + l' / 2048 (ss_split_h2s: the low piece
// carried at 2^11 times its value, normal fp16 range down to 2^-28 of the scaled maximum).  Written on 2-vectors so that the two
// roundings of a pair become ONE v_cvt_pk_f16_f32 (round to nearest even, the same bits as two scalar conversions) and no shift /
// or is needed to pack them: 3 (4) VALU instructions per value instead of 4 (5) -- the split, not the matrix pipe, paces the
// gather kernels (profiles/r02_f_gather_wgrad_pmc_and_l3_probe.md).
typedef _Float16 ss_h2 __attribute__((ext_vector_type(2)));
typedef float ss_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ss_split_h2(float x0, float x1, unsigned int& hh, unsigned int& ll) {
    const ss_f2 v = {x0, x1};
    const ss_h2 h = __builtin_convertvector(v, ss_h2);
    const ss_f2 r = {x0 - (float)h[0], x1 - (float)h[1]};
    const ss_h2 l = __builtin_convertvector(r, ss_h2);
    hh = __builtin_bit_cast(unsigned int, h);
    ll = __builtin_bit_cast(unsigned int, l);
}
__device__ __forceinline__ void ss_split_h2s(float x0, float x1, unsigned int& hh, unsigned int& ll) {
    const ss_f2 v = {x0, x1};
    const ss_h2 h = __builtin_convertvector(v, ss_h2);
    const ss_f2 r = {(x0 - (float)h[0]) * 2048.f, (x1 - (float)h[1]) * 2048.f};
    const ss_h2 l = __builtin_convertvector(r, ss_h2);
    hh = __builtin_bit_cast(unsigned int, h);
    ll = __builtin_bit_cast(unsigned int, l);
}

// Reflection / zero padding index map. Returns -1 when the tap falls into zero padding.
__device__ __forceinline__ int ss_map_index(int i, int size, int reflect) {
    if (reflect) {
        if (i < 0) i = -i;
        if (i >= size) i = 2 * (size - 1) - i;
        return i;
    }
    return (i < 0 || i >= size) ? -1 : i;
}

// ---- internal gather-GEMM problem descriptions (conv_api.hip builds them) -------------------
#define SS_MAX_TAPS 64

struct GTap {
    int16_t dy, dx;    // offset added to the (strided) class-grid coordinate
    int32_t woff;      // element offset of this tap's [Cred x Cout] weight block
};

// out[n, yc*out_s+out_oy, xc*out_s+out_ox, co] (+)= act(bias[co] +
//     sum_t sum_ci in[n, map(yc*in_s+in_oy+dy_t), map(xc*in_s+in_ox+dx_t), ci] * w[woff_t + ci*ldb + co])
struct GConvParams {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    int32_t N, IH, IW, Cin, in_cs;
    int32_t OHc, OWc;             // class grid
    int32_t in_s, in_oy, in_ox;
    int32_t OH, OW, Cout, out_cs;
    int32_t out_s, out_oy, out_ox;
    int32_t ldb;
    int32_t reflect;
    int32_t act;
    float alpha;
    int32_t accumulate;
    int32_t nbatch;               // >= 1 independent problems of identical shape (Winograd: 16 transform positions);
    int64_t in_bs, w_bs, out_bs;  // element strides between the batched problems
    int32_t ntaps;
    const unsigned int* h_amax;   // x3h (conv_mfma_x6.hip): bit pattern of max|input| -- nullptr: three-piece bf16 arithmetic
    const unsigned int* h_amax2;  //      ... of max|weights| (always one word)
    int amax_stripes;             // h_amax is the maximum over this many words, SS_AMAX_STRIDE apart (0 / 1: one word)
    int32_t dtype;                // ss_dtype of `in` / `out` (the pointers are reinterpreted); only the tile kernels take 16-bit storage
    int32_t c1_dtype;             // one-channel layers (conv_c1.hip x3h kernels): ss_dtype of the MULTI-channel tensor (`out` of a 1 -> C problem, `in` of a
                                  // C -> 1 problem; the pointer is reinterpreted), the one-channel tensor is fp32.  0 = SS_DTYPE_F32
    float* stats;                 // optional: [N][stats_chunks][Cout][2] partial (sum, sum of squares) of the stored output for a following norm
    int32_t stats_chunks;         //   (ss_conv_desc::y_stats); only kernels that report chunks for the problem (gconv_stats_chunks) write it
    GTap taps[SS_MAX_TAPS];
};

// part[split][ (t,ca) ][cb] = sum over a pixel range of
//     a[n, map(yc*a_s+a_oy+dy_t), map(xc*a_s+a_ox+dx_t), ca] * b[n, yc, xc, cb]
struct WGradParams {
    const float* a;
    const float* b;
    float* part;
    int32_t N, AH, AW, Ca, a_cs;
    int32_t GH, GW, Cb, b_cs;     // grid == b's spatial dims
    int32_t a_s, a_oy, a_ox;
    int32_t reflect;
    int32_t splits, pix_per_split;
    int32_t nbatch;               // batched problems (Winograd); partials laid out [batch][split][M][Cb]
    int32_t x6;                   // 1: fp32-exact contraction on the bf16 matrix cores where the shape allows (conv_mfma_x6.hip)
    const unsigned int* h_amax;   // x3h: bit pattern of max|a| (one scale per operand tensor); nullptr: three-piece bf16 arithmetic
    const unsigned int* h_amax2;  //      ... of max|b|
    int amax_stripes, amax2_stripes;   // each maximum is spread over this many words, SS_AMAX_STRIDE apart (0 / 1: one word)
    int64_t a_bs, b_bs;
    int32_t dtype;                // ss_dtype of `a` / `b` (tile kernel only); partials and dw are fp32
    int32_t dbg;                  // measurement only (tile_dbg): wgrad_x6 phase skipping
    int32_t ntaps;
    GTap taps[SS_MAX_TAPS];       // woff = destination offset of the tap block inside dw
};

// batched NT GEMM on pre-split bf16 planes (gemm_x6p.hip): C[batch][split][m][n] = sum_k A[batch][m][k] * B[batch][n][k]
#define SS_X6P_BM 256
#define SS_X6P_BN 128
struct X6PParams {
    const unsigned short* a;      // [3 planes][batch][rows padded to SS_X6P_BM][lda]
    const unsigned short* b;      // [3 planes][batch][rows padded to SS_X6P_BN][ldb]
    float* c;                     // [batch][split][M][ldc]
    int32_t M, N, K, nbatch, splits, k_per_split;
    int32_t lda, ldb, ldc;
    int32_t plain_l;              // x3h planes carry the low piece at its own magnitude (x*s = h + l): 256 x 256 tiles, one accumulator set (gemm_x6p.hip WIDE)
    int32_t fp16x2;               // 0: three bf16 planes, six products (x6);  1: two fp16 planes h + 2^-11 l, three products (x3h)
    int32_t dbg;                  // measurement only (tile_dbg & 64: skip the B operand's LDS-DMA; & 128: skip A's): results are wrong
    int64_t a_plane, b_plane, a_bs, b_bs, c_bs, c_ss;
    // one-plane products only (fp16x2 == 2): C stored as fp16 (c points at halfs, ldc / c_bs / c_ss count elements) after a multiplication
    // by the power of two c_scale that keeps the largest possible |sum| (2^28 K) inside fp16's range
    int32_t c16;
    float c_scale;
};
// Up to four gather problems of ONE launch (the sub-pixel phases of a stride-2 data gradient / transposed convolution: same input,
// same output tensor, same class grid; they differ in the output phase, their taps and their weight planes).  Workgroup slot s of
// an XCD takes problem s % count, tile s / count: the phases of one tile run side by side on one XCD and share its input rows in
// that L2; one launch instead of four.  count <= 1: a plain launch, nothing here is read.
#define SS_MAX_PHASES 4
#define SS_MAX_PHASE_TAPS 4
struct GPhases {
    int32_t count;
    int32_t out_oy[SS_MAX_PHASES], out_ox[SS_MAX_PHASES], ntaps[SS_MAX_PHASES], Ktot[SS_MAX_PHASES];
    int16_t tdy[SS_MAX_PHASES][SS_MAX_PHASE_TAPS], tdx[SS_MAX_PHASES][SS_MAX_PHASE_TAPS];
    const unsigned short* planes[SS_MAX_PHASES];
    int64_t plane_elems[SS_MAX_PHASES];
};

// C[batch][split][m][n] = sum_k A[batch][k][m] * B[batch][k][n] on K-major fp16 (h, l) planes (gemm_tn_x3h.hip)
struct TNParams {
    const unsigned short* a;      // [2 planes][batch][K][lda]   values a * 2^(14 - ea), ea = exponent(max|.| slot) + bound_a
    const unsigned short* b;      // [2 planes][batch][K][ldb]
    float* c;                     // [batch][split][M][N]
    int32_t M, N, K, nbatch, splits, k_per_split;
    int32_t lda, ldb;
    int64_t a_plane, b_plane, a_bs, b_bs;
    const unsigned int* amax_a;   // amax slots of the tensors the planes were derived from (striped or one word)
    const unsigned int* amax_b;
    int32_t stripes_a, stripes_b, bound_a, bound_b;
    // A planes that carry their own per-row power-of-two scales, folded into the B rows by the producer of B (the forward pass's
    // per-tile-scaled V planes, conv_wino.hip): ea = 14; B's exponent then also counts the slot amax_b2 (the tensor A came from)
    int32_t a_prescaled;
    const unsigned int* amax_b2;
    int32_t stripes_b2;
    int32_t one_plane;            // read the leading planes only, one product (16-bit activation storage, wino16_products = 1)
};
bool ss_gemm_tn_x3h_ok(int M, int N, long K);
int ss_gemm_tn_splits(int M, int N, long K, int nbatch, int* k_per_split);
int ss_gemm_tn_splits_max(int M, int N, long K, int nbatch);
int ss_launch_gemm_tn_x3h(const TNParams& p, hipStream_t s);
bool ss_x6p_enabled();
bool ss_x3h_enabled();
bool ss_x6p_wanted(long M, int N, int nbatch);
bool ss_x6p_wide_ok(long M, int N, int K, int nbatch);
int ss_launch_gemm_x6p(const X6PParams& p, hipStream_t s);

// kernel-selection switches: ONE explicit table, set through ss_config_set (config.hip); SS_* environment variables give the initial values
struct SsTuning { int x6, x3h, x3h_direct, x6p, winograd, wino_r, wgrad_c1, norm_fused_pix, gconv_fast, nt512, tile256, tile_conv, tile_th, tile_dbg, tile_stagger, weight_cache, wgrad_tn, gemm_persistent, gconv_v2, x6p_wide, twgrad_x3h, wino16_products, c1_mfma, x6p_pp, wino_save, gemm_ilv, gemm_cus, gconv_phases, phases_fused, phases_split, norm_order, norm_fuse_fin, wgrad_mfma_x6, wgrad_stage, x6p_wide1, wino16_m16, norm_bwd_resident, gemm_tn_rounds, gconv16_ragged; };
const SsTuning& ss_tuning();
// ss_prof_*: brackets the kernel launched inside this scope with HIP events on its stream when profiling is enabled (config.hip).
// flops = EXECUTED matrix-instruction FLOPs of the launch (all piece products), bytes = algorithmic HBM bytes (0 if not stated)
struct SsProfScope {
    int rec;
    hipStream_t stream;
    SsProfScope(const char* name, double flops, double bytes, hipStream_t s);
    ~SsProfScope();
};


// kernels / launchers implemented in the .hip files
int ss_launch_gconv_direct(const GConvParams& p, hipStream_t s);
int ss_launch_gconv_mfma(const GConvParams& p, hipStream_t s);
bool ss_gconv_mfma_ok(const GConvParams& p);
int ss_launch_wgrad_direct(const WGradParams& p, float* dw, int ldw, int accumulate, hipStream_t s);
int ss_launch_wgrad_mfma(const WGradParams& p, float* dw, int ldw, int accumulate, hipStream_t s);
// same, but only the first `rows` rows of the (ntaps*Ca) x Cb result are written to dw
int ss_launch_wgrad_mfma_rows(const WGradParams& p, float* dw, int ldw, int accumulate, int rows, hipStream_t s);
int ss_wgrad_mfma_splits(int64_t pixels, int M, int Cb, int* pix_per_split, int nbatch = 1);
int ss_launch_wgrad_mfma_partials(const WGradParams& p, hipStream_t s);   // partials only: part[batch][split][M][Cb]

// dst (dst_dtype view) = src (src_dtype view) (elementwise.hip)
// max|v| of a [rows][C] fp32 view (row stride cs) as a bit pattern, atomically raised in *out (conv_api.hip)
void ss_launch_amax_view(const float* v, long rows, int C, int cs, unsigned int* out, int stripes, hipStream_t s);
int ss_convert_launch(const void* src, int src_dtype, int src_cs, void* dst, int dst_dtype, int dst_cs, long rows, int c, hipStream_t s);
int ss_launch_wgrad_reduce(const WGradParams& p, float* dw, int ldw, int accumulate, int rows, hipStream_t s);

// LDS-staged tile kernels for small-channel stride-1 convs on large maps (conv_tile.hip): forward / data gradient (x3h arithmetic
// with per-tile scales) and weight gradient (fp32 MFMA, one partial per persistent workgroup)
bool ss_tconv_ok(const GConvParams& p);
size_t ss_tconv_ws(const GConvParams& p);
int ss_launch_tconv(const GConvParams& p, void* ws, size_t ws_bytes, hipStream_t s);
bool ss_twgrad_ok(const WGradParams& p);
int ss_twgrad_splits(const WGradParams& p);
int ss_launch_twgrad_partials(const WGradParams& p, hipStream_t s);

// LDS-tiled VALU kernels for stride-1 convs with one channel on one side (conv_c1.hip)
bool ss_conv_out1_ok(const GConvParams& p);
int ss_launch_conv_out1(const GConvParams& p, hipStream_t s);
bool ss_conv_in1_ok(const GConvParams& p);
int ss_launch_conv_in1(const GConvParams& p, hipStream_t s);
int ss_conv_in1_stats_chunks(const GConvParams& p);          // chunks per sample of GConvParams::stats the 1 -> C kernel writes (0: none)
// data gradient of a reflection-padded Cout == 1 layer with the fold applied to the one-channel side (conv_c1.hip)
bool ss_conv_in1_fold_ok(const GConvParams& p, int pt, int pl, int ih, int iw);
int ss_launch_conv_in1_fold(const GConvParams& p, int pt, int pl, int ih, int iw, hipStream_t s);
// weight gradients of the same layers: mode 0: Cout == 1 (X = x, S = dy), mode 1: Cin == 1 (X = dy, S = x)
bool ss_wgrad_c1_ok(int n, int xh, int xw, int C, int kh, int kw);
size_t ss_wgrad_c1_ws(int n, int xh, int xw, int C, int kh, int kw);
int ss_launch_wgrad_c1(int mode, const float* X, int X_cs, int C, int n, int xh, int xw, const float* S, int S_cs, int sh, int sw,
                       int kh, int kw, int pt, int pl, int reflect, float* dw, int accumulate, void* ws, hipStream_t s, int x_dtype = 0);
// the matrix-core one-channel kernels take the problem (the only ones with 16-bit loaders / stores for the multi-channel tensor)
bool ss_conv_out1_typed_ok(const GConvParams& p);
bool ss_conv_in1_typed_ok(const GConvParams& p);
bool ss_wgrad_c1_typed_ok(const void* X, int X_cs, int C, int xh, int xw, int kh, int kw, int pt, int pl);

// fp32-exact contraction on the bf16 matrix cores (conv_mfma_x6.hip): three bf16 pieces per operand, six products
bool ss_gconv_x6_ok(const GConvParams& p);                  // shape / alignment eligibility
bool ss_gconv_x6_typed_ok(const GConvParams& p);            // ... of a 16-bit stored problem for the typed loaders of gconv_x6_kernel
int ss_x6_npad(int cout);
size_t ss_gconv_x6_planes_bytes(const GConvParams& p);      // [3][nbatch][npad(Cout)][ntaps*Cin] bf16
int ss_launch_wprep_x6(const GConvParams& p, unsigned short* planes, hipStream_t s);
// all four sub-pixel phases of a stride-2 data gradient / transposed convolution in one workgroup per input tile (conv_phase.hip)
bool ss_gconv_phases_fused_ok(const GConvParams* ps, int count);
bool ss_gconv_phases_fused_wprob(const GConvParams* ps, int count, GConvParams* w);          // the one weight-plane problem of all the taps
int ss_launch_gconv_phases_fused(const GConvParams* ps, const unsigned short* planes, int count, hipStream_t s);
bool ss_gconv_x6v2_ok(const GConvParams& p);
int ss_gconv_x6v2_stats_chunks(const GConvParams& p);        // chunks per sample of GConvParams::stats gconv_x6v2 writes (0: none)
int ss_launch_gconv_x6_multi(const GConvParams* ps, const unsigned short* const* planes, int count, hipStream_t s);
int ss_launch_gconv_x6v2(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph = nullptr);
int ss_launch_gconv_x6(const GConvParams& p, const unsigned short* planes, hipStream_t s);
bool ss_wgrad_x6_ok(const WGradParams& p);
// stride-2 full-tap-box layers with the operands staged once per spatial tile (conv_wgrad_stage.hip)
bool ss_wgrad_stage_ok(const WGradParams& p);
int ss_wgrad_stage_splits(const WGradParams& p);          // 0: the shape is not taken
int ss_launch_wgrad_stage_partials(const WGradParams& p, hipStream_t s);
int ss_launch_wgrad_x6_partials(const WGradParams& p, hipStream_t s);

// Winograd F(2x2,3x3) path (conv_wino.hip): 3x3, stride 1; out[o] = sum_a in[map(o + a - pt)] * g[a]
// Weight cache (ss_wcache, include/semseg_hip.h): tagged regions of weight-derived operands kept across calls.
typedef ss_wcache WCache;
enum { SS_WC_WINO_UBF = 1, SS_WC_WINO_X3H_INV, SS_WC_WINO_X3H_PLANES, SS_WC_WINO_X6_PLANES, SS_WC_WINO_U32, SS_WC_WAMAX, SS_WC_WT,
       SS_WC_X6_PLANES };
// ---- batched weight preparation (ss_wprep_*; wprep_batch.hip) ---------------------------------------------------------------------
// A network's refresh of its weight-derived operands is ~100 small launches (weight maxima, tap-wise transposes, split planes,
// Winograd-transformed planes), the same ones with the same pointers every step.  While a RECORDER is active on the calling thread
// the hooked launch sites append a job instead of launching; ss_wprep_run then executes a recorded plan as one launch per job TYPE
// (workgroup -> job through a block map): ~6 launches per network and step.
enum { SS_WJ_AMAX = 0, SS_WJ_TRANSPOSE, SS_WJ_WPREP_H, SS_WJ_WPREP_3, SS_WJ_WINO_H, SS_WJ_TYPES };
struct SsWJob {          // POD: the plan is copied to device memory verbatim
    int32_t type, gx, gy, gz;          // grid of the launch this job replaces
    int32_t blk0, e;                   // first workgroup of the job inside its type's batched launch; e: a fifth type-specific integer
    const float* src;
    void* dst;
    void* dst2;
    const unsigned int* amax;
    int64_t n;
    int32_t a, b, c, d;                // type-specific integers (see the launch sites)
    GConvParams p;                     // SS_WJ_WPREP_*: the problem whose weights are split
};
bool ss_wrec_on();                                     // a recorder is active on this thread
void ss_wrec_push(const SsWJob& j);                    // append (dropped when j.dst is not a cache region noted since record_begin)
void ss_wrec_note_region(void* ptr, size_t bytes);     // a cache entry was allocated and expects a fill
void ss_wrec_unbatched();                              // a fill the plan cannot replay was launched directly: the recording is incomplete
// the batched launchers, next to the kernels they batch: jobs / map live in device memory (map[workgroup] = job index)
int ss_wbatch_launch_amax(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s);
int ss_wbatch_launch_transpose(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s);
int ss_wbatch_launch_wprep(bool h, const SsWJob* jobs, const int* map, int nblocks, hipStream_t s);
int ss_wbatch_launch_wino(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s);

static inline uint64_t ss_wc_tag(int kind, uint64_t detail) { return (uint64_t)kind | (detail << 8); }
// region of `n` bytes for the operand `tag`: the cached copy (*fill = false), a new cache entry (*fill = true) or, without a cache /
// when it is full, `fallback` (*fill = true)
static inline void* ss_wc_region(WCache* wc, uint64_t tag, size_t n, void* fallback, bool* fill) {
    *fill = true;
    if (!wc || !wc->base) return fallback;
    for (int i = 0; i < wc->count; ++i)
        if (wc->entry[i].tag == tag && wc->entry[i].bytes == n) { *fill = false; return (char*)wc->base + wc->entry[i].offset; }
    const size_t na = ss_align_up(n, 256);
    if (wc->count >= (int)(sizeof(wc->entry) / sizeof(wc->entry[0])) || wc->used + na > wc->bytes) return fallback;
    ss_wcache_entry& e = wc->entry[wc->count++];
    e.tag = tag; e.offset = wc->used; e.bytes = n;
    wc->used += na;
    wc->fills++;
    if (ss_wrec_on()) ss_wrec_note_region((char*)wc->base + e.offset, n);
    return (char*)wc->base + e.offset;
}

// normalisation applied in a convolution's operand load (ss_conv_desc::in_norm_*): groups == 0 -> off
struct InNorm {
    const float* mean = nullptr;
    const float* rstd = nullptr;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    int groups = 0, act = 0;
    float alpha = 0.f;
    unsigned int* amax_out = nullptr;      // forward: striped slot raised to max|normalised x| (nullptr: not wanted)
};

struct WinoProb {
    int n, h, w, cin, in_cs;      // gathered input (reduction channels = cin)
    int oh, ow, cout, out_cs;     // output grid
    int pt, pl, reflect;
    int bf16x3;                   // 1: the batched GEMMs run as split-bf16 (3 products) on the bf16 matrix cores (opt-in)
    int x6;                       // 1: forward / data-gradient GEMMs as fp32-exact 6-product bf16 contraction (conv_mfma_x6.hip)
    int fold_h, fold_w;           // > 0: the (oh, ow) grid is a shifted padded gradient that the output transform folds onto an
                                  // (fold_h x fold_w) tensor (reflect-pad data gradient, conv_wino.hip wino_output_kernel)
    WCache* wc = nullptr;         // transformed weights are kept here across calls when set
    // weight gradient on pre-split planes (ss_wino_wgrad_tn): maxima of x and dy (amax slots, striped or one word)
    const unsigned int* x_amax = nullptr;
    const unsigned int* dy_amax = nullptr;
    int x_stripes = 0, dy_stripes = 0;
    float* y_stats = nullptr;     // forward: partial (sum y, sum y^2) per sample / chunk / channel from the output transform
    InNorm in_norm;               // forward / weight gradient: x is a pre-normalisation tensor, normalised in the input transform
    void* saved = nullptr;        // ss_conv_desc::saved_operand (ss_wino_saved_bytes(q) bytes): forward writes its V planes + per-tile scales there, the weight gradient reads them
};
size_t ss_wino_saved_bytes(const WinoProb& q);          // 0: this problem's forward / weight-gradient pair keeps nothing
bool ss_wino_wgrad_tn(const WinoProb& q);
bool ss_wino_fwd_x3h(const WinoProb& q);          // the forward pass takes the x3h plane path (the one that fuses InNorm)
int ss_wino_stats_chunks(const WinoProb& q);

// C[b][m][n] = sum_k (Ah+Al)[b][m][k] * (Bh+Bl)[b][n][k], bf16 planes, fp32 output (gemm_bf16x3.hip)
struct BGemmParams {
    const unsigned short *ah, *al, *bh, *bl;
    float* c;
    int32_t M, N, K, nbatch;
    int64_t a_bs, b_bs, c_bs;     // element strides between batches
    int32_t lda, ldb, ldc;
};
int ss_launch_bgemm_bf16x3(const BGemmParams& p, hipStream_t s);
bool ss_wino_ok(const WinoProb& q);
size_t ss_wino_fwd_ws(const WinoProb& q);
int ss_wino_conv_fwd(const WinoProb& q, const float* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, float* y,
                     int act, float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int ss_wino_conv_fwd16(const WinoProb& q, int dtype, const void* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, void* y,
                       int act, float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
size_t ss_wino_wgrad_ws(const WinoProb& q);
int ss_wino_conv_wgrad(const WinoProb& q, const float* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                       hipStream_t s);
int ss_wino_conv_wgrad16(const WinoProb& q, int dtype, const void* x, const void* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t s);
