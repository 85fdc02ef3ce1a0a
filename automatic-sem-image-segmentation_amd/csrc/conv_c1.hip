// Stride-1 convolutions with ONE channel on one side, at full tile resolution -- the generator's 7x7 stem (1 -> F) and
// 7x7 head (F -> 1) and their data gradients.  They carry 0.8 % of the step's FLOPs but are HBM-bound layers (the 64-channel
// side is 0.5 GB at 512x512, batch 8) that map badly onto an implicit GEMM (K = 49 or N = 1), so they get LDS-tiled VALU
// kernels: the halo tile of the input is staged once in LDS, every thread keeps a strip of outputs in registers.
//
//   out1 : out[p]    = act(bias + sum_{t,c} in[map(p + d_t)][c] * w[t][c])          (Cout == 1)
//   in1  : out[p][c] = act(bias[c] + sum_t in[map(p + d_t)] * w[t][c])              (Cin == 1)
// Both take the generic GConvParams (taps = a full KH x KW box, in_s == out_s == 1, class grid == output grid).
#include "common.h"
#include <type_traits>

namespace {

// storage type of the MULTI-channel tensor of a one-channel layer (GConvParams::c1_dtype, C1WParams::x_dtype): four consecutive channels
// <-> f32x4 (the 16-bit types: one 8-byte access); the one-channel tensor is always fp32
typedef _Float16 c1_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 c1_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 c1_ld4(const float* q) { return *(const f32x4*)q; }
__device__ __forceinline__ f32x4 c1_ld4(const _Float16* q) { return __builtin_convertvector(*(const c1_f16x4*)q, f32x4); }
__device__ __forceinline__ f32x4 c1_ld4(const __bf16* q) { return __builtin_convertvector(*(const c1_bf16x4*)q, f32x4); }
__device__ __forceinline__ void c1_st4(float* q, f32x4 v) { *(f32x4*)q = v; }
__device__ __forceinline__ void c1_st4(_Float16* q, f32x4 v) { *(c1_f16x4*)q = __builtin_convertvector(v, c1_f16x4); }
__device__ __forceinline__ void c1_st4(__bf16* q, f32x4 v) { *(c1_bf16x4*)q = __builtin_convertvector(v, c1_bf16x4); }
__device__ __forceinline__ f32x4 c1_round(f32x4 v, const float*) { return v; }          // the value as stored
__device__ __forceinline__ f32x4 c1_round(f32x4 v, const _Float16*) { return __builtin_convertvector(__builtin_convertvector(v, c1_f16x4), f32x4); }
__device__ __forceinline__ f32x4 c1_round(f32x4 v, const __bf16*) { return __builtin_convertvector(__builtin_convertvector(v, c1_bf16x4), f32x4); }


constexpr int C1_TW = 64, C1_TH = 16;        // output tile of a 256-thread block: thread = 4 consecutive x in one row

struct C1Box { int dy0, dx0, kh, kw; };

// ---- many -> 1 ---------------------------------------------------------------------------------------------------------
// LDS: channel-planar halo tile xs[CC][TH+KH-1][HS] (a thread's reads are 16-byte rows of consecutive x: conflict free),
// weights of the chunk ws[CC][KH][8].  Per (channel, tap row): 3 ds_read_b128 of x + 2 broadcast reads of w for 4*KW FMAs.
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_out1_kernel(GConvParams p, C1Box box) {
    constexpr int CC = 8;
    constexpr int HR = C1_TH + KH - 1, HW = C1_TW + KW - 1, HS = C1_TW + 8;    // row stride: the 12-float strip reads end at column 4*15 + 11; 72 floats keep THREE blocks per CU in the LDS
    extern __shared__ __attribute__((aligned(16))) float smem_c1[];
    float* xs = smem_c1;                       // [CC][HR][HS]
    float* ws = smem_c1 + CC * HR * HS;        // [CC][KH][8]

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tiles_x = (p.OW + C1_TW - 1) / C1_TW, tiles_y = (p.OH + C1_TH - 1) / C1_TH;
    int b = blockIdx.x;
    const int bx = b % tiles_x; b /= tiles_x;
    const int by = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = bx * C1_TW, y0 = by * C1_TH;

    // tap index of every (row, column) of the box (the tap list may come in any order), once per block
    int* tapidx = (int*)(ws + CC * KH * 8);       // [KH][8]
    for (int idx = tid; idx < KH * 8; idx += 256) tapidx[idx] = -1;
    __syncthreads();
    for (int t = tid; t < p.ntaps; t += 256) tapidx[(p.taps[t].dy - box.dy0) * 8 + (p.taps[t].dx - box.dx0)] = t;

    // The halo tile of the NEXT channel chunk is requested into registers before the current chunk is multiplied (it used to be
    // loaded, stored and waited for between two barriers: 8 exposed round trips per tile), and stored after the barrier that
    // ends the chunk; the chunk's weights are gathered in one pass through the tap table (was: zero, barrier, scatter).
    constexpr int NU = (HR * HW * (CC / 4) + 255) / 256;          // float4 units per thread and chunk
    f32x4 pre[NU];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + 256 * u;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < HR * HW * (CC / 4)) {
                const int c4 = idx % (CC / 4);
                const int pix = idx / (CC / 4);
                const int rx = pix % HW, ry = pix / HW;
                const int iy = ss_map_index(y0 + ry + p.in_oy + box.dy0, p.IH, p.reflect);
                const int ix = ss_map_index(x0 + rx + p.in_ox + box.dx0, p.IW, p.reflect);
                if (iy >= 0 && ix >= 0 && c0 + 4 * c4 < p.Cin)
                    v = *(const f32x4*)(p.in + ((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs + c0 + 4 * c4);
            }
            pre[u] = v;
        }
    };
    prefetch(0);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p.Cin; c0 += CC) {
        __syncthreads();                               // the previous chunk's tile is no longer read (and tapidx is complete)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + 256 * u;
            if (idx < HR * HW * (CC / 4)) {
                const int c4 = idx % (CC / 4);
                const int pix = idx / (CC / 4);
                const int rx = pix % HW, ry = pix / HW;
#pragma unroll
                for (int e = 0; e < 4; ++e) xs[((4 * c4 + e) * HR + ry) * HS + rx] = pre[u][e];
            }
        }
        for (int idx = tid; idx < CC * KH * 8; idx += 256) {
            const int c = idx / (KH * 8), q = idx - c * (KH * 8);
            const int t = tapidx[q];
            ws[idx] = (t >= 0 && c0 + c < p.Cin) ? p.w[p.taps[t].woff + (long)(c0 + c) * p.ldb] : 0.f;
        }
        __syncthreads();
        if (c0 + CC < p.Cin) prefetch(c0 + CC);        // in flight during the multiplication below
#pragma unroll 2
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const float* xr = xs + (c * HR + ty + r) * HS + 4 * tx;
                const f32x4 xa = *(const f32x4*)xr, xb = *(const f32x4*)(xr + 4), xc = *(const f32x4*)(xr + 8);
                const float xv[12] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3], xc[0], xc[1], xc[2], xc[3]};
                const float* wr = ws + (c * KH + r) * 8;
                const f32x4 wa = *(const f32x4*)wr, wb = *(const f32x4*)(wr + 4);
                const float wv[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
                for (int s = 0; s < KW; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[j + s], wv[s], acc[j]);
            }
        }
    }
    const int oy = y0 + ty;
    if (oy >= p.OH) return;
    const float bv = p.bias ? p.bias[0] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = x0 + 4 * tx + j;
        if (ox >= p.OW) continue;
        float* op = p.out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs;
        float v = ss_apply_act(acc[j] + bv, p.act, p.alpha);
        if (p.accumulate) v += *op;
        *op = v;
    }
}

// ---- 1 -> many ---------------------------------------------------------------------------------------------------------
// LDS: single-channel halo tile xs[TH+KH-1][HS], weights ws[KH*KW][Cout (<= 64 per pass)].  Thread = 4 consecutive x in one
// row x 16 output channels; the 4 threads of a pixel strip cover 64 channels and write 256 contiguous bytes per pixel.
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_in1_kernel(GConvParams p, C1Box box) {
    constexpr int TW = 64, TH = 4;           // 256 threads = 16 strips x 4 rows x 4 channel groups
    constexpr int HR = TH + KH - 1, HW = TW + KW - 1, HS = (HW + 3 + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float xs[HR * HS];
    __shared__ __attribute__((aligned(16))) float ws[KH * KW * 64];

    const int tid = threadIdx.x;
    const int cg = tid & 3, tx = (tid >> 2) & 15, ty = tid >> 6;
    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    int b = blockIdx.x;
    const int bx = b % tiles_x; b /= tiles_x;
    const int by = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = bx * TW, y0 = by * TH;
    const int cbase = blockIdx.y * 64;       // 64 output channels per pass

    for (int idx = tid; idx < HR * HS; idx += 256) {
        const int rx = idx % HS, ry = idx / HS;
        float v = 0.f;
        if (rx < HW) {
            const int iy = ss_map_index(y0 + ry + p.in_oy + box.dy0, p.IH, p.reflect);
            const int ix = ss_map_index(x0 + rx + p.in_ox + box.dx0, p.IW, p.reflect);
            if (iy >= 0 && ix >= 0) v = p.in[((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs];
        }
        xs[idx] = v;
    }
    for (int idx = tid; idx < KH * KW * 64; idx += 256) ws[idx] = 0.f;
    __syncthreads();
    for (int idx = tid; idx < p.ntaps * 64; idx += 256) {
        const int c = idx & 63, t = idx >> 6;
        const int ry = p.taps[t].dy - box.dy0, rx = p.taps[t].dx - box.dx0;
        if (cbase + c < p.Cout) ws[(ry * KW + rx) * 64 + c] = p.w[p.taps[t].woff + cbase + c];
    }
    __syncthreads();

    f32x4 acc[4][4];                         // [pixel j][channel quad q] of channels cbase + 16*cg + 4*q ..
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < KH; ++r) {        // not unrolled: a fully unrolled 7x7 body hoists 196 LDS reads into registers
        const float* xr = xs + (ty + r) * HS + 4 * tx;
        const f32x4 xa = *(const f32x4*)xr, xb = *(const f32x4*)(xr + 4), xc = *(const f32x4*)(xr + 8);
        const float xv[12] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3], xc[0], xc[1], xc[2], xc[3]};
#pragma unroll
        for (int s = 0; s < KW; ++s) {
            const float* wr = ws + (r * KW + s) * 64 + 16 * cg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w4 = *(const f32x4*)(wr + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][q][0] = fmaf(xv[j + s], w4[0], acc[j][q][0]);
                    acc[j][q][1] = fmaf(xv[j + s], w4[1], acc[j][q][1]);
                    acc[j][q][2] = fmaf(xv[j + s], w4[2], acc[j][q][2]);
                    acc[j][q][3] = fmaf(xv[j + s], w4[3], acc[j][q][3]);
                }
            }
        }
    }
    const int oy = y0 + ty;
    if (oy >= p.OH) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ox = x0 + 4 * tx + j;
        if (ox >= p.OW) continue;
        float* op = p.out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs + cbase + 16 * cg;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = cbase + 16 * cg + 4 * q;
            if (co >= p.Cout) continue;
            f32x4 v = acc[j][q];
            if (p.bias) { const f32x4 bq = *(const f32x4*)(p.bias + co); v += bq; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ss_apply_act(v[e], p.act, p.alpha);
            if (p.accumulate) v += *(const f32x4*)(op + 4 * q);
            *(f32x4*)(op + 4 * q) = v;
        }
    }
}

// ---- 1 -> many on the fp16 matrix cores ------------------------------------------------------------------------------------
// The VALU kernel above spends 49 x 64 FMAs per pixel to produce 256 bytes: 396 us per launch at 512 x 512, batch 8, where the
// 537 MB it writes cost ~100 us.  Here the layer is the GEMM  out[c][pixel] = sum_k W[c][k] * U[k][pixel],  k = (a, b) = 8 a + b
// (rows of the tap box padded to 8 columns: K = 64 for a 7 x 7 box), on v_mfma_f32_32x32x16_f16 with the x3h arithmetic
// (x * s = h + l, products hh + hl + lh in one fp32 accumulator, conv_tile.hip):
//   A = weights, rows = 32 output channels; split once per workgroup (persistent) under one power-of-two scale,
//   B = U (im2col of the one-channel halo), columns = 32 consecutive pixels of one tile row; a lane's 8 k-values are 8 consecutive
//       floats of one halo row in LDS; split per WAVE-TILE under the power-of-two scale of that wave-tile's own maximum,
//   D: lane = pixel, registers = channels, 4 consecutive ones per group -> every lane stores 16-byte pieces of its pixel's row.
// FOLD (data gradient of a reflection-padded many -> 1 layer, e.g. the generator's 7x7 head): the gradient with respect to the
// PADDED input would be the plain case on the padded grid with zeros outside dy; the reflection's transpose adds the padded
// positions that mirror onto an image pixel.  Because the layer is linear in U, the fold is applied to U,
//       U[q][(a,b)] = sum over padded positions P that reflect onto q of dyz[P + d0 + (a,b)],
// (the same U as wgrad_c1_kernel MODE 0), so neither the padded gradient (550 MB) nor the fold pass over it exists.
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int c1_u32x4 __attribute__((ext_vector_type(4)));
struct C1Fold { int on, pt, pl, XH, XW; };
constexpr int I1_TW = 64, I1_TH = 8, I1_HS = 72, I1_FRONT = 8, I1_ES = 68;

__device__ __forceinline__ void c1_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// S: input stride (1, or 2: the discriminators' 4x4 stride-2 stem -- the halo tile is S x larger each way, a lane's 8 k-values are still 8
// consecutive floats of one halo row)
template <bool FOLD, int S = 1, typename TO = float>          // TO: storage type of the C-channel output
__global__ __launch_bounds__(256, 2) void conv_in1_x3h_kernel(GConvParams p, C1Box box, C1Fold f, int ntiles, int tiles_x, int tiles_y, int dbg) {
    constexpr int HS = S == 1 ? I1_HS : 2 * I1_TW + 8, HR = S * (I1_TH - 1) + 8;          // halo row stride / rows (boxes up to 8 x 8)
    static_assert(!FOLD || S == 1, "the folded data gradient is a stride-1 problem");
    __shared__ __attribute__((aligned(16))) float xs_all[I1_FRONT + HR * HS + 8];
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    __shared__ int tapidx[64];
    __shared__ float wred[4];
    __shared__ __attribute__((aligned(16))) float ep_s[4 * 32 * I1_ES];          // epilogue transposition, one 32 x 64 block per wave
    __shared__ float st_s[4][64][2];                                               // output statistics of a tile, per wave
    float* const xs = xs_all + I1_FRONT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int cbase = blockIdx.y * 64;
    const int OH = FOLD ? f.XH : p.OH, OW = FOLD ? f.XW : p.OW;
    const int nq = (box.kh + 1) >> 1;                       // K steps of 16 = two box rows

    if (tid < 64) { tapidx[tid] = -1; bias_s[tid] = (p.bias && cbase + tid < p.Cout) ? p.bias[cbase + tid] : 0.f; }
    for (int i = tid; i < I1_FRONT + HR * HS + 8; i += 256) xs_all[i] = 0.f;
    __syncthreads();
    if (tid < p.ntaps) tapidx[(p.taps[tid].dy - box.dy0) * 8 + (p.taps[tid].dx - box.dx0)] = tid;
    __syncthreads();

    // A fragments: lane (row = channel cb*32 + l31, lh) holds k = 16 q + 8 lh + e  <->  box row a = 2 q + lh, column b = e
    c1_u32x4 Ah[2][4], Al[2][4];
    int ew;
    {
        float wv[2][4][8];
        float wmax = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = tapidx[(2 * q + lh) * 8 + e];
                    const int c = cbase + cb * 32 + l31;
                    const float v = (t >= 0 && c < p.Cout) ? p.w[p.taps[t].woff + c] : 0.f;
                    wv[cb][q][e] = v;
                    wmax = fmaxf(wmax, fabsf(v));
                }
        for (int off = 32; off >= 1; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0) wred[wave] = wmax;
        __syncthreads();
        wmax = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        ew = ss_amax_exp(wmax);
        const float sw = ldexpf(1.f, 14 - ew);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned int hh, ll;
                    ss_split_h2(wv[cb][q][2 * j] * sw, wv[cb][q][2 * j + 1] * sw, hh, ll);
                    Ah[cb][q][j] = hh;
                    Al[cb][q][j] = ll;
                }
    }

    // halo of the NEXT tile in registers while the current one is multiplied
    constexpr int NPRE = (HR * HS + 255) / 256;
    float pre[NPRE];
    const int org_y = p.in_oy + box.dy0 + (FOLD ? f.pt : 0), org_x = p.in_ox + box.dx0 + (FOLD ? f.pl : 0);
    auto prefetch = [&](int tile) {
        const int n = tile / (tiles_y * tiles_x);
        const int tr = tile - n * tiles_y * tiles_x;
        const int y0 = (tr / tiles_x) * I1_TH, x0 = (tr % tiles_x) * I1_TW;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / HS, col = idx - r * HS;
            float v = 0.f;
            if (r < HR && r < S * (I1_TH - 1) + box.kh && col < S * (I1_TW - 1) + box.kw) {
                const int iy = ss_map_index(S * y0 + org_y + r, p.IH, p.reflect);
                const int ix = ss_map_index(S * x0 + org_x + col, p.IW, p.reflect);
                if (iy >= 0 && ix >= 0) v = p.in[((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs];
            }
            pre[u] = v;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) prefetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x);
        const int tr = tile - n * tiles_y * tiles_x;
        const int y0 = (tr / tiles_x) * I1_TH, x0 = (tr % tiles_x) * I1_TW;
        c1_lds_barrier();                                   // the previous tile's halo is no longer read
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int idx = tid + 256 * u;
            if (idx < HR * HS) xs[idx] = pre[u];
        }
        c1_lds_barrier();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};          // this lane's 4 channels over the pixels it stores (p.stats)
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const int ry = wave * 2 + (i >> 1), rx = (i & 1) * 32 + l31;
            const int qy = y0 + ry, qx = x0 + rx;
            const int xb0 = x0 + (i & 1) * 32;
            if (qy >= OH || xb0 >= OW) continue;            // wave-uniform
            float bv[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* r = xs + (S * ry + 2 * q + lh) * HS + S * rx;
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[q][e] = (q < nq && !(dbg & 1)) ? r[e] : (float)(e + q);          // dbg 1 (measurement): no im2col reads
            }
            if (FOLD && (qy <= f.pt || qy >= f.XH - 1 - f.pt || xb0 <= f.pl || xb0 + 31 >= f.XW - 1 - f.pl)) {
                // Border rows / columns (wave-uniform test): besides q itself at most ONE more padded row and ONE more padded column
                // mirror onto q (launcher: the image is larger than 2 * pad + 4), at the offsets oy / ox from q's own position; rows /
                // columns outside the staged halo are outside dy for those positions.  Only the pixels within `pad` of an edge
                // receive anything: the row term is wave-uniform, the column term runs for the (at most pad + 1) lanes concerned.
                const bool top = 2 * qy < f.XH, left = 2 * qx < f.XW;
                const bool ay = top ? (qy >= 1 && qy <= f.pt) : (qy <= f.XH - 2 && qy >= f.XH - 1 - f.pt);
                const bool ax = left ? (qx >= 1 && qx <= f.pl) : (qx <= f.XW - 2 && qx >= f.XW - 1 - f.pl);
                const int oy = top ? -2 * qy : 2 * (f.XH - 1 - qy);
                const int ox = left ? -2 * qx : 2 * (f.XW - 1 - qx);
                const int hr_n = I1_TH + box.kh - 1, hc_n = I1_TW + box.kw - 1;
                if (ay) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int hr = ry + 2 * q + lh + oy;
                        if (q < nq && hr >= 0 && hr < hr_n) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[q][e] += xs[hr * HS + rx + e];
                        }
                    }
                }
                if (ax) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int hr = ry + 2 * q + lh;
                        const bool yin = ay && hr + oy >= 0 && hr + oy < hr_n;
                        if (q < nq) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int hc = rx + e + ox;
                                if (hc >= 0 && hc < hc_n) {
                                    bv[q][e] += xs[hr * HS + hc];
                                    if (yin) bv[q][e] += xs[(hr + oy) * HS + hc];
                                }
                            }
                        }
                    }
                }
            }
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(bv[q][e]));
            for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const int eb = ss_amax_exp(m);
            const float sb = ldexpf(1.f, 14 - eb);
            f32x16 acc[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nq && !(dbg & 2)) {          // dbg 2: no split, no MFMAs
                    c1_u32x4 bh, bl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned int hh, ll;
                        ss_split_h2(bv[q][2 * j] * sb, bv[q][2 * j + 1] * sb, hh, ll);
                        bh[j] = hh;
                        bl[j] = ll;
                    }
                    const c1_f16x8 xh = __builtin_bit_cast(c1_f16x8, bh), xl = __builtin_bit_cast(c1_f16x8, bl);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const c1_f16x8 wh = __builtin_bit_cast(c1_f16x8, Ah[cb][q]), wl = __builtin_bit_cast(c1_f16x8, Al[cb][q]);
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[cb], 0, 0, 0);
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[cb], 0, 0, 0);
                    }
                }
            }
            // epilogue: a lane of the C/D layout holds its pixel's channels in 16-byte groups 32 bytes apart -- stored directly, every
            // instruction would touch 32 cache lines for 32 bytes each.  The wave's 32 x 64 block goes through a wave-private LDS
            // scratch and leaves as whole pixel rows: 16 lanes x 16 bytes = one pixel's 256 bytes, 4 pixels per instruction.
            if (dbg & 8) continue;                   // dbg 8: no epilogue at all
            const float inv = ldexpf(1.f, ew + eb - 28);
            float* const tb = ep_s + wave * (32 * I1_ES);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[cb][4 * rg + e] * inv;
                    *(f32x4*)(tb + l31 * I1_ES + cb * 32 + 8 * rg + 4 * lh) = v;
                }
            __builtin_amdgcn_wave_barrier();
            const int c = (lane & 15) * 4;
            if (cbase + c < p.Cout) {
                const f32x4 b4 = *(const f32x4*)(bias_s + c);
                TO* const orow = (TO*)p.out + ((long)(n * OH + qy) * OW + xb0) * p.out_cs + cbase + c;
                // ACC / PLAIN compile-time inside the store loop: a conditional `v += load` there makes the compiler wait for vmcnt(0)
                // around every store (each store then waits for the previous one's round trip AND for the halo prefetch)
                auto stores = [&](auto acc_c, auto plain_c) {
                    constexpr bool ACC = decltype(acc_c)::value, PLAIN = decltype(plain_c)::value;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int px = 4 * k + (lane >> 4);
                        if (xb0 + px >= OW) continue;
                        f32x4 v = *(const f32x4*)(tb + px * I1_ES + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = PLAIN ? v[e] + b4[e] : ss_apply_act(v[e] + b4[e], p.act, p.alpha);
                        TO* op = orow + (long)px * p.out_cs;
                        if (ACC) v += c1_ld4(op);
                        if (!(dbg & 4)) c1_st4(op, v);          // dbg 4: no global stores
                        if (!FOLD) {
                            const f32x4 sv = c1_round(v, (const TO*)nullptr);          // statistics of the STORED values
#pragma unroll
                            for (int e = 0; e < 4; ++e) { st1[e] += sv[e]; st2[e] = fmaf(sv[e], sv[e], st2[e]); }
                        }
                    }
                };
                if (p.accumulate) { if (p.act == SS_ACT_NONE) stores(std::true_type{}, std::true_type{}); else stores(std::true_type{}, std::false_type{}); }
                else { if (p.act == SS_ACT_NONE) stores(std::false_type{}, std::true_type{}); else stores(std::false_type{}, std::false_type{}); }
            }
        }
        if (!FOLD && p.stats) {
            // sum and sum of squares of the STORED values of this tile (one chunk of ss_conv_desc::y_stats), fixed order: lanes with the
            // same channels (16 apart), then the four waves
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st1[e] += __shfl_xor(st1[e], 16, 64); st1[e] += __shfl_xor(st1[e], 32, 64);
                st2[e] += __shfl_xor(st2[e], 16, 64); st2[e] += __shfl_xor(st2[e], 32, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { st_s[wave][4 * lane + e][0] = st1[e]; st_s[wave][4 * lane + e][1] = st2[e]; }
            }
            c1_lds_barrier();
            if (tid < 128) {
                const int ch = tid >> 1, k = tid & 1;
                const float v = (st_s[0][ch][k] + st_s[1][ch][k]) + (st_s[2][ch][k] + st_s[3][ch][k]);
                if (cbase + ch < p.Cout) p.stats[(((long)n * p.stats_chunks + tr) * p.Cout + cbase + ch) * 2 + k] = v;
            }
        }
    }
}

// ---- many -> 1 on the fp16 matrix cores -----------------------------------------------------------------------------------------
// The VALU kernel spends 49 x C FMAs per output pixel and re-stages the halo once per 8-channel chunk (3.3 GB of L2 traffic for a
// 0.55 GB input).  Here the channel contraction is ONE GEMM per halo pixel P, independent of the tap geometry,
//       Z[t][P] = sum_c W[t][c] * X[P][c],        t = 8 a + b (tap box rows padded to 8 columns: M = 64),  K = C,
// on v_mfma_f32_32x32x16_f16 with the x3h arithmetic (A = weights, split once per workgroup; B = pixels: a lane's 8 k-values are 8
// consecutive channels of ITS pixel straight from global memory, split under the power-of-two scale of that PIXEL's own maximum,
// which leaves the lane's accumulators as a per-lane factor), followed by the tap sum  out[q] = sum_{a,b} Z[(a,b)][q + (a,b)]  over
// LDS: a workgroup owns a 32-row x (64 - (kw-1))-column output tile, walks its halo two rows (4 waves x 32 pixels) at a time, writes
// the 64 x 128 block of Z tap-major into a double-buffered LDS stage (one barrier per step) and adds the 7 x 2 entries every output
// of the 8 rows concerned receives from it into the tile's output accumulator in LDS.  Every input element is read once per tile
// (halo overhead 1.3x, mostly L2 hits), LDS traffic is 450 bytes per halo pixel.
constexpr int O1_TH = 32, O1_ZS = 132;          // output rows per tile; floats per tap row of the Z stage (128 pixels + pad)

template <int CQ, typename TI = float>          // channel steps of 16: C == 16 * CQ (CQ = 2, 4); TI: storage type of the C-channel input
__global__ __launch_bounds__(256, 2) void conv_out1_x3h_kernel(GConvParams p, C1Box box, int ntiles, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float o1_lds[];
    float* const zs = o1_lds;                                   // [2][64 taps][O1_ZS]
    float* const oacc = o1_lds + 2 * 64 * O1_ZS;                // [O1_TH][64]
    int* const tapidx = (int*)(oacc + O1_TH * 64);              // [64]
    float* const wred = (float*)(tapidx + 64);                  // [4]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int TW = 64 - (box.kw - 1);
    const int HRN = O1_TH + box.kh - 1;                         // halo rows of a tile
    const int nsteps = (HRN + 1) >> 1;

    if (tid < 64) tapidx[tid] = -1;
    __syncthreads();
    if (tid < p.ntaps) tapidx[(p.taps[tid].dy - box.dy0) * 8 + (p.taps[tid].dx - box.dx0)] = tid;
    __syncthreads();

    // A fragments: lane (row = tap cb*32 + l31, lh) holds k = 16 q + 8 lh + e = channel
    c1_u32x4 Ah[2][CQ], Al[2][CQ];
    int ew;
    {
        float wv[2][CQ][8];
        float wmax = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int t = tapidx[cb * 32 + l31];
#pragma unroll
            for (int q = 0; q < CQ; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = t >= 0 ? p.w[p.taps[t].woff + (long)(16 * q + 8 * lh + e) * p.ldb] : 0.f;
                    wv[cb][q][e] = v;
                    wmax = fmaxf(wmax, fabsf(v));
                }
        }
        for (int off = 32; off >= 1; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0) wred[wave] = wmax;
        __syncthreads();
        wmax = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        ew = ss_amax_exp(wmax);
        const float sw = ldexpf(1.f, 14 - ew);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < CQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned int hh, ll;
                    ss_split_h2(wv[cb][q][2 * j] * sw, wv[cb][q][2 * j + 1] * sw, hh, ll);
                    Ah[cb][q][j] = hh;
                    Al[cb][q][j] = ll;
                }
    }
    const float bias = p.bias ? p.bias[0] : 0.f;

    // this wave's pixels of a step: halo row 2 * step + (wave >> 1), halo columns (wave & 1) * 32 + l31
    const int hrw = wave >> 1, hc = (wave & 1) * 32 + l31;
    f32x4 xr[CQ][2];                                            // the NEXT step's pixel (8 channels per 16-channel step), raw
    bool xvalid = false;
    auto fetch = [&](int n, int y0, int x0, int step) {
        const int hr = 2 * step + hrw;
        int iy = ss_map_index(y0 + p.in_oy + box.dy0 + hr, p.IH, p.reflect);
        int ix = ss_map_index(x0 + p.in_ox + box.dx0 + hc, p.IW, p.reflect);
        xvalid = hr < HRN && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;          // tiles past the image edge reflect out of range
        if (xvalid) {
            const TI* src = (const TI*)p.in + ((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs + 8 * lh;
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                xr[q][0] = c1_ld4(src + 16 * q);
                xr[q][1] = c1_ld4(src + 16 * q + 4);
            }
        }
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x);
        const int tr = tile - n * tiles_y * tiles_x;
        const int y0 = (tr / tiles_x) * O1_TH, x0 = (tr % tiles_x) * TW;
        fetch(n, y0, x0, 0);
        c1_lds_barrier();                                       // the previous tile's accumulator has been written out
        for (int i = tid; i < O1_TH * 64; i += 256) oacc[i] = 0.f;
#pragma unroll 1
        for (int step = 0; step < nsteps; ++step) {
            // ---- Z of this step's 128 pixels ----
            float xv[CQ][8];
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < CQ; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xv[q][e] = xvalid ? xr[q][e >> 2][e & 3] : 0.f;
                    m = fmaxf(m, fabsf(xv[q][e]));
                }
            if (step + 1 < nsteps) fetch(n, y0, x0, step + 1);  // in flight during the multiplication
            m = fmaxf(m, __shfl_xor(m, 32, 64));                // the pixel's two lanes
            const int eb = ss_amax_exp(m);
            const float sb = ldexpf(1.f, 14 - eb);
            f32x16 acc[2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                c1_u32x4 bh, bl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned int hh, ll;
                    ss_split_h2(xv[q][2 * j] * sb, xv[q][2 * j + 1] * sb, hh, ll);
                    bh[j] = hh;
                    bl[j] = ll;
                }
                const c1_f16x8 xh = __builtin_bit_cast(c1_f16x8, bh), xl = __builtin_bit_cast(c1_f16x8, bl);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const c1_f16x8 wh = __builtin_bit_cast(c1_f16x8, Ah[cb][q]), wl = __builtin_bit_cast(c1_f16x8, Al[cb][q]);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[cb], 0, 0, 0);
                }
            }
            // D: lane = pixel, register r of block cb = tap cb*32 + 8 (r>>2) + 4 lh + (r&3)  ->  tap-major stage, pixel contiguous
            const float inv = ldexpf(1.f, ew + eb - 28);
            float* const zb = zs + (step & 1) * (64 * O1_ZS) + hrw * 64 + hc;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) zb[(cb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * O1_ZS] = acc[cb][r] * inv;
            c1_lds_barrier();                                   // Z of this step complete (and the other buffer's readers of step-1 are past it)
            // ---- tap sum: the output rows 2 step - (kh-1) .. 2 step + 1 receive from halo rows 2 step, 2 step + 1 ----
            const float* const zr = zs + (step & 1) * (64 * O1_ZS);
            for (int item = tid; item < (box.kh + 1) * 64; item += 256) {
                const int x = item & 63;
                const int o = 2 * step - (box.kh - 1) + (item >> 6);
                if (x >= TW || o < 0 || o >= O1_TH) continue;
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int a = 2 * step + r - o;
                    if (a < 0 || a >= box.kh) continue;
                    const float* z = zr + (8 * a) * O1_ZS + r * 64 + x;
                    // eight unconditional, independent LDS reads (columns >= kw re-read the last one and are masked) instead of a
                    // runtime-bounded loop of dependent ones
                    float zz[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) { const int bb = b < box.kw ? b : box.kw - 1; zz[b] = z[bb * O1_ZS + bb]; }
#pragma unroll
                    for (int b = 0; b < 8; ++b) s += b < box.kw ? zz[b] : 0.f;
                }
                oacc[o * 64 + x] += s;
            }
        }
        c1_lds_barrier();                                       // all contributions are in
        for (int i = tid; i < O1_TH * 64; i += 256) {
            const int o = i >> 6, x = i & 63;
            const int oy = y0 + o, ox = x0 + x;
            if (x >= TW || oy >= p.OH || ox >= p.OW) continue;
            float* op = p.out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs;
            float v = ss_apply_act(oacc[i] + bias, p.act, p.alpha);
            if (p.accumulate) v += *op;          // (one store per thread and pass: the wait this costs is not on a store chain)
            *op = v;
        }
    }
}

// taps must fill a KH x KW box exactly once
bool tap_box(const GConvParams& p, C1Box* box) {
    if (p.ntaps < 1) return false;
    int dy0 = p.taps[0].dy, dy1 = dy0, dx0 = p.taps[0].dx, dx1 = dx0;
    for (int t = 1; t < p.ntaps; ++t) {
        dy0 = p.taps[t].dy < dy0 ? p.taps[t].dy : dy0; dy1 = p.taps[t].dy > dy1 ? p.taps[t].dy : dy1;
        dx0 = p.taps[t].dx < dx0 ? p.taps[t].dx : dx0; dx1 = p.taps[t].dx > dx1 ? p.taps[t].dx : dx1;
    }
    *box = C1Box{dy0, dx0, dy1 - dy0 + 1, dx1 - dx0 + 1};
    if (box->kh * box->kw != p.ntaps) return false;
    unsigned long long seen = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        const int i = (p.taps[t].dy - dy0) * box->kw + (p.taps[t].dx - dx0);
        if (seen >> i & 1ull) return false;
        seen |= 1ull << i;
    }
    return true;
}

bool plain_grid(const GConvParams& p) {
    return p.in_s == 1 && p.out_s == 1 && p.out_oy == 0 && p.out_ox == 0 && p.OHc == p.OH && p.OWc == p.OW && p.nbatch <= 1 &&
           (long)p.N * p.IH * p.IW * p.in_cs < (1L << 31);
}

template <int KH, int KW>
int launch_out1(const GConvParams& p, const C1Box& box, hipStream_t s) {
    constexpr int CC = 8, HR = C1_TH + KH - 1, HS = C1_TW + 8;
    const size_t smem = (size_t)(CC * HR * HS + CC * KH * 8 + KH * 8) * sizeof(float);
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)conv_out1_kernel<KH, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const unsigned blocks = (unsigned)(p.N * ((p.OH + C1_TH - 1) / C1_TH) * ((p.OW + C1_TW - 1) / C1_TW));
    hipLaunchKernelGGL((conv_out1_kernel<KH, KW>), dim3(blocks), dim3(256), smem, s, p, box);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <int KH, int KW>
int launch_in1(const GConvParams& p, const C1Box& box, hipStream_t s) {
    const unsigned blocks = (unsigned)(p.N * ((p.OH + 3) / 4) * ((p.OW + 63) / 64));
    hipLaunchKernelGGL((conv_in1_kernel<KH, KW>), dim3(blocks, (p.Cout + 63) / 64), dim3(256), 0, s, p, box);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// ---- weight gradient of the same layers -------------------------------------------------------------------------------------------
//   dw[(a,b)][c] = sum over the pixels of the C-channel tensor X of  X[n,y,x,c] * U[(n,y,x)][(a,b)]
//   MODE 0 (Cout == 1, the 7x7 head):  X = x (C = Cin), U[q][(a,b)] = sum over the padded positions that reflect onto q of
//           dy[pos - tap]  (the tap scatter of the one-channel gradient: the reflection is folded onto the one-channel side);
//   MODE 1 (Cin == 1, the 7x7 stem):   X = dy (C = Cout), U[p][(a,b)] = x[map(py + a - pt), map(px + b - pl)].
// The generic path wrote U for all pixels to HBM (436 MB, 0.4-0.7 ms) and ran a tall-skinny split-K GEMM over U and X (0.8 ms).
// Here U never leaves the CU: a persistent block walks 4 x 64 pixel tiles, stages the one-channel halo in LDS, builds the
// tile's U[256 pixels][64 taps (zero padded)] in LDS from it and contracts it with X on the fp32 matrix cores,
//        D[c][t] += sum_k X[k][c] * U[k][t]        (v_mfma_f32_32x32x2_f32, exact fp32: one 32 x 32 tile of D per wave)
// Both operands are read in their NATURAL layouts with 4-byte accesses (lane = channel for X straight from global memory, 16
// loads in flight per lane; lane = tap for U from LDS): at 64 cycles per MFMA there is nothing to gain from wider fragments.
// Deterministic: one partial per block, summed in block order by the reduce kernel.
struct C1WParams {
    const float* X; const float* S; float* part;
    int N, XH, XW, C, X_cs;        // the C-channel tensor and its pixel grid
    int SH, SW, S_cs;              // the one-channel tensor
    int kh, kw, pt, pl, reflect;
    int tiles_y, tiles_x, ntiles;
    int dbg;                       // measurement only (tile_dbg): 1 no X loads, 2 no U build, 4 no MFMAs, 8 no X split / stores
    int x_dtype;                   // ss_dtype of X (the matrix-core kernel only); S is fp32
};
constexpr int C1W_TH = 4, C1W_TW = 64, C1W_PIX = C1W_TH * C1W_TW, C1W_T = 64;

template <int MODE>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(C1WParams p) {
    extern __shared__ __attribute__((aligned(16))) float c1w_lds[];
    float* const us = c1w_lds;                       // U tile [C1W_PIX][C1W_T]
    float* const hs = c1w_lds + C1W_PIX * C1W_T;     // one-channel halo [TH + kh - 1][TW + kw - 1]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int c0 = blockIdx.y * 64 + (wave & 1) * 32;        // this wave's 32 channels
    const int t0 = (wave >> 1) * 32;                         // ... and 32 taps
    const int HR = C1W_TH + p.kh - 1, HW = C1W_TW + p.kw - 1;
    // the U entries this thread builds: tap tid & 63 of pixels (tid >> 6) + 4j
    const int ut = tid & 63;
    const int ua = ut / p.kw, ub = ut - ua * p.kw;
    const bool utap = ut < p.kh * p.kw;
    const int cc = c0 + l31 < p.C ? c0 + l31 : p.C - 1;      // clamped channel (rows >= C are not written)
    // B operand straight from the halo (tiles that need no folded / masked U): U[pixel][t] = halo[tapoff(t) + ry * HW + rx]
    int tapoff = 0;
    {
        const int bt = t0 + l31;
        if (bt < p.kh * p.kw) {
            const int ba = bt / p.kw, bb = bt - ba * p.kw;
            tapoff = MODE == 0 ? (p.kh - 1 - ba) * HW + (p.kw - 1 - bb) : ba * HW + bb;
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / (p.tiles_y * p.tiles_x);
        const int tr = tile - n * p.tiles_y * p.tiles_x;
        const int y0 = (tr / p.tiles_x) * C1W_TH, x0 = (tr % p.tiles_x) * C1W_TW;
        // halo of the one-channel tensor.  MODE 0: rows gy = hy0 + r of dy, zero outside;  MODE 1: rows map(hy0 + r) of x
        const int hy0 = MODE == 0 ? y0 + p.pt - (p.kh - 1) : y0 - p.pt;
        const int hx0 = MODE == 0 ? x0 + p.pl - (p.kw - 1) : x0 - p.pl;
        __syncthreads();                             // the previous tile's U / halo are no longer read
        for (int idx = tid; idx < HR * HW; idx += 256) {
            const int r = idx / HW, col = idx - r * HW;
            int sy = hy0 + r, sx = hx0 + col;
            if (MODE == 1) { sy = ss_map_index(sy, p.SH, p.reflect); sx = ss_map_index(sx, p.SW, p.reflect); }
            float v = 0.f;
            if (sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW) v = p.S[((long)(n * p.SH + sy) * p.SW + sx) * p.S_cs];
            hs[idx] = v;
        }
        __syncthreads();
        // whole tiles whose pixels receive nothing through the reflection (MODE 1: the halo is already the padded image) read
        // the halo directly; the others (26 % at 512x512) build the folded / masked U tile first
        const bool direct = y0 + C1W_TH <= p.XH && x0 + C1W_TW <= p.XW &&
                            (MODE == 1 || !p.reflect ||
                             (y0 > p.pt && y0 + C1W_TH - 1 < p.XH - 1 - p.pt && x0 > p.pl && x0 + C1W_TW - 1 < p.XW - 1 - p.pl));
        if (!direct) {
#pragma unroll 4
        for (int j = 0; j < C1W_PIX / 4; ++j) {
            const int px = (tid >> 6) + 4 * j;
            const int ry = px / C1W_TW, rx = px % C1W_TW;
            const int qy = y0 + ry, qx = x0 + rx;
            float u = 0.f;
            if (utap && qy < p.XH && qx < p.XW) {
                if (MODE == 1) {
                    u = hs[(ry + ua) * HW + (rx + ub)];
                } else if (!p.reflect || (qy > p.pt && qy < p.XH - 1 - p.pt && qx > p.pl && qx < p.XW - 1 - p.pl)) {
                    // pixels away from the border (no second padded position reflects onto them; wave-uniform: a wave's lanes are
                    // the taps of ONE pixel): one read, gy = qy + pt - a is always inside the halo
                    u = hs[(ry + p.kh - 1 - ua) * HW + (rx + p.kw - 1 - ub)];
                } else {
                    // border pixel: besides q itself at most ONE more padded row and ONE more padded column reflect onto q (the
                    // tile is shorter / narrower than half the image: launcher), i.e. up to four lookups of the plain formula
                    auto h = [&](int y, int x) -> float {
                        const int hr = y + p.pt - ua - hy0, hc = x + p.pl - ub - hx0;      // outside the halo = outside dy
                        return (hr >= 0 && hr < HR && hc >= 0 && hc < HW) ? hs[hr * HW + hc] : 0.f;
                    };
                    // (a pixel of the upper half cannot receive from the bottom padding and vice versa: image >= 2*pad + 4, launcher)
                    const bool top = 2 * qy < p.XH, left = 2 * qx < p.XW;
                    const bool ay = p.reflect && (top ? qy >= 1 : qy <= p.XH - 2);
                    const bool ax = p.reflect && (left ? qx >= 1 : qx <= p.XW - 2);
                    const int y2 = top ? -qy : 2 * (p.XH - 1) - qy;
                    const int x2 = left ? -qx : 2 * (p.XW - 1) - qx;
                    u = h(qy, qx);
                    if (ay) u += h(y2, qx);
                    if (ax) u += h(qy, x2);
                    if (ay && ax) u += h(y2, x2);
                }
            }
            us[px * C1W_T + ut] = u;
        }
        __syncthreads();
        }
        // contraction over the tile's 256 pixels, 32 per batch (16 MFMAs of K = 2): A = X[pixel][c] from global, B = U[pixel][t] from LDS
        // (the loads of batch k+1 are issued before the MFMAs of batch k: two register sets, fully unrolled)
        const float* ub_ = us + t0 + l31;
        const float* xb_ = p.X + (long)n * p.XH * p.XW * p.X_cs;      // wave-uniform base, 32-bit per-lane offsets (launcher: < 2^31 elements)
        float av0[16], bv0[16], av1[16], bv1[16], av2[16], bv2[16];
        auto fetch = [&](float (&a_)[16], float (&b_)[16], int k0) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int px = k0 + 2 * m + lh;
                int qy = y0 + px / C1W_TW, qx = x0 + px % C1W_TW;
                qy = qy < p.XH ? qy : p.XH - 1;              // pad pixels: U is zero there, any finite X will do
                qx = qx < p.XW ? qx : p.XW - 1;
                a_[m] = xb_[(unsigned)((qy * p.XW + qx) * p.X_cs + cc)];
                b_[m] = direct ? hs[tapoff + (px / C1W_TW) * HW + (px % C1W_TW)] : ub_[px * C1W_T];
            }
        };
        auto mma = [&](const float (&a_)[16], const float (&b_)[16]) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[m], b_[m], acc, 0, 0, 0);
        };
        // three register sets: the loads run TWO batches (32 MFMAs = 2048 cycles) ahead of their use
        constexpr int NB = C1W_PIX / 32;
        fetch(av0, bv0, 0);
        fetch(av1, bv1, 32);
#pragma unroll 1
        for (int kb = 0; kb < NB; kb += 3) {
            if (kb + 2 < NB) fetch(av2, bv2, (kb + 2) * 32);
            mma(av0, bv0);
            if (kb + 1 < NB) {
                if (kb + 3 < NB) fetch(av0, bv0, (kb + 3) * 32);
                mma(av1, bv1);
            }
            if (kb + 2 < NB) {
                if (kb + 4 < NB) fetch(av1, bv1, (kb + 4) * 32);
                mma(av2, bv2);
            }
        }
    }
    // D[row = channel][col = tap]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int t = t0 + l31;
    if (t < p.kh * p.kw) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.C) p.part[((long)blockIdx.x * p.kh * p.kw + t) * p.C + c] = acc[r];
        }
    }
}

// ---- the same weight gradient on the fp16 matrix cores ("x3h") --------------------------------------------------------------------
// K = the pixels of a tile, both operands K-major in LDS ([pixel][channel | tap]) as 16-byte units of 4 values = 8 B of h + 8 B of l,
// fragments by the transposing LDS read ds_read_b64_tr_b16 -- the scheme of twgrad_x3h_kernel (conv_tile.hip), with its running
// power-of-two unit for the accumulators that live across all tiles of a workgroup:
//  * X tile (128 pixels x 64 channels) is requested into registers one tile ahead, split under the power-of-two scale of the TILE's
//    maximum and stored as planes (no fp32 staging);
//  * U is built from the one-channel halo in LDS, 4 taps (one unit) of one pixel per thread and step, split under the scale of the
//    halo's maximum (x 4 where up to four padded positions fold onto a pixel) and stored as planes; it never exists in fp32;
//  * one 32 x 32 block of D[channel][tap] per wave, 8 K steps of 16 pixels per tile: 24 MFMAs (768 cycles) where the fp32 kernel
//    above issues 64 (4096 cycles) for its share of a 256-pixel tile.
constexpr int XW_TH = 2, XW_TW = 64, XW_PIX = XW_TH * XW_TW, XW_PS = 272;          // pixel stride in LDS: 16 units + 16 B (odd multiple of 16)
typedef short c1_s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 c1_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ c1_s16x4 c1_tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c1_s16x4*)p);
}

template <int MODE, typename TX = float>          // TX: storage type of the C-channel tensor X
__global__ __launch_bounds__(256, 2) void wgrad_c1_x3h_kernel(C1WParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xw_lds[];
    unsigned char* const sX = xw_lds;                              // [XW_PIX][XW_PS]
    unsigned char* const sU = xw_lds + XW_PIX * XW_PS;             // [XW_PIX][XW_PS]
    float* const hs = (float*)(xw_lds + 2 * XW_PIX * XW_PS);       // one-channel halo [XW_TH + kh - 1][XW_TW + kw - 1]
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int cb = wave & 1, tb = wave >> 1;                      // this wave's 32 channels / 32 taps of the 64 x 64 block
    const int c0 = blockIdx.y * 64;
    const int HR = XW_TH + p.kh - 1, HW = XW_TW + p.kw - 1;
    const int NT = p.kh * p.kw;

    // fragment addressing (bytes), as twgrad_x3h_kernel: a lane addresses unit `quad` of K row kq of its 16-lane group
    const int kq = (lane & 15) >> 2, quad = lane & 3, mh = (lane >> 4) & 1;
    const int a_base = (8 * lh + kq) * XW_PS + ((cb * 32 + 16 * mh + 4 * quad) >> 2) * 16;
    const int b_base = (8 * lh + kq) * XW_PS + ((tb * 32 + 16 * mh + 4 * quad) >> 2) * 16;

    // the unit of U this thread builds: taps 4 uu .. 4 uu + 3 of pixels tid / 16 + 16 j
    const int uu = tid & 15;
    int ta[4], tbx[4];
    bool tv[4];
    int toff[4];                   // halo offset of the tap relative to the pixel's own halo position (0 for taps past the box: masked)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = 4 * uu + k;
        tv[k] = t < NT;
        ta[k] = tv[k] ? t / p.kw : 0;
        tbx[k] = tv[k] ? t - ta[k] * p.kw : 0;
        toff[k] = !tv[k] ? 0 : (MODE == 1 ? ta[k] * HW + tbx[k] : (p.kh - 1 - ta[k]) * HW + (p.kw - 1 - tbx[k]));
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int E = -(1 << 20);

    // next tile's operands in registers: 8 units of X (pixel tid / 16 + 16 j, channels 4 (tid % 16) ..), 3 halo values
    f32x4 px_[8];
    float ph[3];
    auto prefetch = [&](int tile) {
        const int n = tile / (p.tiles_y * p.tiles_x);
        const int tr = tile - n * p.tiles_y * p.tiles_x;
        const int y0 = (tr / p.tiles_x) * XW_TH, x0 = (tr % p.tiles_x) * XW_TW;
        const int cu = c0 + 4 * uu;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = (tid >> 4) + 16 * j;
            const int qy = y0 + px / XW_TW, qx = x0 + px % XW_TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (qy < p.XH && qx < p.XW && cu < p.C && !(p.dbg & 1)) v = c1_ld4((const TX*)p.X + ((long)(n * p.XH + qy) * p.XW + qx) * p.X_cs + cu);
            if (p.dbg & 1) v[0] = 1.f;
            px_[j] = v;
        }
        const int hy0 = MODE == 0 ? y0 + p.pt - (p.kh - 1) : y0 - p.pt;
        const int hx0 = MODE == 0 ? x0 + p.pl - (p.kw - 1) : x0 - p.pl;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx / HW, col = idx - r * HW;
            int sy = hy0 + r, sx = hx0 + col;
            if (MODE == 1) { sy = ss_map_index(sy, p.SH, p.reflect); sx = ss_map_index(sx, p.SW, p.reflect); }
            float v = 0.f;
            if (r < HR && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW) v = p.S[((long)(n * p.SH + sy) * p.SW + sx) * p.S_cs];
            ph[j] = v;
        }
    };

    int tile = blockIdx.x;
    if (tile < p.ntiles) prefetch(tile);
    for (; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / (p.tiles_y * p.tiles_x);
        const int tr = tile - n * p.tiles_y * p.tiles_x;
        const int y0 = (tr / p.tiles_x) * XW_TH, x0 = (tr % p.tiles_x) * XW_TW;
        // tile maxima from the registers
        float ma = 0.f, mu = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ma = fmaxf(ma, fmaxf(fmaxf(fabsf(px_[j][0]), fabsf(px_[j][1])), fmaxf(fabsf(px_[j][2]), fabsf(px_[j][3]))));
#pragma unroll
        for (int j = 0; j < 3; ++j) mu = fmaxf(mu, fabsf(ph[j]));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, off, 64)); mu = fmaxf(mu, __shfl_xor(mu, off, 64)); }
        c1_lds_barrier();                                        // the previous tile's fragments have been read (and `red`)
        if (lane == 0) { red[wave] = ma; red[4 + wave] = mu; }
        c1_lds_barrier();
        ma = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        mu = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        const bool live = ma > 0.f && mu > 0.f;                  // uniform over the workgroup
        float sa = 0.f, sb = 0.f;
        if (live) {
            const int ea = ss_amax_exp(ma), eb = ss_amax_exp(mu) + ((MODE == 0 && p.reflect) ? 2 : 0);
            const int et = ea + eb;
            int shift = 0;
            if (et > E) {
                if (E > -(1 << 19)) {
                    const float fr = ldexpf(1.f, (E - et) < -126 ? -126 : (E - et));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] *= fr;
                }
                E = et;
            } else {
                shift = et - E;
                if (shift < -60) shift = -60;
            }
            sa = ldexpf(1.f, 14 - ea);
            sb = ldexpf(1.f, 14 - eb + shift);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (p.dbg & 8) break;
                const int px = (tid >> 4) + 16 * j;
                c1_f16x4 h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float x = px_[j][k] * sa; h[k] = (_Float16)x; l[k] = (_Float16)(x - (float)h[k]); }
                unsigned char* u = sX + px * XW_PS + uu * 16;
                *(c1_f16x4*)u = h;
                *(c1_f16x4*)(u + 8) = l;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int idx = tid + 256 * j;
                if (idx < HR * HW) hs[idx] = ph[j];
            }
        }
        if (tile + (int)gridDim.x < p.ntiles) prefetch(tile + gridDim.x);      // in flight while U is built and the tile is multiplied
        c1_lds_barrier();                                        // halo visible
        if (live && !(p.dbg & 2)) {
            const int hy0 = MODE == 0 ? y0 + p.pt - (p.kh - 1) : y0 - p.pt;
            const int hx0 = MODE == 0 ? x0 + p.pl - (p.kw - 1) : x0 - p.pl;
            // interior tiles (75 % at 512 x 512) skip the fold terms with one uniform branch: evaluating the per-pixel border tests for
            // every pixel cost 45 of the kernel's 250 us
            const bool tile_border = MODE == 0 && (y0 <= p.pt || y0 + XW_TH - 1 >= p.XH - 1 - p.pt || x0 <= p.pl || x0 + XW_TW - 1 >= p.XW - 1 - p.pl);
            // The own-position reads are UNCONDITIONAL (a tap past the box reads offset 0 and is masked): guarded reads came out as
            // four serial LDS round trips per pixel.  Pixels outside the image need no mask: their X rows are zero.
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = (tid >> 4) + 16 * j;
                const int ry = px / XW_TW, rx = px % XW_TW;
                const float* hb = hs + ry * HW + rx;
                float u4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) u4[k] = hb[toff[k]];
#pragma unroll
                for (int k = 0; k < 4; ++k) u4[k] = tv[k] ? u4[k] : 0.f;
                if (MODE == 0 && p.reflect && tile_border && !(p.dbg & 16)) {          // (dbg 16, measurement: no fold terms)
                    const int qy = y0 + ry, qx = x0 + rx;
                    const bool top = 2 * qy < p.XH, left = 2 * qx < p.XW;
                    const bool ay = qy < p.XH && (top ? (qy >= 1 && qy <= p.pt) : (qy <= p.XH - 2 && qy >= p.XH - 1 - p.pt));
                    const bool ax = qx < p.XW && (left ? (qx >= 1 && qx <= p.pl) : (qx <= p.XW - 2 && qx >= p.XW - 1 - p.pl));
                    if (ay || ax) {
                        // border pixel: besides q itself at most ONE more padded row and ONE more padded column reflect onto q
                        const int y2 = top ? -qy : 2 * (p.XH - 1) - qy;
                        const int x2 = left ? -qx : 2 * (p.XW - 1) - qx;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (!tv[k]) continue;
                            auto h = [&](int y, int x) -> float {
                                const int hr = y + p.pt - ta[k] - hy0, hc = x + p.pl - tbx[k] - hx0;      // outside the halo = outside dy
                                return (hr >= 0 && hr < HR && hc >= 0 && hc < HW) ? hs[hr * HW + hc] : 0.f;
                            };
                            if (ay) u4[k] += h(y2, qx);
                            if (ax) u4[k] += h(qy, x2);
                            if (ay && ax) u4[k] += h(y2, x2);
                        }
                    }
                }
                c1_f16x4 h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float x = u4[k] * sb; h[k] = (_Float16)x; l[k] = (_Float16)(x - (float)h[k]); }
                unsigned char* u = sU + px * XW_PS + uu * 16;
                *(c1_f16x4*)u = h;
                *(c1_f16x4*)(u + 8) = l;
            }
        }
        c1_lds_barrier();                                        // planes complete
        if (live && !(p.dbg & 4)) {
#pragma unroll
            for (int ks = 0; ks < XW_PIX / 16; ++ks) {
                const unsigned char* ap = sX + ks * 16 * XW_PS + a_base;
                const unsigned char* bp = sU + ks * 16 * XW_PS + b_base;
                const c1_s16x4 ah0 = c1_tr4(ap), ah1 = c1_tr4(ap + 4 * XW_PS), al0 = c1_tr4(ap + 8), al1 = c1_tr4(ap + 4 * XW_PS + 8);
                const c1_s16x4 bh0 = c1_tr4(bp), bh1 = c1_tr4(bp + 4 * XW_PS), bl0 = c1_tr4(bp + 8), bl1 = c1_tr4(bp + 4 * XW_PS + 8);
                const c1_f16x8 ah = __builtin_bit_cast(c1_f16x8, __builtin_shufflevector(ah0, ah1, 0, 1, 2, 3, 4, 5, 6, 7));
                const c1_f16x8 al = __builtin_bit_cast(c1_f16x8, __builtin_shufflevector(al0, al1, 0, 1, 2, 3, 4, 5, 6, 7));
                const c1_f16x8 bh = __builtin_bit_cast(c1_f16x8, __builtin_shufflevector(bh0, bh1, 0, 1, 2, 3, 4, 5, 6, 7));
                const c1_f16x8 bl = __builtin_bit_cast(c1_f16x8, __builtin_shufflevector(bl0, bl1, 0, 1, 2, 3, 4, 5, 6, 7));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
        }
    }
    // D[row = channel][col = tap] in the unit 2^(E - 28): col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float unit = E > -(1 << 19) ? ldexpf(1.f, (E - 28) < -126 ? -126 : (E - 28)) : 0.f;
    const int t = tb * 32 + l31;
    if (t < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.C) p.part[((long)blockIdx.x * NT + t) * p.C + c] = acc[r] * unit;
        }
    }
}

// dw[t][c] (+)= sum over blocks (in block order) of part[blk][t][c]
__global__ __launch_bounds__(256) void wgrad_c1_reduce_kernel(const float* __restrict__ part, int nblk, int total, float* __restrict__ dw,
                                                              int accumulate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float s0 = 0.f;
    for (int k = 0; k < nblk; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k + u < nblk ? part[(long)(k + u) * total + e] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s0 += v[u];
    }
    dw[e] = accumulate ? dw[e] + s0 : s0;
}

// 2 blocks per CU; an ODD block count, so that the tiles of one image column (the border columns are slower) spread over all blocks
inline int c1w_blocks(long ntiles) { return (int)(ntiles < 511 ? ntiles : 511); }

template <int MODE>
int launch_wgrad_c1(const C1WParams& p, float* dw, int accumulate, hipStream_t s) {
    const size_t smem = ((size_t)C1W_PIX * C1W_T + (size_t)(C1W_TH + p.kh - 1) * (C1W_TW + p.kw - 1)) * sizeof(float);
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)wgrad_c1_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const int nblk = c1w_blocks(p.ntiles);
    hipLaunchKernelGGL((wgrad_c1_kernel<MODE>), dim3(nblk, (p.C + 63) / 64), dim3(256), smem, s, p);
    SS_LAUNCH_CHECK();
    const int total = p.kh * p.kw * p.C;
    hipLaunchKernelGGL(wgrad_c1_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p.part, nblk, total, dw, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <int MODE, typename TX>
int launch_wgrad_c1_x3h_t(const C1WParams& p, float* dw, int accumulate, hipStream_t s) {
    const size_t smem = (size_t)2 * XW_PIX * XW_PS + (size_t)(XW_TH + p.kh - 1) * (XW_TW + p.kw - 1) * sizeof(float);
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)wgrad_c1_x3h_kernel<MODE, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);      // + the static `red`
        return true;
    }();
    (void)attr_set;
    const int nblk = c1w_blocks(p.ntiles);
    hipLaunchKernelGGL((wgrad_c1_x3h_kernel<MODE, TX>), dim3(nblk, (p.C + 63) / 64), dim3(256), smem, s, p);
    SS_LAUNCH_CHECK();
    const int total = p.kh * p.kw * p.C;
    hipLaunchKernelGGL(wgrad_c1_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p.part, nblk, total, dw, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <int MODE>
int launch_wgrad_c1_x3h(const C1WParams& p, float* dw, int accumulate, hipStream_t s) {
    if (p.x_dtype == SS_DTYPE_F16) return launch_wgrad_c1_x3h_t<MODE, _Float16>(p, dw, accumulate, s);
    if (p.x_dtype == SS_DTYPE_BF16) return launch_wgrad_c1_x3h_t<MODE, __bf16>(p, dw, accumulate, s);
    return launch_wgrad_c1_x3h_t<MODE, float>(p, dw, accumulate, s);
}

}  // namespace

// Cout == 1, stride 1, full 7x7 / 4x4 / 3x3 tap box, Cin % 4 == 0 and 16-byte aligned pixels
bool ss_conv_out1_ok(const GConvParams& p) {
    C1Box box;
    if (p.Cout != 1 || !plain_grid(p) || p.Cin % 4 != 0 || p.Cin > 128 ||     // wide inputs: the 1x1 GEMM + tap sum path is better
        p.in_cs % 4 != 0 || (((uintptr_t)p.in) & 15) != 0) return false;
    if ((long)p.N * p.OH * p.OW < 16384 || !tap_box(p, &box)) return false;
    return (box.kh == 7 && box.kw == 7) || (box.kh == 4 && box.kw == 4) || (box.kh == 3 && box.kw == 3);
}

namespace {
bool out1_x3h_shape(const GConvParams& p, const C1Box& box) {
    return ss_tuning().c1_mfma && (p.Cin == 32 || p.Cin == 64) && box.kh <= 8 && box.kw <= 8 && p.dtype == SS_DTYPE_F32;
}
template <int CQ, typename TI>
int launch_out1_x3h_t(const GConvParams& p, const C1Box& box, hipStream_t s) {
    const size_t smem = (size_t)(2 * 64 * O1_ZS + O1_TH * 64 + 64 + 4) * sizeof(float);
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)conv_out1_x3h_kernel<CQ, TI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const int TW = 64 - (box.kw - 1);
    const int tiles_y = (p.OH + O1_TH - 1) / O1_TH, tiles_x = (p.OW + TW - 1) / TW;
    const int ntiles = p.N * tiles_y * tiles_x;
    hipLaunchKernelGGL((conv_out1_x3h_kernel<CQ, TI>), dim3(ntiles < 511 ? ntiles : 511), dim3(256), smem, s, p, box, ntiles, tiles_x, tiles_y);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <int CQ>
int launch_out1_x3h(const GConvParams& p, const C1Box& box, hipStream_t s) {
    if (p.c1_dtype == SS_DTYPE_F16) return launch_out1_x3h_t<CQ, _Float16>(p, box, s);
    if (p.c1_dtype == SS_DTYPE_BF16) return launch_out1_x3h_t<CQ, __bf16>(p, box, s);
    return launch_out1_x3h_t<CQ, float>(p, box, s);
}
}  // namespace

// the matrix-core kernel takes the problem: the only one that reads a 16-bit stored input (GConvParams::c1_dtype)
bool ss_conv_out1_typed_ok(const GConvParams& p) {
    C1Box box;
    return ss_conv_out1_ok(p) && tap_box(p, &box) && out1_x3h_shape(p, box);
}

int ss_launch_conv_out1(const GConvParams& p, hipStream_t s) {
    C1Box box;
    if (!ss_conv_out1_ok(p) || !tap_box(p, &box)) return SS_ERR_UNSUPPORTED;
    if (out1_x3h_shape(p, box)) return p.Cin == 64 ? launch_out1_x3h<4>(p, box, s) : launch_out1_x3h<2>(p, box, s);
    if (p.c1_dtype != SS_DTYPE_F32) { ss_set_error("conv_out1: only the matrix-core kernel reads a 16-bit stored input (ss_conv_out1_typed_ok)"); return SS_ERR_UNSUPPORTED; }
    if (box.kh == 7) return launch_out1<7, 7>(p, box, s);
    if (box.kh == 4) return launch_out1<4, 4>(p, box, s);
    return launch_out1<3, 3>(p, box, s);
}

// ... on the matrix cores: >= 32 output channels in multiples of 4, any full tap box up to 8 x 8
namespace {
bool in1_x3h_shape(const GConvParams& p, C1Box* box) {
    if (!ss_tuning().c1_mfma || p.Cin != 1 || (p.in_s != 1 && p.in_s != 2) || p.out_s != 1 || p.nbatch > 1 || p.Cout < 32 || p.Cout % 4 != 0 || p.out_cs % 4 != 0 ||
        (((uintptr_t)p.out) & 15) != 0 || p.dtype != SS_DTYPE_F32) return false;
    if (!tap_box(p, box) || box->kh > 8 || box->kw > 8) return false;
    return true;
}
int launch_in1_x3h(const GConvParams& p, const C1Box& box, const C1Fold& f, hipStream_t s) {
    const int OH = f.on ? f.XH : p.OH, OW = f.on ? f.XW : p.OW;
    const int tiles_y = (OH + I1_TH - 1) / I1_TH, tiles_x = (OW + I1_TW - 1) / I1_TW;
    const int ntiles = p.N * tiles_y * tiles_x;
    const int gy = (p.Cout + 63) / 64;
    const dim3 grid(ntiles < 511 ? ntiles : 511, gy);          // odd: the tiles of one image column (the border columns are slower) spread over all workgroups
    const int dbg = ss_tuning().tile_dbg;
    auto go = [&](auto tc) {
        typedef decltype(tc) TO;
        if (f.on) hipLaunchKernelGGL((conv_in1_x3h_kernel<true, 1, TO>), grid, dim3(256), 0, s, p, box, f, ntiles, tiles_x, tiles_y, dbg);
        else if (p.in_s == 2) hipLaunchKernelGGL((conv_in1_x3h_kernel<false, 2, TO>), grid, dim3(256), 0, s, p, box, f, ntiles, tiles_x, tiles_y, dbg);
        else hipLaunchKernelGGL((conv_in1_x3h_kernel<false, 1, TO>), grid, dim3(256), 0, s, p, box, f, ntiles, tiles_x, tiles_y, dbg);
    };
    if (p.c1_dtype == SS_DTYPE_F16) go((_Float16)0);
    else if (p.c1_dtype == SS_DTYPE_BF16) go((__bf16)0);
    else go(0.f);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
}  // namespace

// data gradient of a reflection-padded Cout == 1 layer written straight into dx (p: the zero-padded problem on the PADDED grid,
// p.in = dy, taps = -(a, b); p.out / out_cs / accumulate describe dx, whose grid is ih x iw)
bool ss_conv_in1_fold_ok(const GConvParams& p, int pt, int pl, int ih, int iw) {
    C1Box box;
    return p.reflect == 0 && !p.bias && p.act == SS_ACT_NONE && (long)p.N * ih * iw >= 16384 && ih >= 2 * pt + 4 && iw >= 2 * pl + 4 && pt >= 0 && pl >= 0 && in1_x3h_shape(p, &box) &&
           pt < box.kh && pl < box.kw && (long)p.N * ih * iw * p.out_cs < (1L << 31);
}
int ss_launch_conv_in1_fold(const GConvParams& p, int pt, int pl, int ih, int iw, hipStream_t s) {
    C1Box box;
    if (!ss_conv_in1_fold_ok(p, pt, pl, ih, iw) || !tap_box(p, &box)) return SS_ERR_UNSUPPORTED;
    return launch_in1_x3h(p, box, C1Fold{1, pt, pl, ih, iw}, s);
}

// output statistics (GConvParams::stats): the matrix-core kernel on a plain problem, one chunk per 8 x 64 tile
int ss_conv_in1_stats_chunks(const GConvParams& p) {
    C1Box box;
    if (!ss_conv_in1_ok(p) || !in1_x3h_shape(p, &box) || p.act != SS_ACT_NONE || p.accumulate) return 0;
    return ((p.OH + I1_TH - 1) / I1_TH) * ((p.OW + I1_TW - 1) / I1_TW);
}

// Cin == 1, stride 1, full tap box, Cout % 16 == 0, weights with the output channel contiguous (ldb irrelevant for one input channel)
bool ss_conv_in1_ok(const GConvParams& p) {
    C1Box box;
    if (p.Cin == 1 && p.in_s == 2) {          // stride 2 (the discriminators' stem): the matrix-core kernel only
        GConvParams q = p;
        q.in_s = 1;
        return plain_grid(q) && (long)p.N * p.OH * p.OW >= 16384 && (long)p.N * p.IH * p.IW * p.in_cs < (1L << 31) && in1_x3h_shape(p, &box) && box.kh <= 8 && box.kw <= 8;
    }
    if (p.Cin != 1 || !plain_grid(p) || p.Cout % 16 != 0 || p.out_cs % 4 != 0 || (((uintptr_t)p.out) & 15) != 0) return false;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return false;
    if ((long)p.N * p.OH * p.OW < 16384 || !tap_box(p, &box)) return false;
    return (box.kh == 7 && box.kw == 7) || (box.kh == 4 && box.kw == 4) || (box.kh == 3 && box.kw == 3);
}

// the matrix-core kernel takes the problem: the only one that writes a 16-bit stored output (GConvParams::c1_dtype)
bool ss_conv_in1_typed_ok(const GConvParams& p) {
    C1Box box;
    return ss_conv_in1_ok(p) && in1_x3h_shape(p, &box);
}

int ss_launch_conv_in1(const GConvParams& p, hipStream_t s) {
    C1Box box;
    if (!ss_conv_in1_ok(p) || !tap_box(p, &box)) return SS_ERR_UNSUPPORTED;
    if (in1_x3h_shape(p, &box)) return launch_in1_x3h(p, box, C1Fold{0, 0, 0, 0, 0}, s);
    if (p.c1_dtype != SS_DTYPE_F32) { ss_set_error("conv_in1: only the matrix-core kernel writes a 16-bit stored output (ss_conv_in1_typed_ok)"); return SS_ERR_UNSUPPORTED; }
    if (p.stats) { ss_set_error("conv_in1: output statistics requested on the VALU path"); return SS_ERR_UNSUPPORTED; }
    if (box.kh == 7) return launch_in1<7, 7>(p, box, s);
    if (box.kh == 4) return launch_in1<4, 4>(p, box, s);
    return launch_in1<3, 3>(p, box, s);
}

// ---- weight gradient, one channel on one side (stride 1, kh * kw <= 64 taps) ----------------------------------------------------
// mode 0: Cout == 1 (X = conv input x with C = Cin channels, S = dy);  mode 1: Cin == 1 (X = dy with C = Cout channels, S = x)
bool ss_wgrad_c1_ok(int n, int xh, int xw, int C, int kh, int kw) {
    return ss_tuning().wgrad_c1 && C >= 32 && C <= 128 && kh >= 1 && kw >= 1 && kh * kw <= C1W_T && kh <= 8 && kw <= 8 && (long)n * xh * xw >= 65536 &&
           (long)n * xh * xw * C < (1L << 31);
}

size_t ss_wgrad_c1_ws(int n, int xh, int xw, int C, int kh, int kw) {
    const long ntiles = (long)n * ((xh + XW_TH - 1) / XW_TH) * ((xw + XW_TW - 1) / XW_TW);          // the smaller tiles of the two kernels: more partials
    return ss_align_up((size_t)c1w_blocks(ntiles) * kh * kw * C * sizeof(float), 256);
}

// the matrix-core kernel takes the problem: the only one that reads a 16-bit stored X
bool ss_wgrad_c1_typed_ok(const void* X, int X_cs, int C, int xh, int xw, int kh, int kw, int pt, int pl) {
    return ss_tuning().c1_mfma && C % 4 == 0 && X_cs % 4 == 0 && (((uintptr_t)X) & 15) == 0 && (XW_TH + kh - 1) * (XW_TW + kw - 1) <= 768 &&
           xh >= 2 * pt + 4 && xw >= 2 * pl + 4;
}

int ss_launch_wgrad_c1(int mode, const float* X, int X_cs, int C, int n, int xh, int xw, const float* S, int S_cs, int sh, int sw,
                       int kh, int kw, int pt, int pl, int reflect, float* dw, int accumulate, void* ws, hipStream_t s, int x_dtype) {
    C1WParams p{};
    p.X = X; p.S = S; p.part = (float*)ws;
    p.N = n; p.XH = xh; p.XW = xw; p.C = C; p.X_cs = X_cs;
    p.SH = sh; p.SW = sw; p.S_cs = S_cs;
    p.kh = kh; p.kw = kw; p.pt = pt; p.pl = pl; p.reflect = reflect;
    p.x_dtype = x_dtype;
    if (x_dtype != SS_DTYPE_F32 && !ss_wgrad_c1_typed_ok(X, X_cs, C, xh, xw, kh, kw, pt, pl)) {
        ss_set_error("wgrad_c1: only the matrix-core kernel reads a 16-bit stored X (ss_wgrad_c1_typed_ok)");
        return SS_ERR_UNSUPPORTED;
    }
    if (ss_wgrad_c1_typed_ok(X, X_cs, C, xh, xw, kh, kw, pt, pl)) {
        p.dbg = ss_tuning().tile_dbg;
        p.tiles_y = (xh + XW_TH - 1) / XW_TH;
        p.tiles_x = (xw + XW_TW - 1) / XW_TW;
        p.ntiles = n * p.tiles_y * p.tiles_x;
        return mode == 0 ? launch_wgrad_c1_x3h<0>(p, dw, accumulate, s) : launch_wgrad_c1_x3h<1>(p, dw, accumulate, s);
    }
    p.tiles_y = (xh + C1W_TH - 1) / C1W_TH;
    p.tiles_x = (xw + C1W_TW - 1) / C1W_TW;
    p.ntiles = n * p.tiles_y * p.tiles_x;
    return mode == 0 ? launch_wgrad_c1<0>(p, dw, accumulate, s) : launch_wgrad_c1<1>(p, dw, accumulate, s);
}
