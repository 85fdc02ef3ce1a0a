// Stride-1 convolutions with ONE channel on one side, at full tile resolution -- the generator's 7x7 stem (1 -> F) and
// 7x7 head (F -> 1) and their data gradients.  They carry 0.8 % of the step's FLOPs but are HBM-bound layers (the 64-channel
// side is 0.5 GB at 512x512, batch 8) that map badly onto an implicit GEMM (K = 49 or N = 1), so they get LDS-tiled VALU
// kernels: the halo tile of the input is staged once in LDS, every thread keeps a strip of outputs in registers.
//
//   out1 : out[p]    = act(bias + sum_{t,c} in[map(p + d_t)][c] * w[t][c])          (Cout == 1)
//   in1  : out[p][c] = act(bias[c] + sum_t in[map(p + d_t)] * w[t][c])              (Cin == 1)
// Both take the generic GConvParams (taps = a full KH x KW box, in_s == out_s == 1, class grid == output grid).
#include "common.h"

namespace {

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

}  // namespace

// Cout == 1, stride 1, full 7x7 / 4x4 / 3x3 tap box, Cin % 4 == 0 and 16-byte aligned pixels
bool ss_conv_out1_ok(const GConvParams& p) {
    C1Box box;
    if (p.Cout != 1 || !plain_grid(p) || p.Cin % 4 != 0 || p.Cin > 128 ||     // wide inputs: the 1x1 GEMM + tap sum path is better
        p.in_cs % 4 != 0 || (((uintptr_t)p.in) & 15) != 0) return false;
    if ((long)p.N * p.OH * p.OW < 16384 || !tap_box(p, &box)) return false;
    return (box.kh == 7 && box.kw == 7) || (box.kh == 4 && box.kw == 4) || (box.kh == 3 && box.kw == 3);
}

int ss_launch_conv_out1(const GConvParams& p, hipStream_t s) {
    C1Box box;
    if (!ss_conv_out1_ok(p) || !tap_box(p, &box)) return SS_ERR_UNSUPPORTED;
    if (box.kh == 7) return launch_out1<7, 7>(p, box, s);
    if (box.kh == 4) return launch_out1<4, 4>(p, box, s);
    return launch_out1<3, 3>(p, box, s);
}

// Cin == 1, stride 1, full tap box, Cout % 16 == 0, weights with the output channel contiguous (ldb irrelevant for one input channel)
bool ss_conv_in1_ok(const GConvParams& p) {
    C1Box box;
    if (p.Cin != 1 || !plain_grid(p) || p.Cout % 16 != 0 || p.out_cs % 4 != 0 || (((uintptr_t)p.out) & 15) != 0) return false;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return false;
    if ((long)p.N * p.OH * p.OW < 16384 || !tap_box(p, &box)) return false;
    return (box.kh == 7 && box.kw == 7) || (box.kh == 4 && box.kw == 4) || (box.kh == 3 && box.kw == 3);
}

int ss_launch_conv_in1(const GConvParams& p, hipStream_t s) {
    C1Box box;
    if (!ss_conv_in1_ok(p) || !tap_box(p, &box)) return SS_ERR_UNSUPPORTED;
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
    const long ntiles = (long)n * ((xh + C1W_TH - 1) / C1W_TH) * ((xw + C1W_TW - 1) / C1W_TW);
    return ss_align_up((size_t)c1w_blocks(ntiles) * kh * kw * C * sizeof(float), 256);
}

int ss_launch_wgrad_c1(int mode, const float* X, int X_cs, int C, int n, int xh, int xw, const float* S, int S_cs, int sh, int sw,
                       int kh, int kw, int pt, int pl, int reflect, float* dw, int accumulate, void* ws, hipStream_t s) {
    C1WParams p{};
    p.X = X; p.S = S; p.part = (float*)ws;
    p.N = n; p.XH = xh; p.XW = xw; p.C = C; p.X_cs = X_cs;
    p.SH = sh; p.SW = sw; p.S_cs = S_cs;
    p.kh = kh; p.kw = kw; p.pt = pt; p.pl = pl; p.reflect = reflect;
    p.tiles_y = (xh + C1W_TH - 1) / C1W_TH;
    p.tiles_x = (xw + C1W_TW - 1) / C1W_TW;
    p.ntiles = n * p.tiles_y * p.tiles_x;
    return mode == 0 ? launch_wgrad_c1<0>(p, dw, accumulate, s) : launch_wgrad_c1<1>(p, dw, accumulate, s);
}
