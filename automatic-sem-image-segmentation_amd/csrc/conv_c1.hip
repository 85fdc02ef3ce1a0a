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
    constexpr int HR = C1_TH + KH - 1, HW = C1_TW + KW - 1, HS = (HW + 3 + 3) / 4 * 4;    // row stride: covers the 12-float strip reads
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

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p.Cin; c0 += CC) {
        __syncthreads();
        // halo tile: HR x HW pixels x CC channels, two float4 per pixel
        for (int idx = tid; idx < HR * HW * (CC / 4); idx += 256) {
            const int c4 = idx % (CC / 4);
            const int pix = idx / (CC / 4);
            const int rx = pix % HW, ry = pix / HW;
            const int iy = ss_map_index(y0 + ry + p.in_oy + box.dy0, p.IH, p.reflect);
            const int ix = ss_map_index(x0 + rx + p.in_ox + box.dx0, p.IW, p.reflect);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && ix >= 0 && c0 + 4 * c4 < p.Cin)
                v = *(const f32x4*)(p.in + ((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs + c0 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) xs[((4 * c4 + e) * HR + ry) * HS + rx] = v[e];
        }
        for (int idx = tid; idx < CC * KH * 8; idx += 256) ws[idx] = 0.f;
        __syncthreads();
        for (int idx = tid; idx < p.ntaps * CC; idx += 256) {
            const int c = idx % CC, t = idx / CC;
            const int ry = p.taps[t].dy - box.dy0, rx = p.taps[t].dx - box.dx0;
            if (c0 + c < p.Cin) ws[(c * KH + ry) * 8 + rx] = p.w[p.taps[t].woff + (long)(c0 + c) * p.ldb];
        }
        __syncthreads();
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
    constexpr int CC = 8, HR = C1_TH + KH - 1, HW = C1_TW + KW - 1, HS = (HW + 3 + 3) / 4 * 4;
    const size_t smem = (size_t)(CC * HR * HS + CC * KH * 8) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_out1_kernel<KH, KW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
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
