// LDS-staged "tile" convolution kernels for the small-channel, high-resolution layers of the MultiResUNet
// (UNet_Segmentation.py:401-503: 512x512 and 256x256 maps with 1 .. 64 channels).
//
// Why: on these layers the implicit-GEMM kernels (conv_mfma.hip) re-gather every input element once per tap through L1/L2 in
// 16-byte pieces and pad a 32-wide N tile; they ran at 0.10 of the HBM roofline.  Here a workgroup owns a
// TH x 32 pixel tile: the input tile + halo is read from global memory ONCE, staged in LDS, and all taps' MFMA A operands are
// formed from LDS.  Workgroups are persistent (weights stay in LDS across tiles) and walk XCD-contiguous chunks of the tile space.
//
//  * tconv_kernel (forward and data gradient: any stride-1 GConvParams problem whose taps reach at most a few pixels):
//    fp32-grade arithmetic on the fp16 matrix cores, x * s = h + l (two fp16 pieces), products hh + hl + lh in one fp32
//    accumulator, with ONE POWER-OF-TWO SCALE PER TILE taken from the tile's own maximum while it is staged (no pass over the
//    tensor for a global maximum; values within 2^-17 of their TILE's maximum keep 22 significand bits) and one scale for the
//    weight tensor; both are undone exactly in the epilogue.  v_mfma_f32_32x32x16_f16, M = 32 pixels of a tile row, N = output
//    channels (1 .. 4 blocks of 32), K = (tap, channel group of 8).
//  * twgrad_kernel (weight gradient of the same layers): x halo tile and dy tile staged in LDS as fp32, contraction over the
//    pixels with v_mfma_f32_32x32x2_f32 (exact fp32 products: the reduction runs over ~2 M pixels of heavy-tailed dy), rows
//    (tap, ci) packed densely, every workgroup accumulates ALL its tiles in registers and writes one partial; the fixed-order
//    reduction over workgroups (wgrad_reduce, conv_mfma.hip) keeps the result bit-reproducible.
#include "common.h"

#include <stdio.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 32;                 // tile width = one 32-row MFMA block per tile row
constexpr int T_THREADS = 256;         // 4 waves
constexpr int MAXQ = 96;               // k-groups (8 channels of one tap) per problem: 9 taps x 64 channels = 72 (+ padding)

struct TileGeom {
    int th;                 // tile rows (4 or 8)
    int cg;                 // channel groups of 8 per tap (CinPad / 8)
    int cgp;                // K steps (16 channels = two groups) per tap: (cg + 1) / 2; the odd group of the last step has zero weights
    int nq;                 // k-groups in the weight planes: ntaps * cgp * 2
    int nb;                 // 32-wide output-channel blocks
    unsigned m_c4n, m_cg;   // ceil(2^32 / d) for d = 2*cg (float4 slots per staged pixel) and d = cg: x / d == umulhi(x, m) for x*d < 2^32
    int wgs_per_cu;
    int hy0, hx0;           // smallest tap offset (in_oy + dy, in_ox + dx)
    int hh, hw;             // halo tile extents (pixels)
    int psb;                // bytes per staged pixel: cg * 32 (+16 so that it is an odd multiple of 16 -> conflict-free b128 reads)
    int ksb;                // bytes per weight row in LDS: nq * 16 (+16, same rule)
    size_t in_bytes, w_bytes, smem;
};

inline int odd16(int bytes) { return ((bytes / 16) % 2 == 0) ? bytes + 16 : bytes; }

bool tile_geom(const GConvParams& p, TileGeom* g) {
    if (p.ntaps < 1 || p.nbatch > 1 || p.in_s != 1 || p.out_s != 1 || p.out_oy || p.out_ox || p.OHc != p.OH || p.OWc != p.OW) return false;
    int y0 = 1 << 20, y1 = -(1 << 20), x0 = 1 << 20, x1 = -(1 << 20);
    for (int t = 0; t < p.ntaps; ++t) {
        const int oy = p.in_oy + p.taps[t].dy, ox = p.in_ox + p.taps[t].dx;
        y0 = oy < y0 ? oy : y0; y1 = oy > y1 ? oy : y1; x0 = ox < x0 ? ox : x0; x1 = ox > x1 ? ox : x1;
    }
    if (y1 - y0 > 2 || x1 - x0 > 2) return false;               // 1x1 .. 3x3 footprints
    g->cg = (p.Cin + 7) / 8;
    g->cgp = (g->cg + 1) / 2;
    g->nq = p.ntaps * g->cgp * 2;
    g->nb = (p.Cout + 31) / 32;
    g->m_c4n = (unsigned)((0x100000000ULL + 2 * g->cg - 1) / (2 * g->cg));
    g->m_cg = (unsigned)((0x100000000ULL + g->cg - 1) / g->cg);
    if (g->nq > MAXQ || g->nb > 4) return false;
    g->hy0 = y0; g->hx0 = x0;
    g->hw = TW + (x1 - x0);
    g->psb = odd16(g->cg * 32);
    g->ksb = odd16(g->nq * 16);
    g->w_bytes = (size_t)2 * g->nb * 32 * g->ksb;
    // tile height: 8 rows (2 MFMA row blocks per wave: weight fragments and the halo are amortised better) when two workgroups
    // of it fit a CU, unless 4 rows allow twice the resident workgroups (the phases of a tile -- load, convert, contract, store --
    // are serial inside a workgroup: co-resident workgroups are what overlaps them); "tile_th" overrides for measurements
    const int force = ss_tuning().tile_th;
    for (int th = 8; th >= 4; th -= 4) {
        g->th = th;
        g->hh = th + (y1 - y0);
        g->in_bytes = (size_t)g->hh * g->hw * g->psb;
        g->smem = g->in_bytes + g->w_bytes + 64;
        int per = (int)((160 * 1024) / (g->smem + 1024));
        const int cap = th == 8 ? 2 : 4;                        // VGPR budget: 202 / 128 registers
        g->wgs_per_cu = per > cap ? cap : per;
        if (force == th && g->wgs_per_cu >= 1) return true;
        if (force == 0 && th == 8 && g->wgs_per_cu >= 2) {
            TileGeom h = *g;
            h.th = 4; h.hh = 4 + (y1 - y0);
            h.in_bytes = (size_t)h.hh * h.hw * h.psb;
            h.smem = h.in_bytes + h.w_bytes + 64;
            const int per4 = (int)((160 * 1024) / (h.smem + 1024));
            if (per4 < 4) return true;                          // 4-row tiles would not reach 4 workgroups per CU: keep 8 rows
        }
    }
    return g->wgs_per_cu >= 1;                                  // th = 4
}

// ---- weights -> two fp16 planes [plane][nb*32 rows (co)][nq*8 (k = (tap, cg, 8 channels))], one power-of-two scale for the tensor
// ws layout: int e_w at byte 0, planes from byte 256.  One workgroup (the tensors have at most a few 10^4 elements).
__global__ __launch_bounds__(256) void tconv_wprep_kernel(GConvParams p, int cg, int cgp, int nq, int nb, unsigned char* __restrict__ ws) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int total = p.ntaps * p.Cin * p.Cout;
    float m = 0.f;
    for (int e = tid; e < total; e += 256) {
        const int co = e % p.Cout, r = e / p.Cout, ci = r % p.Cin, t = r / p.Cin;
        m = fmaxf(m, fabsf(p.w[p.taps[t].woff + (long)ci * p.ldb + co]));
    }
    red[tid] = m;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
        __syncthreads();
    }
    const int ew = ss_amax_exp(red[0]);
    if (tid == 0) *(int*)ws = ew;
    const float sw = ldexpf(1.f, 14 - ew);
    unsigned short* ph = (unsigned short*)(ws + 256);
    const int K = nq * 8, rows = nb * 32;
    unsigned short* pl = ph + (long)rows * K;
    for (int e = tid; e < rows * K; e += 256) {
        const int k = e % K, n = e / K;
        const int q = k / 8, j = k % 8, t = q / (2 * cgp), c = q % (2 * cgp), ci = c * 8 + j;      // k = ((t * cgp + step) * 2 + half) * 8 + j
        float v = 0.f;
        if (t < p.ntaps && c < cg && ci < p.Cin && n < p.Cout) v = p.w[p.taps[t].woff + (long)ci * p.ldb + n] * sw;
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)(v - (float)h);
        ph[e] = __builtin_bit_cast(unsigned short, h);
        pl[e] = __builtin_bit_cast(unsigned short, l);
    }
}

// XCD-contiguous persistent schedule: block -> (first tile, stride, end) inside its XCD's chunk of the tile space
__device__ __forceinline__ void tile_walk(int ntiles, int& first, int& stride, int& end) {
    const int nwg = gridDim.x, bid = blockIdx.x;            // nwg is a multiple of 8
    const int xcd = bid & 7, slot = bid >> 3, per = nwg >> 3;
    const int chunk = (ntiles + 7) / 8;
    first = xcd * chunk + slot;
    stride = per;
    end = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
}

template <int TM>      // tile rows per wave (tile height = 4 * TM)
__global__ __launch_bounds__(T_THREADS, (TM == 1 ? 4 : 2)) void tconv_kernel(GConvParams p, TileGeom g, const unsigned char* __restrict__ wprep) {
    constexpr int TH = 4 * TM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sIn = smem;                               // [hh*hw pixels][psb]: per channel group 16 B of h then 16 B of l
    unsigned char* sW = smem + g.in_bytes;                   // [2 planes][nb*32][ksb]
    float* red = (float*)(sW + g.w_bytes);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- once per workgroup: weights -> LDS ----
    const int ew = *(const int*)wprep;
    {
        const int rows = g.nb * 32, kb = g.nq * 16;          // bytes per row in the global planes
        const unsigned char* src = wprep + 256;
        const int per_row = kb / 16;
        for (int r = wave; r < 2 * rows; r += 4)             // r over [plane][row]
            for (int c16 = lane; c16 < per_row; c16 += 64)
                *(u32x4*)(sW + (long)r * g.ksb + c16 * 16) = *(const u32x4*)(src + (long)r * kb + c16 * 16);
    }

    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ntiles = p.N * tiles_y * tiles_x;
    int first, stride, end;
    tile_walk(ntiles, first, stride, end);
    const bool vec4 = (p.Cin % 4 == 0) && (p.in_cs % 4 == 0) && ((((uintptr_t)p.in) & 15) == 0);
    const int hp = g.hh * g.hw;                              // halo pixels
    const int c4n = g.cg * 2;                                // float4 slots per pixel (padded channels)
    const int row_slots = g.hw * c4n;

    for (int tile = first; tile < end; tile += stride) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- stage the halo tile as fp32 (in its final 32-byte units), tracking max |x|: a wave per halo row ----
        float vmax = 0.f;
        for (int hy = wave; hy < g.hh; hy += 4) {
            int iy = ss_map_index(oy0 + g.hy0 + hy, p.IH, p.reflect);
            if (iy >= p.IH) iy = -1;                         // far overhang of an edge tile under reflection: feeds no stored output
            const float* rowp = p.in + (long)(n * p.IH + (iy < 0 ? 0 : iy)) * p.IW * p.in_cs;
            unsigned char* drow = sIn + (long)hy * g.hw * g.psb;
            for (int e = lane; e < row_slots; e += 64) {
                const int hx = (int)__umulhi((unsigned)e, g.m_c4n), c4 = e - hx * c4n;
                int ix = ss_map_index(ox0 + g.hx0 + hx, p.IW, p.reflect);
                if (ix >= p.IW) ix = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int c = c4 * 4;
                if (iy >= 0 && ix >= 0 && c < p.Cin) {
                    const float* src = rowp + (long)ix * p.in_cs + c;
                    if (vec4) v = *(const f32x4*)src;
                    else {
                        v[0] = src[0];
                        if (c + 1 < p.Cin) v[1] = src[1];
                        if (c + 2 < p.Cin) v[2] = src[2];
                        if (c + 3 < p.Cin) v[3] = src[3];
                    }
                }
                vmax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), vmax);
                *(f32x4*)(drow + (long)hx * g.psb + c4 * 16) = v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int ex = ss_amax_exp(vmax);
        const float sx = ldexpf(1.f, 14 - ex);
        // ---- in place: every 32-byte unit (8 channels, fp32) -> 16 B of h + 16 B of l ----
        for (int e = tid; e < hp * g.cg; e += T_THREADS) {
            const int hpix = (int)__umulhi((unsigned)e, g.m_cg), c = e - hpix * g.cg;
            unsigned char* u = sIn + (long)hpix * g.psb + c * 32;
            const f32x4 a = *(const f32x4*)u, b = *(const f32x4*)(u + 16);
            f16x8 h, l;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float x0 = a[jj] * sx, x1 = b[jj] * sx;
                h[jj] = (_Float16)x0; h[4 + jj] = (_Float16)x1;
                l[jj] = (_Float16)(x0 - (float)h[jj]); l[4 + jj] = (_Float16)(x1 - (float)h[4 + jj]);
            }
            *(f16x8*)u = h;
            *(f16x8*)(u + 16) = l;
        }
        __syncthreads();

        // ---- contraction: wave w owns tile rows w*TM .. w*TM+TM-1, all output-channel blocks ----
        f32x16 acc[TM][4];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;
        const unsigned char* abase = sIn + (long)((wave * TM) * g.hw + l31) * g.psb;
        const unsigned char* bbase = sW + (long)l31 * g.ksb + lh * 16;
        const long wplane = (long)g.nb * 32 * g.ksb;
        const int rowb = g.hw * g.psb;
        // K steps = (tap, 16 channels): lanes 0..31 take the step's first channel group, lanes 32..63 its second (clamped to the
        // last group when cg is odd: its weight rows are zero)
        int step = 0;
        for (int t = 0; t < p.ntaps; ++t) {
            const int tapoff = ((p.in_oy + p.taps[t].dy - g.hy0) * g.hw + (p.in_ox + p.taps[t].dx - g.hx0)) * g.psb;
            for (int c2 = 0; c2 < g.cgp; ++c2, ++step) {
                int cgi = 2 * c2 + lh;
                cgi = cgi < g.cg ? cgi : g.cg - 1;
                const unsigned char* ap = abase + tapoff + cgi * 32;
                f16x8 ah[TM], al[TM];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    ah[mi] = *(const f16x8*)(ap + (long)mi * rowb);
                    al[mi] = *(const f16x8*)(ap + (long)mi * rowb + 16);
                }
                const unsigned char* bp0 = bbase + step * 32;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    if (nb < g.nb) {
                        const unsigned char* bp = bp0 + (long)nb * 32 * g.ksb;
                        const f16x8 bh = *(const f16x8*)bp, bl = *(const f16x8*)(bp + wplane);
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) {
                            acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh, acc[mi][nb], 0, 0, 0);
                            acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl, acc[mi][nb], 0, 0, 0);
                            acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh, acc[mi][nb], 0, 0, 0);
                        }
                    }
                }
            }
        }

        // ---- epilogue: undo the scales, bias, activation, store (lane = channel, 16 pixels of the row per lane) ----
        const float oscale = ldexpf(1.f, (ex - 14) + (ew - 14));
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int oy = oy0 + wave * TM + mi;
            if (oy >= p.OH) continue;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int co = nb * 32 + l31;
                if (nb >= g.nb || co >= p.Cout) continue;
                const float bv = p.bias ? p.bias[co] : 0.f;
                float* orow = p.out + ((long)(n * p.OH + oy) * p.OW + ox0 + 4 * lh) * p.out_cs + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dx = (r & 3) + 8 * (r >> 2);
                    if (ox0 + 4 * lh + dx >= p.OW) continue;
                    float* o = orow + (long)dx * p.out_cs;
                    float v = ss_apply_act(acc[mi][nb][r] * oscale + bv, p.act, p.alpha);
                    if (p.accumulate) v += *o;
                    *o = v;
                }
            }
        }
        __syncthreads();          // the next tile overwrites sIn
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// weight gradient: part[wg][(t,ca)][cb] = sum over the workgroup's tiles of a[pixel + tap t][ca] * b[pixel][cb]
struct WTileGeom {
    int th, mb, nb;          // tile rows; 32-row blocks of M = ntaps*Ca; 32-wide blocks of Cb
    int mw, kw;              // the 4 waves = mw groups over the row blocks x kw groups over the tile's pixel rows (mw * kw == 4)
    int hy0, hx0, hh, hw;
    int psa, psb;            // floats per staged pixel (a halo tile / b tile)
    unsigned m_a4, m_b4;     // ceil(2^32 / d) for d = psa / 4, psb / 4
    size_t a_bytes, b_bytes, smem;
    int mb_per_wave;
    int wgs_per_cu;
};

bool wtile_geom(const WGradParams& p, WTileGeom* g) {
    if (p.ntaps < 1 || p.nbatch > 1 || p.a_s != 1) return false;
    int y0 = 1 << 20, y1 = -(1 << 20), x0 = 1 << 20, x1 = -(1 << 20);
    for (int t = 0; t < p.ntaps; ++t) {
        const int oy = p.a_oy + p.taps[t].dy, ox = p.a_ox + p.taps[t].dx;
        y0 = oy < y0 ? oy : y0; y1 = oy > y1 ? oy : y1; x0 = ox < x0 ? ox : x0; x1 = ox > x1 ? ox : x1;
    }
    if (y1 - y0 > 2 || x1 - x0 > 2) return false;
    const int M = p.ntaps * p.Ca;
    g->mb = (M + 31) / 32;
    g->nb = (p.Cb + 31) / 32;
    g->mw = g->mb >= 3 ? 4 : g->mb;                  // 1, 2 or 4 wave groups over the row blocks
    g->kw = 4 / g->mw;                               // the others split the pixel rows (partial sums joined through LDS at the end)
    g->mb_per_wave = (g->mb + g->mw - 1) / g->mw;
    if (g->mb_per_wave * g->nb > 6 || g->nb > 2) return false;          // <= 96 accumulator registers per lane
    g->hy0 = y0; g->hx0 = x0; g->hw = TW + (x1 - x0);
    // pixel strides (floats), multiples of 4 (16-byte staging stores); a multiple of 32 would put every pixel on the same banks
    g->psa = (p.Ca + 3) / 4 * 4;
    if (g->psa % 32 == 0) g->psa += 4;
    g->psb = (p.Cb + 3) / 4 * 4;
    if (g->psb % 32 == 0) g->psb += 4;
    g->m_a4 = (unsigned)((0x100000000ULL + g->psa / 4 - 1) / (g->psa / 4));
    g->m_b4 = (unsigned)((0x100000000ULL + g->psb / 4 - 1) / (g->psb / 4));
    const size_t red_bytes = g->kw > 1 ? (size_t)T_THREADS * g->mb_per_wave * g->nb * 16 * 4 : 0;   // cross-wave join of the accumulators
    for (int th = 8; th >= 4; th -= 4) {
        g->th = th;
        g->hh = th + (y1 - y0);
        g->a_bytes = ss_align_up((size_t)g->hh * g->hw * g->psa * 4, 16);
        g->b_bytes = ss_align_up((size_t)th * TW * g->psb * 4, 16);
        g->smem = g->a_bytes + g->b_bytes;
        if (g->smem < red_bytes) g->smem = red_bytes;
        const int per = (int)((160 * 1024) / (g->smem + 1024));
        g->wgs_per_cu = per > 2 ? 2 : per;
        if (g->wgs_per_cu >= 2) return true;
    }
    return g->wgs_per_cu >= 1;
}

template <int MBW, int NB>      // 32-row blocks per wave, 32-wide column blocks
__global__ __launch_bounds__(T_THREADS, 2) void twgrad_kernel(WGradParams p, WTileGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sA = (float*)smem;                                // [hh*hw][psa]
    float* sB = (float*)(smem + g.a_bytes);                  // [th*32][psb]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int M = p.ntaps * p.Ca;
    const int wm = wave % g.mw, wk = wave / g.mw;            // row-block group, pixel-row group

    // per-lane A row -> offset of (tap, ca) relative to an output pixel's halo position
    int a_off[MBW];
    bool a_live[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int blk = wm + g.mw * i;
        const int m = blk * 32 + l31;
        a_live[i] = blk < g.mb && m < M;
        const int t = a_live[i] ? m / p.Ca : 0, ca = a_live[i] ? m - t * p.Ca : 0;
        a_off[i] = ((p.a_oy + p.taps[t].dy - g.hy0) * g.hw + (p.a_ox + p.taps[t].dx - g.hx0)) * g.psa + ca;
    }
    const bool b_live0 = l31 < p.Cb, b_live1 = 32 + l31 < p.Cb;

    f32x16 acc[MBW][NB];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int tiles_x = (p.GW + TW - 1) / TW, tiles_y = (p.GH + g.th - 1) / g.th;
    const int ntiles = p.N * tiles_y * tiles_x;
    int first, stride, end;
    tile_walk(ntiles, first, stride, end);
    const bool va = (p.Ca % 4 == 0) && (p.a_cs % 4 == 0) && ((((uintptr_t)p.a) & 15) == 0);
    const bool vb = (p.Cb % 4 == 0) && (p.b_cs % 4 == 0) && ((((uintptr_t)p.b) & 15) == 0);
    const int a4 = g.psa / 4, b4 = g.psb / 4;

    for (int tile = first; tile < end; tile += stride) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int gy0 = ty * g.th, gx0 = tx * TW;
        // stage: a wave per halo row of a / per tile row of b
        for (int hy = wave; hy < g.hh; hy += 4) {
            int iy = ss_map_index(gy0 + g.hy0 + hy, p.AH, p.reflect);
            if (iy >= p.AH) iy = -1;
            const float* rowp = p.a + (long)(n * p.AH + (iy < 0 ? 0 : iy)) * p.AW * p.a_cs;
            float* drow = sA + (long)hy * g.hw * g.psa;
            for (int e = lane; e < g.hw * a4; e += 64) {
                const int hx = (int)__umulhi((unsigned)e, g.m_a4), c = (e - hx * a4) * 4;
                int ix = ss_map_index(gx0 + g.hx0 + hx, p.AW, p.reflect);
                if (ix >= p.AW) ix = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && ix >= 0 && c < p.Ca) {
                    const float* src = rowp + (long)ix * p.a_cs + c;
                    if (va) v = *(const f32x4*)src;
                    else {
                        v[0] = src[0];
                        if (c + 1 < p.Ca) v[1] = src[1];
                        if (c + 2 < p.Ca) v[2] = src[2];
                        if (c + 3 < p.Ca) v[3] = src[3];
                    }
                }
                *(f32x4*)(drow + (long)hx * g.psa + c) = v;
            }
        }
        for (int py = wave; py < g.th; py += 4) {
            const int gy = gy0 + py;
            const float* rowp = p.b + (long)(n * p.GH + (gy < p.GH ? gy : 0)) * p.GW * p.b_cs;
            float* drow = sB + (long)py * TW * g.psb;
            for (int e = lane; e < TW * b4; e += 64) {
                const int px = (int)__umulhi((unsigned)e, g.m_b4), c = (e - px * b4) * 4;
                const int gx = gx0 + px;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy < p.GH && gx < p.GW && c < p.Cb) {          // pixels outside the grid contribute zero (b = 0)
                    const float* src = rowp + (long)gx * p.b_cs + c;
                    if (vb) v = *(const f32x4*)src;
                    else {
                        v[0] = src[0];
                        if (c + 1 < p.Cb) v[1] = src[1];
                        if (c + 2 < p.Cb) v[2] = src[2];
                        if (c + 3 < p.Cb) v[3] = src[3];
                    }
                }
                *(f32x4*)(drow + (long)px * g.psb + c) = v;
            }
        }
        __syncthreads();
        // K loop over pixel pairs (k = lh selects the pixel of the pair); this wave's share of the tile rows: py = wk, wk + kw, ...
        for (int py = wk; py < g.th; py += g.kw) {
            const float* arow = sA + (long)(py * g.hw + lh) * g.psa;
            const float* brow = sB + (long)(py * TW + lh) * g.psb + l31;
#pragma unroll 4
            for (int px = 0; px < TW; px += 2) {
                float av[MBW], bv[NB];
#pragma unroll
                for (int i = 0; i < MBW; ++i) av[i] = a_live[i] ? arow[(long)px * g.psa + a_off[i]] : 0.f;
                bv[0] = b_live0 ? brow[(long)px * g.psb] : 0.f;
                if (NB > 1) bv[NB - 1] = b_live1 ? brow[(long)px * g.psb + 32] : 0.f;
#pragma unroll
                for (int i = 0; i < MBW; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // join the kw pixel-row groups (fixed order: group 0 + group 1 (+ 2 + 3)), then one partial per workgroup
    if (g.kw > 1) {
        float* red = (float*)smem;                           // [wave][i][j][r][lane]
        constexpr int PER = MBW * NB * 16;
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((long)wave * PER + (i * NB + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r];
                        for (int k = 1; k < g.kw; ++k) v += red[((long)(wm + k * g.mw) * PER + (i * NB + j) * 16 + r) * 64 + lane];
                        acc[i][j][r] = v;
                    }
        }
    }
    if (wk != 0) return;
    float* part = p.part + (long)blockIdx.x * M * p.Cb;
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int blk = wm + g.mw * i;
        if (blk >= g.mb) continue;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int cb = j * 32 + l31;
            if (cb >= p.Cb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) part[(long)m * p.Cb + cb] = acc[i][j][r];
            }
        }
    }
}

template <int MBW, int NB>
int launch_twgrad(const WGradParams& p, const WTileGeom& g, int nwg, hipStream_t s) {
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)twgrad_kernel<MBW, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    char name[64];
    snprintf(name, sizeof(name), "twgrad_kernel<%d,%d>", MBW, NB);
    const double pix = (double)p.N * p.GH * p.GW;
    SsProfScope prof(name, 2.0 * p.ntaps * p.Ca * p.Cb * pix, 4.0 * pix * (p.Ca + p.Cb), s);
    hipLaunchKernelGGL((twgrad_kernel<MBW, NB>), dim3(nwg), dim3(T_THREADS), g.smem, s, p, g);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int tile_nwg(int ntiles, int wgs_per_cu) {
    const int cap = 256 * wgs_per_cu;                       // resident workgroups on 256 CUs
    int n = ntiles < cap ? ntiles : cap;
    n = (n + 7) / 8 * 8;
    return n;
}

}  // namespace

// ---- forward / data gradient ------------------------------------------------------------------------------------------------------
bool ss_tconv_ok(const GConvParams& p) {
    if (!ss_tuning().tile_conv || !ss_tuning().x6) return false;
    TileGeom g;
    if (!tile_geom(p, &g)) return false;
    // small channel counts on large maps: the layers whose gather (not the matrix pipe) bounds the implicit-GEMM kernels
    return p.Cin <= 64 && p.Cout <= 128 && (long)p.N * p.OH * p.OW >= 65536 && (long)p.N * p.IH * p.IW * p.in_cs < (1L << 31);
}

size_t ss_tconv_ws(const GConvParams& p) {
    TileGeom g;
    if (!tile_geom(p, &g)) return 0;
    return 256 + (size_t)2 * g.nb * 32 * g.nq * 16;
}

int ss_launch_tconv(const GConvParams& p, void* ws, size_t ws_bytes, hipStream_t s) {
    TileGeom g;
    if (!tile_geom(p, &g)) return SS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < ss_tconv_ws(p)) return SS_ERR_WORKSPACE;
    hipLaunchKernelGGL(tconv_wprep_kernel, dim3(1), dim3(256), 0, s, p, g.cg, g.cgp, g.nq, g.nb, (unsigned char*)ws);
    SS_LAUNCH_CHECK();
    const int tiles = p.N * ((p.OH + g.th - 1) / g.th) * ((p.OW + TW - 1) / TW);
    const int nwg = tile_nwg(tiles, g.wgs_per_cu);
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)tconv_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tconv_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const double pix = (double)p.N * p.OH * p.OW;
    SsProfScope prof(g.th == 8 ? "tconv_kernel<2> x3h" : "tconv_kernel<1> x3h", 2.0 * pix * p.Cout * p.ntaps * p.Cin * 3,
                     4.0 * pix * (p.Cin + p.Cout * (p.accumulate ? 2 : 1)), s);
    if (g.th == 8) hipLaunchKernelGGL(tconv_kernel<2>, dim3(nwg), dim3(T_THREADS), g.smem, s, p, g, (const unsigned char*)ws);
    else hipLaunchKernelGGL(tconv_kernel<1>, dim3(nwg), dim3(T_THREADS), g.smem, s, p, g, (const unsigned char*)ws);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// ---- weight gradient --------------------------------------------------------------------------------------------------------------
bool ss_twgrad_ok(const WGradParams& p) {
    if (!ss_tuning().tile_conv) return false;
    WTileGeom g;
    if (!wtile_geom(p, &g)) return false;
    return p.Ca <= 64 && p.Cb <= 64 && (long)p.N * p.GH * p.GW >= 65536 && (long)p.N * p.AH * p.AW * p.a_cs < (1L << 31) &&
           (long)p.N * p.GH * p.GW * p.b_cs < (1L << 31);
}

// number of partials (= workgroups) the launch will write: the caller sizes part[splits][M][Cb] and reduces over them
int ss_twgrad_splits(const WGradParams& p) {
    WTileGeom g;
    if (!wtile_geom(p, &g)) return 0;
    const int tiles = p.N * ((p.GH + g.th - 1) / g.th) * ((p.GW + TW - 1) / TW);
    return tile_nwg(tiles, g.wgs_per_cu);
}

int ss_launch_twgrad_partials(const WGradParams& p, hipStream_t s) {
    WTileGeom g;
    if (!wtile_geom(p, &g)) return SS_ERR_UNSUPPORTED;
    const int nwg = ss_twgrad_splits(p);
    if (p.splits != nwg) return SS_ERR_INVALID;
    const int k = g.mb_per_wave;
    if (g.nb == 1) {
        switch (k) {
            case 1: return launch_twgrad<1, 1>(p, g, nwg, s);
            case 2: return launch_twgrad<2, 1>(p, g, nwg, s);
            case 3: return launch_twgrad<3, 1>(p, g, nwg, s);
            case 4: return launch_twgrad<4, 1>(p, g, nwg, s);
            case 5: return launch_twgrad<5, 1>(p, g, nwg, s);
            case 6: return launch_twgrad<6, 1>(p, g, nwg, s);
        }
    } else {
        switch (k) {
            case 1: return launch_twgrad<1, 2>(p, g, nwg, s);
            case 2: return launch_twgrad<2, 2>(p, g, nwg, s);
            case 3: return launch_twgrad<3, 2>(p, g, nwg, s);
        }
    }
    return SS_ERR_UNSUPPORTED;
}
