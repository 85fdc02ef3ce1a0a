// Implicit-GEMM convolution on the bf16 matrix cores with fp32-exact operands ("x6"):
//   every fp32 value is carried as three bf16 pieces  v = h + m + l  (8+8+8 significant bits with signs: the split is
//   EXACT for finite normal values), and a product a*b is formed as
//        ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm          (six v_mfma_f32_32x32x16_bf16, fp32 accumulate)
//   Every bf16 x bf16 product is exact in fp32; the three dropped cross terms are <= 2^-26 |a*b|, below the rounding of a
//   single fp32 multiply.  The bf16 MFMA runs at 16x the fp32 MFMA rate, so six products still cost 0.375x.
//
// Same problem description as gconv_mfma (GConvParams): C[pixel][cout] = sum_{tap,ci} A_gather[pixel][(tap,ci)] * W[(tap,ci)][cout],
// requires Cin % 32 == 0 (one tap per K step) and float4-aligned activations.
//   A: fp32 activations gathered through the per-block LDS offset table (as the fp32 FAST path), split into the three
//      pieces in registers and stored as three bf16 planes in LDS  [plane][128 pixels][32 k (+8 pad)].
//   B: weights pre-split once per launch into K-contiguous bf16 planes [plane][batch][cout (padded to 128)][ntaps*Cin]
//      (wprep_x6_kernel below, or directly by the Winograd weight transform), staged global -> VGPR -> LDS as 16-byte rows.
// LDS rows are 80 bytes apart: the ds_read_b128 MFMA operand fetches (lane = row, 8 consecutive k) are conflict free.
// One LDS stage (61 KB at 128x128) so that two workgroups share a CU and cover each other's barriers.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // dword-aligned 16-byte load
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// the fp16 piece splits: common.h (ss_split_h2 / ss_split_h2s).  With the low piece carried at 2^11 times its value the cross
// products go to their OWN accumulator, folded in as 2^-11 * (h*l' + l'*h) at the end (wgrad_x6_kernel)
__device__ float ss_zero_page16[4] = {0.f, 0.f, 0.f, 0.f};           // what masked 16-byte loads read instead of selecting zeros afterwards

constexpr int XK = 32;          // K step (elements)
constexpr int XLD = XK + 8;     // LDS row stride in bf16 elements (80 bytes)
constexpr int XBM = 128;

// H = true ("x3h"): two fp16 pieces per operand instead of three bf16 ones,  x*s = h + l  with h = fp16(x*s), l = fp16(x*s - h) and
// ONE power-of-two scale s per operand TENSOR (p.h_amax: bit patterns of max|activation|, max|weight|, set by conv_api.hip), three
// products h*h + h*l + l*h in the same accumulator (the cross terms are 2^-11 smaller by themselves); values within 2^-17 of
// the tensor maximum keep 22 bits, smaller ones keep an absolute 2^-38 of the maximum.  The scales are undone in the epilogue.
typedef _Float16 xw_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 xw_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 xw_ld4(const float* q) { return *(const f32x4*)q; }
__device__ __forceinline__ f32x4 xw_ld4(const _Float16* q) { return __builtin_convertvector(*(const xw_f16x4*)q, f32x4); }
__device__ __forceinline__ f32x4 xw_ld4(const __bf16* q) { return __builtin_convertvector(*(const xw_bf16x4*)q, f32x4); }
// ... at ELEMENT alignment (gfx950 global memory takes multi-dword accesses at any dword, the 16-bit types at any 2-byte address): the
// 4-channel units of odd channel counts / odd pixel strides
typedef _Float16 xw_f16x4u __attribute__((ext_vector_type(4), aligned(2)));
typedef __bf16 xw_bf16x4u __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ f32x4 xw_ld4u(const float* q) { const f32x4u u = *(const f32x4u*)q; return f32x4{u[0], u[1], u[2], u[3]}; }
__device__ __forceinline__ f32x4 xw_ld4u(const _Float16* q) { return __builtin_convertvector((xw_f16x4)(*(const xw_f16x4u*)q), f32x4); }
__device__ __forceinline__ f32x4 xw_ld4u(const __bf16* q) { return __builtin_convertvector((xw_bf16x4)(*(const xw_bf16x4u*)q), f32x4); }
__device__ __forceinline__ void xw_st4(float* q, f32x4 v) { *(f32x4*)q = v; }
__device__ __forceinline__ void xw_st4(_Float16* q, f32x4 v) { *(xw_f16x4*)q = __builtin_convertvector(v, xw_f16x4); }
__device__ __forceinline__ void xw_st4(__bf16* q, f32x4 v) { *(xw_bf16x4*)q = __builtin_convertvector(v, xw_bf16x4); }

// T: activation storage type; NPROD: products per K step.  16-bit storage (H only, whole 32-channel chunks): the stored value (times a
// power of two) is the ONE fp16 operand plane of A; NPROD = 2 multiplies it with both weight pieces, 1 with the leading one
// (ss_tuning wino16_products), as conv_mfma_x6v2.hip.
template <int BM, int BN, bool H, typename T = float, int NPROD = 3>
__global__ __launch_bounds__(256, 2) void gconv_x6_kernel(GConvParams p, const unsigned short* __restrict__ bpl, long plane_elems,
                                                          int Npad, int Ktot, GPhases ph) {
    constexpr bool S16 = !std::is_same<T, float>::value;
    static_assert(!S16 || (H && NPROD <= 2), "16-bit storage: fp16 matrix cores, one or two products");
    constexpr int NP = H ? 2 : 3;        // operand planes
    typedef typename std::conditional<H, f16x8, bf16x8>::type FT;
    constexpr int TM = BM / 64;          // 32-row MFMA tiles per wave (2 waves along M)
    constexpr int TN = BN / 64;          // 32-wide MFMA column tiles per wave (2 waves along N)
    constexpr int BROWS = BN / 64;       // B loader: rows (tid>>2) + 64*j
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* sA = lds;                          // [NP][BM][XLD]
    unsigned short* sB = lds + NP * BM * XLD;          // [NP][BN][XLD]
    int* pixtab = (int*)(sB + NP * BN * XLD);
    int* offtab = pixtab + BM;                         // [BM][ntaps]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const long M = (long)p.N * p.OHc * p.OWc;
    const int gridN = (p.Cout + BN - 1) / BN;
    int tile;
    // per-phase fields (GPhases: several problems in one launch) or the plain problem's
    int P_ntaps = p.ntaps, P_oy = p.out_oy, P_ox = p.out_ox, prob = -1;
    {   // XCD-aware order (speed only): contiguous chunk of the tile space per XCD, N fastest
        int nwg = gridDim.x;
        const int bid = blockIdx.x;
        const int xcd = bid & 7;
        int slot = bid >> 3;
        if (ph.count > 1) {          // (the launcher makes the per-problem grid a multiple of 8)
            prob = slot % ph.count;
            slot /= ph.count;
            nwg /= ph.count;
            P_ntaps = ph.ntaps[prob]; P_oy = ph.out_oy[prob]; P_ox = ph.out_ox[prob];
            bpl = ph.planes[prob]; plane_elems = ph.plane_elems[prob]; Ktot = ph.Ktot[prob];
        }
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int gridM = (int)((M + BM - 1) / BM);
    const int batch = tile / (gridM * gridN);
    tile -= batch * gridM * gridN;
    const T* const g_in = (const T*)p.in + (long)batch * p.in_bs;
    T* const g_out = (T*)p.out + (long)batch * p.out_bs;
    const long m0 = (long)(tile / gridN) * BM;
    const int n0 = (tile % gridN) * BN;
    const int nchunks = Ktot / XK;
    const int Cq = Ktot / P_ntaps;       // reduction channels per tap, padded to a multiple of 32
    const bool ragged = Cq != p.Cin || (p.in_cs & 3) != 0 || (((uintptr_t)p.in) & (S16 ? 7 : 15)) != 0;
    float a_scale = 1.f, out_scale = 1.f;
    if constexpr (H) {
        const int ea = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax, p.amax_stripes))), ew = ss_amax_exp(__uint_as_float(p.h_amax2[0]));
        a_scale = ldexpf(1.f, 14 - ea);
        out_scale = ldexpf(1.f, ea - 14 + ew - 14);
    }

    {   // 32-bit pixel decode (ss_gconv_x6_ok: M < 2^31), one division pair per tile row; the offset table divides nothing
        int* rowc = offtab + BM * P_ntaps;      // [3][BM]
        if (tid < BM) {
            const unsigned m = (unsigned)m0 + (unsigned)tid;
            int v = -1, rn = -1, ry = 0, rx = 0;
            if (m < (unsigned)M) {
                const unsigned r = m / (unsigned)p.OWc;
                const int xc = (int)(m - r * (unsigned)p.OWc);
                const unsigned n = r / (unsigned)p.OHc;
                const int yc = (int)(r - n * (unsigned)p.OHc);
                const int oy = yc * p.out_s + P_oy, ox = xc * p.out_s + P_ox;
                if (oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW) v = ((int)n * p.OH + oy) * p.OW + ox;
                rn = (int)n;
                ry = yc * p.in_s + p.in_oy;
                rx = xc * p.in_s + p.in_ox;
            }
            pixtab[tid] = v;
            rowc[tid] = rn;
            rowc[BM + tid] = ry;
            rowc[2 * BM + tid] = rx;
        }
        __syncthreads();
        constexpr int TSTEP = 256 / BM;
        const int row = tid % BM;
        const int rn = rowc[row], ry = rowc[BM + row], rx = rowc[2 * BM + row];
        for (int t = tid / BM; t < P_ntaps; t += TSTEP) {
            int off = -1;
            if (rn >= 0) {
                const int iy = ss_map_index(ry + (prob >= 0 ? (int)ph.tdy[prob][t] : (int)p.taps[t].dy), p.IH, p.reflect);
                const int ix = ss_map_index(rx + (prob >= 0 ? (int)ph.tdx[prob][t] : (int)p.taps[t].dx), p.IW, p.reflect);
                if (iy >= 0 && ix >= 0) off = ((rn * p.IH + iy) * p.IW + ix) * p.in_cs;
            }
            offtab[row * P_ntaps + t] = off;
        }
    }
    __syncthreads();

    const int c4a = tid & 7, arow = tid >> 3;          // A loader: rows arow + 32*j, k = 4*c4a .. +3
    const int brow = tid >> 2, bpc = tid & 3;          // B loader: rows brow + 64*j, k = 8*bpc .. +7
    const unsigned short* bbase = bpl + ((long)batch * Npad + n0 + brow) * Ktot + bpc * 8;

    // two register sets: the global loads run TWO K steps ahead of the MFMAs (one step of 48 MFMAs per wave is shorter than
    // the L2 / HBM latency).  Zero-padding taps READ a 16-byte zero page (pointer select per load).
    constexpr int AU = BM / 32;          // A loader: rows arow + 32*j
    f32x4 ra[2][AU];
    u32x4 rb[2][NP][BROWS];

    auto load_tiles = [&](auto setc, int k0) {
        constexpr int S = decltype(setc)::value;
        const int t = k0 / Cq;                         // block-uniform
        const int ci0 = k0 - t * Cq;
        if (!ragged) {
            const T* abase = g_in + ci0 + c4a * 4;
#pragma unroll
            for (int j = 0; j < AU; ++j) {
                const int off = offtab[(arow + 32 * j) * P_ntaps + t];
                const T* pa = off < 0 ? (const T*)ss_zero_page16 : abase + off;      // masked taps read zeros
                ra[S][j] = xw_ld4(pa);
            }
        } else {
            // any channel count / pixel stride (the MultiResUNet's odd widths), any storage type: the reduction index runs over (tap, ci
            // padded to a multiple of 32 -- the weight planes hold zeros there); a 4-channel unit is fetched with ONE element-aligned
            // access (global_load_dwordx4 / dwordx2) when it lies fully inside the pixel's channels, element-wise when it straddles the end
            const int ci = ci0 + c4a * 4;
            const int nin = p.Cin - ci;                // channels of this unit that exist
#pragma unroll
            for (int j = 0; j < AU; ++j) {
                const int off = offtab[(arow + 32 * j) * P_ntaps + t];
                const bool ok = off >= 0 && nin > 0;
                const T* ptr = ok ? g_in + (off + ci) : (const T*)ss_zero_page16;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (nin >= 4) {
                    v = xw_ld4u(ptr);
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (e < nin) v[e] = (float)ptr[e];
                }
                ra[S][j] = v;
            }
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
                rb[S][pl][j] = *(const u32x4*)(bbase + pl * plane_elems + (long)(64 * j) * Ktot + k0);
    };
    auto store_tiles = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int j = 0; j < AU; ++j) {
            const f32x4 v = ra[S][j];
            unsigned short* dst = sA + (arow + 32 * j) * XLD + c4a * 4;
            if constexpr (H) {
                unsigned int hh[2], ll[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ss_split_h2(v[2 * e] * a_scale, v[2 * e + 1] * a_scale, hh[e], ll[e]);
                }
                *(u32x2*)(dst) = u32x2{hh[0], hh[1]};
                if (!S16) *(u32x2*)(dst + BM * XLD) = u32x2{ll[0], ll[1]};
            } else {
                unsigned int h[2], m[2], l[2];
                ss_split3x2(f32x2{v[0], v[1]}, h[0], m[0], l[0]);
                ss_split3x2(f32x2{v[2], v[3]}, h[1], m[1], l[1]);
                *(u32x2*)(dst) = u32x2{h[0], h[1]};
                *(u32x2*)(dst + BM * XLD) = u32x2{m[0], m[1]};
                *(u32x2*)(dst + 2 * BM * XLD) = u32x2{l[0], l[1]};
            }
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < BROWS; ++j)
                *(u32x4*)(sB + pl * BN * XLD + (brow + 64 * j) * XLD + bpc * 8) = rb[S][pl][j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_tiles(S0{}, 0);
    if (nchunks > 1) load_tiles(S1{}, XK);
    store_tiles(S0{});
    __syncthreads();

    const unsigned short* fa = sA + (wm * (BM / 2) + l31) * XLD + 8 * lh;
    const unsigned short* fb = sB + (wn * (BN / 2) + l31) * XLD + 8 * lh;

    // step c: chunk c is in LDS, chunk c+1 is in (or on its way to) register set (c+1)&1, chunk c+2 is requested into set c&1
    auto step = [&](int c, auto cur, auto nxt) {
        if (c + 2 < nchunks) load_tiles(cur, (c + 2) * XK);
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            FT a[NP][TM], b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                if (pl == 0 || !S16) {
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) a[pl][mi] = *(const FT*)(fa + pl * BM * XLD + mi * 32 * XLD + ks * 16);
                }
                if (pl == 0 || !S16 || NPROD >= 2) {
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) b[pl][ni] = *(const FT*)(fb + pl * BN * XLD + ni * 32 * XLD + ks * 16);
                }
            }
            // x6: six products, smallest terms first;  x3h: l*h, h*l, h*h (16-bit storage: h*l, h*h or h*h).  Consecutive MFMAs go to
            // different accumulators
            constexpr int NQ = H ? (S16 ? NPROD : 3) : 6;
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
            constexpr int HA[3] = {S16 ? 0 : 1, 0, 0}, HB[3] = {S16 ? (NPROD == 2 ? 1 : 0) : 0, S16 ? 0 : 1, 0};
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        if constexpr (H)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], acc[mi][ni], 0, 0, 0);
                        else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][mi], b[PB[q]][ni], acc[mi][ni], 0, 0, 0);
                    }
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            store_tiles(nxt);
            __syncthreads();
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, S0{}, S1{});
        if (c + 1 < nchunks) step(c + 1, S1{}, S0{});
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Pixel outer, column tile inner:
    // one table read and one offset product per output row
    int co[TN];
    float bv[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        co[ni] = n0 + wn * (BN / 2) + ni * 32 + l31;
        bv[ni] = (p.bias && co[ni] < p.Cout) ? p.bias[co[ni]] : 0.f;
    }
    // Coalesced epilogue (as gemm_x6p.hip / conv_mfma_x6v2.hip): every wave transposes its sub-tile through 2 KiB of the operand
    // tiles (free after the last __syncthreads above), 8 rows at a time, and stores 16 bytes per lane -- whole pixel rows of the wave's
    // columns per instruction instead of one 4-byte element per lane and row.  (The sub-pixel phases of the generators' last transposed
    // convolution, 64 outputs and K = 128 .. 512, are paced by their output stream.)
    {
        constexpr int COLS = 32 * TN, LPR = COLS / 4, RPI = 64 / LPR, NRD = 8 / RPI;
        const int cbase = n0 + wn * (BN / 2);
        const bool vec_ok = cbase + COLS <= p.Cout && (p.out_cs & 3) == 0 && (((uintptr_t)g_out) & 15) == 0;
        if (vec_ok) {
            float* tb = (float*)lds + wave * 512;          // [8 rows][COLS]
            const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *(const f32x4*)(p.bias + cbase + rcol);
            // ACC / PLAIN compile-time inside the store loop (see conv_mfma_x6v2.hip: a conditional load in it costs a vmcnt(0) per store)
            auto epi = [&](auto acc_c, auto plain_c) {
                constexpr bool ACC = decltype(acc_c)::value, PLAIN = decltype(plain_c)::value;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                            for (int ni = 0; ni < TN; ++ni)
                                tb[(rr + 4 * lh) * COLS + ni * 32 + l31] = H ? acc[mi][ni][rq * 4 + rr] * out_scale : acc[mi][ni][rq * 4 + rr];
                        __builtin_amdgcn_wave_barrier();          // one wave's LDS operations execute in issue order
#pragma unroll
                        for (int k = 0; k < NRD; ++k) {
                            const int row = rrow + RPI * k;
                            f32x4 v = *(const f32x4*)(tb + row * COLS + rcol);
                            const int pix = pixtab[wm * (BM / 2) + mi * 32 + 8 * rq + row];
                            if (pix >= 0) {
                                T* op = g_out + (long)pix * p.out_cs + cbase + rcol;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = PLAIN ? v[e] + b4[e] : ss_apply_act(v[e] + b4[e], p.act, p.alpha);
                                if (ACC) v += xw_ld4(op);
                                xw_st4(op, v);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            };
            if (p.accumulate) { if (p.act == SS_ACT_NONE) epi(std::true_type{}, std::true_type{}); else epi(std::true_type{}, std::false_type{}); }
            else { if (p.act == SS_ACT_NONE) epi(std::false_type{}, std::true_type{}); else epi(std::false_type{}, std::false_type{}); }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int pix = pixtab[wm * (BM / 2) + mi * 32 + row];
            if (pix < 0) continue;
            T* orow = g_out + (long)pix * p.out_cs;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                if (co[ni] >= p.Cout) continue;
                T* op = orow + co[ni];
                float v = ss_apply_act((H ? acc[mi][ni][r] * out_scale : acc[mi][ni][r]) + bv[ni], p.act, p.alpha);
                if (p.accumulate) v += (float)*op;
                *op = (T)v;
            }
        }
    }
}

// planes[pl][batch][n][t*Cin + ci] = piece pl of w[batch*w_bs + taps[t].woff + ci*ldb + n]; rows n in [Cout, Npad) are zero.
// 32x32 LDS transpose: coalesced reads along n, coalesced writes along k.
template <bool H>
__device__ __forceinline__ void wprep_x6_body(const GConvParams& p, unsigned short* __restrict__ planes, long plane_elems, int Npad, int Ktot,
                                              int bx, int by, int bz) {
    __shared__ float tl[32][33];
    const int batch = bz;
    const int k0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const float* w = p.w + (long)batch * p.w_bs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, n = n0 + tx;
        float v = 0.f;
        if (k < Ktot && n < p.Cout) {
            const int Cq = Ktot / p.ntaps;
            const int t = k / Cq, ci = k - t * Cq;
            if (ci < p.Cin) v = w[p.taps[t].woff + (long)ci * p.ldb + n];
        }
        tl[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        if (n < Npad && k < Ktot) {
            const long o = ((long)batch * Npad + n) * Ktot + k;
            if constexpr (H) {
                const int ew = ss_amax_exp(__uint_as_float(p.h_amax2[0]));
                const float x0 = tl[tx][ty + 8 * i] * ldexpf(1.f, 14 - ew);
                const _Float16 h0 = (_Float16)x0;
                const _Float16 l0 = (_Float16)(x0 - (float)h0);
                planes[o] = __builtin_bit_cast(unsigned short, h0);
                planes[o + plane_elems] = __builtin_bit_cast(unsigned short, l0);
            } else {
                unsigned int h, m, l;
                ss_split3x2(f32x2{tl[tx][ty + 8 * i], 0.f}, h, m, l);
                planes[o] = (unsigned short)(h & 0xffffu);
                planes[o + plane_elems] = (unsigned short)(m & 0xffffu);
                planes[o + 2 * plane_elems] = (unsigned short)(l & 0xffffu);
            }
        }
    }
}

template <bool H>
__global__ __launch_bounds__(256) void wprep_x6_kernel(GConvParams p, unsigned short* __restrict__ planes, long plane_elems, int Npad, int Ktot) {
    wprep_x6_body<H>(p, planes, plane_elems, Npad, Ktot, blockIdx.x, blockIdx.y, blockIdx.z);
}
// the split-plane jobs of a recorded plan in ONE launch (wprep_batch.hip): the problem descriptions are read from the plan in device memory
template <bool H>
__global__ __launch_bounds__(256) void wprep_x6_batch_kernel(const SsWJob* __restrict__ jobs, const int* __restrict__ map) {
    const SsWJob& j = jobs[map[blockIdx.x]];
    const int l = blockIdx.x - j.blk0;
    wprep_x6_body<H>(j.p, (unsigned short*)j.dst, j.n, j.a, j.b, l % j.gx, (l / j.gx) % j.gy, l / (j.gx * j.gy));
}

// Weight gradient with the same arithmetic:  part[split][(t,ca)][cb] = sum_pixels A_gather[pixel][(t,ca)] * B[pixel][cb].
// The reduction index is the PIXEL, which is the slow index of both NHWC operands, so the tiles are transposed on their way
// into LDS: a thread owns 4 consecutive pixels x 4 consecutive channels (four 16-byte global loads), splits them, and
// writes, per channel and piece, the 4 pixels as one 8-byte LDS store into the k-contiguous row of that channel.
// Lane -> (pixel group = tid & 7, channel quad = tid >> 3): a 16-lane store group covers two rows x 16 dwords = 32 banks.
// T: activation storage type.  16-bit storage (H only): both operands ARE stored 16-bit values -- one fp16 plane each (scaled by a
// power of two: exact), ONE product h*h per K step is the exact product of the stored values, accumulated in fp32.
template <int BN, bool H, typename T = float>
__global__ __launch_bounds__(256, 2) void wgrad_x6_kernel(WGradParams p) {
    constexpr bool S16 = !std::is_same<T, float>::value;          // single-plane operands
    static_assert(!S16 || H, "16-bit storage runs on the fp16 matrix cores");
    constexpr int BM = XBM;
    constexpr int NP = H ? 2 : 3;        // operand planes (H: fp16 two-piece split with one scale per operand tensor, see gconv_x6_kernel)
    typedef typename std::conditional<H, f16x8, bf16x8>::type FT;
    constexpr int TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short* sA = lds;                          // [NP][BM][XLD]
    unsigned short* sB = lds + NP * BM * XLD;          // [NP][BN][XLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int M = p.ntaps * p.Ca;
    // XCD-aware order.  The gx*gy output tiles of one (batch, split) slab read the SAME pixel range of both operands (each operand
    // tile gy resp. gx times); workgroups are dealt round-robin to the 8 XCDs (own L2 each), so in launch order a slab's tiles
    // land on all of them and every re-read goes to HBM (rocprofv3 FETCH_SIZE: 1.84 GB per trunk weight gradient against 0.30 GB of
    // operands -- the kernel ran at the HBM roofline, not the matrix pipe's).  Remap: linear id -> (xcd, q); a slab's tiles are
    // consecutive q on ONE XCD, whose L2 (4 MB) holds a slab's operand ranges.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int G = gridDim.x * gridDim.y, Z = gridDim.z, Zf = Z - (Z & 7);
        const long L = bx + (long)gridDim.x * (by + (long)gridDim.y * bz);
        if (L < (long)G * Zf) {
            const int xcd = (int)(L & 7);
            const long q = L >> 3;
            const int mn = (int)(q % G);
            bz = (int)(q / G) * 8 + xcd;
            bx = mn % gridDim.x;
            by = mn / gridDim.x;
        }
    }
    const int m0 = bx * BM;
    const int n0 = by * BN;
    const int split = bz % p.splits;
    const int batch = bz / p.splits;
    const T* const g_a = (const T*)p.a + (long)batch * p.a_bs;
    const T* const g_b = (const T*)p.b + (long)batch * p.b_bs;
    const long P = (long)p.N * p.GH * p.GW;
    const long ps = (long)split * p.pix_per_split;
    const long pe = (ps + p.pix_per_split < P) ? ps + p.pix_per_split : P;
    const int nchunks = (int)((pe - ps + XK - 1) / XK);
    float a_scale = 1.f, b_scale = 1.f, out_scale = 1.f;
    if constexpr (H) {
        const int ea = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax, p.amax_stripes))), eb = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax2, p.amax2_stripes)));
        a_scale = ldexpf(1.f, 14 - ea);
        b_scale = ldexpf(1.f, 14 - eb);
        out_scale = ldexpf(1.f, ea - 14 + eb - 14);
    }

    const int kq = tid & 7, cq = tid >> 3;
    // A: this thread's channel quad (fixed over the pixel loop)
    const int am = m0 + 4 * cq;
    const bool a_val = am < M;
    const int a_t = a_val ? am / p.Ca : 0;
    const int a_c = a_val ? am - a_t * p.Ca : 0;
    const int a_dy = p.taps[a_t].dy, a_dx = p.taps[a_t].dx;
    // B: channel quad n0 + 4*cq (BN = 64: only quads 0..15 exist)
    const int bn = n0 + 4 * cq;
    const bool b_val = (4 * cq < BN) && (bn + 4 <= p.Cb);

    // decoded coordinates of this thread's first pixel of the current K step
    int f_x, f_y, f_n;
    {
        const long pk = ps + 4 * kq;
        f_x = (int)(pk % p.GW);
        const long r = pk / p.GW;
        f_y = (int)(r % p.GH);
        f_n = (int)(r / p.GH);
    }

    // TWO register sets: the global loads run two K steps ahead of their split + LDS store (with one set a step was a serial chain of
    // load latency -> split -> barrier -> 24 MFMAs -> barrier: ~3 us per 32-pixel chunk whatever the operand bytes)
    f32x4 ra[2][4], rb[2][4];
    // Out-of-image / out-of-range pieces are READ from a 16-byte zero page (a pointer select per load instead of a value select per
    // element at store time).  Fast addressing (the usual case): the class grid's rows are multiples of 4 pixels and the split
    // starts on one, so a thread's 4 pixels lie in ONE row -- the row is mapped once, the columns step by the stride -- and both
    // tensors are below 2^31 elements, so the offsets are 32-bit until the final pointer add.
    const T* const zpage = (const T*)ss_zero_page16;
    const bool fast = (p.GW & 3) == 0 && (ps & 3) == 0 && (long)p.N * p.AH * p.AW * p.a_cs < (1L << 31) && P * p.b_cs < (1L << 31);
    auto load_tiles = [&](auto setc, long pk0) {
        constexpr int S = decltype(setc)::value;
        if (p.dbg & 1) {          // measurement: no global loads
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[S][i] = f32x4{1.f, 2.f, 3.f, 4.f}; rb[S][i] = f32x4{1.f, 1.f, 1.f, 1.f}; }
            return;
        }
        if (fast) {
            const long pk = pk0 + 4 * kq;
            const bool in = pk < pe;              // pe is a multiple of 4 here: all four pixels or none
            const int iy = ss_map_index(f_y * p.a_s + p.a_oy + a_dy, p.AH, p.reflect);
            const bool rowok = a_val && in && iy >= 0;
            const int rowbase = (f_n * p.AH + iy) * p.AW;
            const int xb = f_x * p.a_s + p.a_ox + a_dx;
            const int bo = (int)pk * p.b_cs + bn;
            const bool okb = b_val && in;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ix = ss_map_index(xb + i * p.a_s, p.AW, p.reflect);
                const T* pa = (rowok && ix >= 0) ? g_a + ((rowbase + ix) * p.a_cs + a_c) : zpage;
                ra[S][i] = xw_ld4(pa);
                const T* pb = okb ? g_b + (bo + i * p.b_cs) : zpage;
                rb[S][i] = xw_ld4(pb);
            }
        } else {
            int x = f_x, y = f_y, n = f_n;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long pk = pk0 + 4 * kq + i;
                const int iy = ss_map_index(y * p.a_s + p.a_oy + a_dy, p.AH, p.reflect);
                const int ix = ss_map_index(x * p.a_s + p.a_ox + a_dx, p.AW, p.reflect);
                const bool oka = a_val && pk < pe && iy >= 0 && ix >= 0;
                const T* pa = oka ? g_a + (((long)(n * p.AH + iy) * p.AW + ix) * p.a_cs + a_c) : zpage;
                ra[S][i] = xw_ld4(pa);
                const bool okb = b_val && pk < pe;
                const T* pb = okb ? g_b + (pk * p.b_cs + bn) : zpage;
                rb[S][i] = xw_ld4(pb);
                if (++x >= p.GW) { x = 0; if (++y >= p.GH) { y = 0; ++n; } }
            }
        }
        // advance the base pixel by one K step
        f_x += XK;
        while (f_x >= p.GW) { f_x -= p.GW; if (++f_y >= p.GH) { f_y = 0; ++f_n; } }
    };
    auto store_tiles = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        if (p.dbg & 2) return;          // measurement: no split, no LDS stores
        const f32x4 (&va)[4] = ra[S];
        const f32x4 (&vb)[4] = rb[S];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned short* dst = sA + (4 * cq + e) * XLD + 4 * kq;
            if constexpr (H) {
                unsigned int hh[2], ll[2];
                ss_split_h2s(va[0][e] * a_scale, va[1][e] * a_scale, hh[0], ll[0]);
                ss_split_h2s(va[2][e] * a_scale, va[3][e] * a_scale, hh[1], ll[1]);
                *(u32x2*)(dst) = u32x2{hh[0], hh[1]};
                if (!S16) *(u32x2*)(dst + BM * XLD) = u32x2{ll[0], ll[1]};
            } else {
                unsigned int h0, m0_, l0, h1, m1, l1;
                ss_split3x2(f32x2{va[0][e], va[1][e]}, h0, m0_, l0);
                ss_split3x2(f32x2{va[2][e], va[3][e]}, h1, m1, l1);
                *(u32x2*)(dst) = u32x2{h0, h1};
                *(u32x2*)(dst + BM * XLD) = u32x2{m0_, m1};
                *(u32x2*)(dst + 2 * BM * XLD) = u32x2{l0, l1};
            }
        }
        if (4 * cq < BN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned short* dst = sB + (4 * cq + e) * XLD + 4 * kq;
                if constexpr (H) {
                    unsigned int hh[2], ll[2];
                    ss_split_h2s(vb[0][e] * b_scale, vb[1][e] * b_scale, hh[0], ll[0]);
                    ss_split_h2s(vb[2][e] * b_scale, vb[3][e] * b_scale, hh[1], ll[1]);
                    *(u32x2*)(dst) = u32x2{hh[0], hh[1]};
                    if (!S16) *(u32x2*)(dst + BN * XLD) = u32x2{ll[0], ll[1]};
                } else {
                    unsigned int h0, m0_, l0, h1, m1, l1;
                    ss_split3x2(f32x2{vb[0][e], vb[1][e]}, h0, m0_, l0);
                    ss_split3x2(f32x2{vb[2][e], vb[3][e]}, h1, m1, l1);
                    *(u32x2*)(dst) = u32x2{h0, h1};
                    *(u32x2*)(dst + BN * XLD) = u32x2{m0_, m1};
                    *(u32x2*)(dst + 2 * BN * XLD) = u32x2{l0, l1};
                }
            }
        }
    };

    // H: acc = sum h*h, accx = sum (h*l' + l'*h) with l' = 2^11 * l: the cross terms keep full relative precision for operands of
    // any magnitude (an output dominated by outlier x tiny needs the tiny element's low bits; tests/test_direct_gpu.py "outlier")
    f32x16 acc[2][TN], accx[H ? 2 : 1][H ? TN : 1];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[mi][ni][r] = 0.f;
                if constexpr (H) accx[mi][ni][r] = 0.f;
            }

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (nchunks > 0) load_tiles(S0{}, ps);
    if (nchunks > 1) load_tiles(S1{}, ps + XK);
    if (nchunks > 0) store_tiles(S0{});
    __syncthreads();

    const unsigned short* fa = sA + (wm * 64 + l31) * XLD + 8 * lh;
    const unsigned short* fb = sB + (wn * (BN / 2) + l31) * XLD + 8 * lh;
    // step c: chunk c is in LDS, chunk c+1 in register set (c+1)&1 (requested one step ago), chunk c+2 is requested into set c&1
    auto step = [&](int c, auto cur, auto nxt) {
        if (c + 2 < nchunks) load_tiles(cur, ps + (long)(c + 2) * XK);
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            if (p.dbg & 4) break;          // measurement: no fragment reads, no MFMAs
            FT a[NP][2], b[NP][TN];
#pragma unroll
            for (int pl = 0; pl < (S16 ? 1 : NP); ++pl) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[pl][mi] = *(const FT*)(fa + pl * BM * XLD + mi * 32 * XLD + ks * 16);
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) b[pl][ni] = *(const FT*)(fb + pl * BN * XLD + ni * 32 * XLD + ks * 16);
            }
            constexpr int NQ = H ? 3 : 6;
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
            constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        if constexpr (H) {
                            if (q == 2) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
                            else if (!S16) accx[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], accx[mi][ni], 0, 0, 0);
                        } else
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][mi], b[PB[q]][ni], acc[mi][ni], 0, 0, 0);
                    }
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            store_tiles(nxt);
            __syncthreads();
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, S0{}, S1{});
        if (c + 1 < nchunks) step(c + 1, S1{}, S0{});
    }

    float* part = p.part + (long)bz * M * p.Cb;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + wn * (BN / 2) + ni * 32 + l31;
        if (n >= p.Cb) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    if constexpr (H) part[(long)m * p.Cb + n] = (acc[mi][ni][r] + accx[mi][ni][r] * (1.f / 2048.f)) * out_scale;
                    else part[(long)m * p.Cb + n] = acc[mi][ni][r];
                }
            }
        }
    }
}

template <int BN, bool H, typename T = float>
int launch_wgrad_x6h(const WGradParams& p, hipStream_t s) {
    const int M = p.ntaps * p.Ca;
    dim3 grid((M + XBM - 1) / XBM, (p.Cb + BN - 1) / BN, p.splits * (p.nbatch > 1 ? p.nbatch : 1));
    const size_t smem = (size_t)(H ? 2 : 3) * (XBM + BN) * XLD * sizeof(unsigned short);
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)wgrad_x6_kernel<BN, H, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    char pname[64];
    if (getenv("SS_PROF_SHAPES")) snprintf(pname, sizeof(pname), "wgrad_x6<%d,%d> M%d N%d K%ld b%d", BN, (int)H, M, p.Cb, (long)p.N * p.GH * p.GW, p.nbatch);
    else snprintf(pname, sizeof(pname), "wgrad_x6_kernel<%d,%s%s>", BN, H ? "true" : "false", std::is_same<T, float>::value ? "" : ",16-bit");
    const double pix = (double)p.N * p.GH * p.GW * (p.nbatch > 1 ? p.nbatch : 1);
    SsProfScope prof(pname, 2.0 * M * p.Cb * pix * (std::is_same<T, float>::value ? (H ? 3 : 6) : 1), (double)sizeof(T) * pix * (p.Ca + p.Cb) + 4.0 * M * p.Cb * p.splits, s);
    WGradParams pd = p;
    pd.dbg = ss_tuning().tile_dbg;
    hipLaunchKernelGGL((wgrad_x6_kernel<BN, H, T>), grid, dim3(256), smem, s, pd);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <int BN>
int launch_wgrad_x6(const WGradParams& p, hipStream_t s) {
    if (p.dtype != SS_DTYPE_F32) {          // 16-bit stored operands: the fp16 matrix cores, one plane each
        if (!p.h_amax) { ss_set_error("wgrad_x6: 16-bit storage needs the operand maxima (x3h)"); return SS_ERR_UNSUPPORTED; }
        return p.dtype == SS_DTYPE_F16 ? launch_wgrad_x6h<BN, true, _Float16>(p, s) : launch_wgrad_x6h<BN, true, __bf16>(p, s);
    }
    return p.h_amax ? launch_wgrad_x6h<BN, true>(p, s) : launch_wgrad_x6h<BN, false>(p, s);
}

template <int BM, int BN, bool H, typename T = float, int NPROD = 3>
int launch_x6h(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph) {
    const long M = (long)p.N * p.OHc * p.OWc;
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    const int nph = ph && ph->count > 1 ? ph->count : 1;
    const long nwg1 = ((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN) * nb;
    if (nph > 1 && nwg1 % 8) return SS_ERR_UNSUPPORTED;          // several problems per launch: whole XCD rounds per problem
    dim3 grid((unsigned)(nwg1 * nph));
    int mt = p.ntaps, taps_all = p.ntaps;
    if (nph > 1) { taps_all = 0; for (int i = 0; i < nph; ++i) { taps_all += ph->ntaps[i]; mt = ph->ntaps[i] > mt ? ph->ntaps[i] : mt; } }
    const size_t smem = (size_t)(H ? 2 : 3) * (BM + BN) * XLD * sizeof(unsigned short) + (size_t)BM * sizeof(int) * (4 + mt);
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gconv_x6_kernel<BM, BN, H, T, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    char pname[64];
    if (getenv("SS_PROF_SHAPES")) snprintf(pname, sizeof(pname), "gconv_x6<%d,%d,%d> M%ld N%d K%dx%d s%d b%d", BM, BN, (int)H, M, p.Cout, p.ntaps, p.Cin, p.in_s, nb);
    else if (std::is_same<T, float>::value) snprintf(pname, sizeof(pname), "gconv_x6_kernel<%d,%d,%s>", BM, BN, H ? "true" : "false");
    else snprintf(pname, sizeof(pname), "gconv_x6_kernel<%d,%d,16-bit,%d>", BM, BN, NPROD);
    SsProfScope prof(pname, 2.0 * M * p.Cout * taps_all * p.Cin * nb * (std::is_same<T, float>::value ? (H ? 3 : 6) : NPROD),
                     (double)sizeof(T) * nb * ((double)p.N * p.IH * p.IW * p.Cin + (double)M * p.Cout * nph) + 4.0 * nb * taps_all * p.Cin * p.Cout, s);
    GPhases phv{};
    if (nph > 1) phv = *ph;
    hipLaunchKernelGGL((gconv_x6_kernel<BM, BN, H, T, NPROD>), grid, dim3(256), smem, s, p, planes, plane_elems, Npad, Ktot, phv);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <int BM, int BN>
int launch_x6(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph) {
    if (p.dtype != SS_DTYPE_F32) {          // 16-bit stored activations: typed loaders, one fp16 operand plane (x3h)
        if (!ss_gconv_x6_typed_ok(p)) { ss_set_error("gconv_x6: this 16-bit problem has no typed loader"); return SS_ERR_UNSUPPORTED; }
        const bool two = ss_tuning().wino16_products == 3;
        if (p.dtype == SS_DTYPE_F16)
            return two ? launch_x6h<BM, BN, true, _Float16, 2>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_x6h<BM, BN, true, _Float16, 1>(p, planes, plane_elems, Npad, Ktot, s, ph);
        return two ? launch_x6h<BM, BN, true, __bf16, 2>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_x6h<BM, BN, true, __bf16, 1>(p, planes, plane_elems, Npad, Ktot, s, ph);
    }
    return p.h_amax ? launch_x6h<BM, BN, true>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_x6h<BM, BN, false>(p, planes, plane_elems, Npad, Ktot, s, ph);
}

}  // namespace

int ss_x6_npad(int cout) { return (cout + 127) / 128 * 128; }

bool ss_gconv_x6_ok(const GConvParams& p) {
    const long in_elems = (long)p.N * p.IH * p.IW * p.in_cs;       // per batched problem: the LDS offset table holds 32-bit element offsets
    return p.ntaps >= 1 && p.Cin >= 16 && p.Cout >= 32 &&
           in_elems < (1L << 31) && (long)p.N * p.OHc * p.OWc < (1L << 31) && (p.nbatch <= 1 || (p.in_bs % 4 == 0));
}

size_t ss_gconv_x6_planes_bytes(const GConvParams& p) {
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    return ss_align_up((size_t)3 * nb * ss_x6_npad(p.Cout) * p.ntaps * ((p.Cin + 31) / 32 * 32) * sizeof(unsigned short), 256);
}

// weights of `p` (fp32, addressed through p.w / taps / ldb / w_bs) -> the three K-contiguous bf16 planes
// 16-bit storage through gconv_x6_kernel: the x3h maxima present (any channel count, stride and alignment: the kernel's ragged loader and
// its element-wise epilogue take the MultiResUNet's odd widths in the stored type as they do in fp32)
bool ss_gconv_x6_typed_ok(const GConvParams& p) {
    // gconv16_ragged bits (measurement): 1 = stride-1 problems, 2 = strided-output problems (sub-pixel phases of a data gradient /
    // transposed forward), 4 = 1 x 1 kernels; a ragged problem takes the typed path only if all its classes are switched on
    const int need = (p.out_s > 1 ? 2 : 1) | (p.ntaps == 1 ? 4 : 0);
    if ((ss_tuning().gconv16_ragged & need) != need)          // (measurement: the round-5 rule -- whole 32-channel chunks at 8-byte aligned pixels, 4-aligned outputs)
        return p.h_amax && p.h_amax2 && p.Cin % 32 == 0 && (p.in_cs & 3) == 0 && (((uintptr_t)p.in) & 7) == 0 && p.Cout % 4 == 0 && (p.out_cs & 3) == 0 &&
               (((uintptr_t)p.out) & 7) == 0;
    return p.h_amax && p.h_amax2;
}

int ss_launch_wprep_x6(const GConvParams& p, unsigned short* planes, hipStream_t s) {
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    const int Npad = ss_x6_npad(p.Cout), Ktot = p.ntaps * ((p.Cin + 31) / 32 * 32);
    const long plane_elems = (long)nb * Npad * Ktot;
    if (ss_wrec_on()) {          // recorded (ss_wprep_*): the plan replays it inside one launch per arithmetic
        SsWJob j{};
        j.type = p.h_amax ? SS_WJ_WPREP_H : SS_WJ_WPREP_3;
        j.gx = (Ktot + 31) / 32; j.gy = Npad / 32; j.gz = nb;
        j.dst = planes; j.n = plane_elems; j.a = Npad; j.b = Ktot;
        j.p = p;
        j.p.in = nullptr; j.p.out = nullptr; j.p.bias = nullptr; j.p.h_amax = nullptr; j.p.stats = nullptr;      // weights-only view (h_amax2, the weights' maximum, stays)
        ss_wrec_push(j);
        return SS_OK;
    }
    if (p.h_amax) hipLaunchKernelGGL(wprep_x6_kernel<true>, dim3((Ktot + 31) / 32, Npad / 32, nb), dim3(256), 0, s, p, planes, plane_elems, Npad, Ktot);
    else hipLaunchKernelGGL(wprep_x6_kernel<false>, dim3((Ktot + 31) / 32, Npad / 32, nb), dim3(256), 0, s, p, planes, plane_elems, Npad, Ktot);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// The same for `count` problems that differ in the output phase, the taps and the weight planes only (GPhases, common.h): ONE launch.
// SS_ERR_UNSUPPORTED (nothing launched) when the problems do not line up -- the caller then launches them one by one.
int ss_launch_gconv_x6_multi(const GConvParams* ps, const unsigned short* const* planes, int count, hipStream_t s) {
    if (count < 2 || count > SS_MAX_PHASES) return SS_ERR_UNSUPPORTED;
    const GConvParams& p0 = ps[0];
    GPhases ph{};
    ph.count = count;
    const int nb = p0.nbatch > 1 ? p0.nbatch : 1;
    const int Npad = ss_x6_npad(p0.Cout), Cq = (p0.Cin + 31) / 32 * 32;
    const bool v2 = ss_gconv_x6v2_ok(p0);
    for (int i = 0; i < count; ++i) {
        const GConvParams& q = ps[i];
        if (!ss_gconv_x6_ok(q) || q.ntaps > SS_MAX_PHASE_TAPS || q.OHc != p0.OHc || q.OWc != p0.OWc || q.N != p0.N || q.Cin != p0.Cin || q.Cout != p0.Cout ||
            q.in != p0.in || q.out != p0.out || q.in_s != p0.in_s || q.out_s != p0.out_s || q.in_oy != p0.in_oy || q.in_ox != p0.in_ox || q.nbatch != p0.nbatch ||
            q.h_amax != p0.h_amax || q.h_amax2 != p0.h_amax2 || q.accumulate != p0.accumulate || q.act != p0.act || q.dtype != p0.dtype || q.stats ||
            ss_gconv_x6v2_ok(q) != v2)
            return SS_ERR_UNSUPPORTED;
        ph.out_oy[i] = q.out_oy; ph.out_ox[i] = q.out_ox; ph.ntaps[i] = q.ntaps; ph.Ktot[i] = q.ntaps * Cq;
        for (int t = 0; t < q.ntaps; ++t) { ph.tdy[i][t] = q.taps[t].dy; ph.tdx[i][t] = q.taps[t].dx; }
        ph.planes[i] = planes[i];
        ph.plane_elems[i] = (long)nb * Npad * ph.Ktot[i];
    }
    const long M = (long)p0.N * p0.OHc * p0.OWc;
    if (M == 0) return SS_OK;
    if (v2) return ss_launch_gconv_x6v2(p0, planes[0], ph.plane_elems[0], Npad, ph.Ktot[0], s, &ph);
    const long want = 200;
    auto nblk = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p0.Cout + bn - 1) / bn) * nb; };
    if (p0.Cout > 64 && nblk(128, 128) >= want) return launch_x6<128, 128>(p0, planes[0], ph.plane_elems[0], Npad, ph.Ktot[0], s, &ph);
    if (nblk(128, 64) >= want) return launch_x6<128, 64>(p0, planes[0], ph.plane_elems[0], Npad, ph.Ktot[0], s, &ph);
    return launch_x6<64, 64>(p0, planes[0], ph.plane_elems[0], Npad, ph.Ktot[0], s, &ph);
}

int ss_launch_gconv_x6(const GConvParams& p, const unsigned short* planes, hipStream_t s) {
    const long M = (long)p.N * p.OHc * p.OWc;
    if (M == 0) return SS_OK;
    if (!ss_gconv_x6_ok(p)) return SS_ERR_UNSUPPORTED;
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    const int Npad = ss_x6_npad(p.Cout), Ktot = p.ntaps * ((p.Cin + 31) / 32 * 32);
    const long plane_elems = (long)nb * Npad * Ktot;
    const GPhases* ph = nullptr;
    if (ss_gconv_x6v2_ok(p)) return ss_launch_gconv_x6v2(p, planes, plane_elems, Npad, Ktot, s);
    // tile choice: the largest tile that still yields >= ~200 workgroups (small grids at per-GPU batch 1 want more, smaller ones)
    const long want = 200;      // swept at per-GPU batch 1 / 2 (200 / 600 / 1200): 200 is the fastest
    auto nblk = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * nb; };
    if (p.Cout > 64 && nblk(128, 128) >= want) return launch_x6<128, 128>(p, planes, plane_elems, Npad, Ktot, s, ph);
    if (nblk(128, 64) >= want) return launch_x6<128, 64>(p, planes, plane_elems, Npad, Ktot, s, ph);
    return launch_x6<64, 64>(p, planes, plane_elems, Npad, Ktot, s, ph);
}

bool ss_wgrad_x6_ok(const WGradParams& p) {
    return p.ntaps >= 1 && p.Ca % 32 == 0 && p.a_cs % 4 == 0 && (((uintptr_t)p.a) & 15) == 0 && p.Cb >= 32 && p.Cb % 4 == 0 &&
           p.b_cs % 4 == 0 && (((uintptr_t)p.b) & 15) == 0 && p.GW >= 4 && p.pix_per_split % 32 == 0 &&
           (p.nbatch <= 1 || (p.a_bs % 4 == 0 && p.b_bs % 4 == 0));
}

// partials only: part[batch][split][(t,ca)][cb]
int ss_launch_wgrad_x6_partials(const WGradParams& p, hipStream_t s) {
    if (!ss_wgrad_x6_ok(p)) return SS_ERR_UNSUPPORTED;
    if (p.Cb > 64) return launch_wgrad_x6<128>(p, s);
    return launch_wgrad_x6<64>(p, s);
}

int ss_wbatch_launch_wprep(bool h, const SsWJob* jobs, const int* map, int nblocks, hipStream_t s) {
    if (h) hipLaunchKernelGGL(wprep_x6_batch_kernel<true>, dim3(nblocks), dim3(256), 0, s, jobs, map);
    else hipLaunchKernelGGL(wprep_x6_batch_kernel<false>, dim3(nblocks), dim3(256), 0, s, jobs, map);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
