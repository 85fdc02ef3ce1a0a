// Matrix-core (MFMA) implicit-GEMM kernels for gfx950, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
//  gconv_mfma : C[pixel][cout] = sum_{tap,ci} A_gather[pixel][(tap,ci)] * W[(tap,ci)][cout]
//               serves conv forward, conv backward-data (transposed weights, parity classes for
//               stride 2) and through them Conv2DTranspose forward/backward-data.
//  wgrad_mfma : C[(tap,ca)][cb] = sum_pixel A_gather[pixel][(tap,ca)] * B[pixel][cb]   (split over pixels)
//
// Tiling: 256 threads = 4 wave64; block tile BM x BN, K step 32; operands staged global -> VGPR ->
// LDS (double buffered, one barrier per K step, next tile's global loads issued before the MFMAs).
// LDS layouts are chosen so that the MFMA operand reads are bank-conflict free:
//   A (pixel-major, row stride 36 floats): lane (i = l&31, h = l>>5) reads one ds_read_b128 holding
//     k = 8*kk + 4h .. +3; the four MFMAs of a group consume k = 8kk+jj (h=0) and 8kk+4+jj (h=1),
//     a permutation of the K order that A and B share (sums are order-independent up to rounding).
//   B (k-major): lanes 0..31 read consecutive cout -> consecutive banks.
#include "common.h"
#include <stdlib.h>

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte global access

template <int BM_, int BN_, int NT_ = 256>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, BK = 32, NT = NT_;
    static constexpr int WAVES_N = (BN >= 64) ? 2 : 1;
    static constexpr int WAVES_M = (NT / 64) / WAVES_N;
    static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    static constexpr int TM = WM / 32, TN = WN / 32;
    static constexpr int LDA = BK + 4;   // gconv A: [BM][LDA]
    static constexpr int LDB = BN + 4;   // B: [BK][LDB]
    static constexpr int LDAT = BM + 4;  // wgrad A: [BK][LDAT]
    static constexpr int A_UNITS = BM * (BK / 4) / NT;
    static constexpr int B_UNITS = BK * (BN / 4) / NT;
    static constexpr int AT_UNITS = BK * (BM / 4) / NT;
    static constexpr int A_ROWS = NT / 8;     // pixel rows covered by one pass of the A loader
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    static_assert(A_UNITS >= 1 && B_UNITS >= 1, "tile too small");
    static constexpr size_t smem_gconv = (2 * BM * LDA + 2 * BK * LDB) * sizeof(float) + BM * sizeof(int);
    static constexpr size_t smem_wgrad = (2 * BK * LDAT + 2 * BK * LDB) * sizeof(float);
};

// ------------------------------------------------------------------------------------------------
// FAST: Cin % 32 == 0 (every K step lies inside one tap), float4-aligned operands.  The gather offsets of the
// block's BM pixels for every tap are precomputed once into LDS (offtab), and the per-step global loads are
// branch-free (clamped address + select) so that the compiler can interleave them with the MFMA stream.
template <int BM, int BN, bool FAST, int NT>
__global__ __launch_bounds__(NT) void gconv_mfma_kernel(GConvParams p, int vecA, int vecB) {
    using C = Cfg<BM, BN, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * C::LDA;
    int* pixtab = (int*)(Bs + 2 * C::BK * C::LDB);
    int* offtab = pixtab + BM;      // [BM][ntaps] element offsets of the gathered pixel, -1 = zero padding (FAST only)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const long M = (long)p.N * p.OHc * p.OWc;
    // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (speed only, never correctness).  Give each
    // XCD a contiguous chunk of the tile space, N-tile fastest, so that the workgroups resident on one XCD share the
    // same few activation rows (A) across their N-tiles and neighbouring M-tiles share halo rows in that XCD's L2.
    const int gridN = (p.Cout + BN - 1) / BN;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int gridM = (int)((M + BM - 1) / BM);
    const int batch = tile / (gridM * gridN);      // batched problems: consecutive tiles of one XCD chunk stay in one batch
    tile -= batch * gridM * gridN;
    const float* const g_in = p.in + (long)batch * p.in_bs;
    const float* const g_w = p.w + (long)batch * p.w_bs;
    float* const g_out = p.out + (long)batch * p.out_bs;
    const long m0 = (long)(tile / gridN) * BM;
    const int n0 = (tile % gridN) * BN;
    // FAST with vecA == 0 ("padded K"): any channel count.  The reduction index runs over (tap, ci padded to a multiple of 4),
    // every 4-float unit stays inside one tap and is fetched with ONE dword-aligned global_load_dwordx4 when it lies fully
    // inside the pixel's channels (scalar masked loads for the ragged last unit) -- the MultiResUNet's odd widths.
    const int Cq = (p.Cin + 3) & ~3;
    const bool padk = FAST && !vecA;
    const int K = padk ? p.ntaps * Cq : p.ntaps * p.Cin;
    const int nchunks = (K + C::BK - 1) / C::BK;

    // output pixel table (linear NHW index of the destination pixel, -1 = masked)
    if (FAST) {
        // 32-bit pixel decode (the launcher guarantees M < 2^31), ONE division pair per tile row, kept in LDS; the gather-offset
        // table below divides nothing.  (The 64-bit `m % OWc`, `r / OHc` per table entry -- emulated, ~150 instructions each --
        // were most of a workgroup's life for the MultiResUNet's full-resolution layers, whose K loop is 1-8 steps.)
        int* rowc = offtab + BM * p.ntaps;      // [3][BM]: image index, base input y, base input x of the row's pixel
        if (tid < BM) {
            const unsigned m = (unsigned)m0 + (unsigned)tid;
            int v = -1, rn = -1, ry = 0, rx = 0;
            if (m < (unsigned)M) {
                const unsigned r = m / (unsigned)p.OWc;
                const int xc = (int)(m - r * (unsigned)p.OWc);
                const unsigned n = r / (unsigned)p.OHc;
                const int yc = (int)(r - n * (unsigned)p.OHc);
                const int oy = yc * p.out_s + p.out_oy, ox = xc * p.out_s + p.out_ox;
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
        {
            constexpr int TSTEP = NT / BM;
            const int row = tid % BM;
            const int rn = rowc[row], ry = rowc[BM + row], rx = rowc[2 * BM + row];
            for (int t = tid / BM; t < p.ntaps; t += TSTEP) {
                int off = -1;
                if (rn >= 0) {
                    const int iy = ss_map_index(ry + p.taps[t].dy, p.IH, p.reflect);
                    const int ix = ss_map_index(rx + p.taps[t].dx, p.IW, p.reflect);
                    if (iy >= 0 && ix >= 0) off = ((rn * p.IH + iy) * p.IW + ix) * p.in_cs;
                }
                offtab[row * p.ntaps + t] = off;
            }
        }
        __syncthreads();
    } else if (tid < BM) {
        const long m = m0 + tid;
        int v = -1;
        if (m < M) {
            const int xc = (int)(m % p.OWc);
            const long r = m / p.OWc;
            const int yc = (int)(r % p.OHc);
            const int n = (int)(r / p.OHc);
            const int oy = yc * p.out_s + p.out_oy, ox = xc * p.out_s + p.out_ox;
            if (oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW) v = (n * p.OH + oy) * p.OW + ox;
        }
        pixtab[tid] = v;
    }

    // per-thread A rows: row = (tid>>3) + 32*j, float4 column c4a = tid&7
    const int c4a = tid & 7;
    int a_nb[C::A_UNITS], a_by[C::A_UNITS], a_bx[C::A_UNITS];
#pragma unroll
    for (int j = 0; j < C::A_UNITS; ++j) {
        const long m = m0 + (tid >> 3) + C::A_ROWS * j;
        if (!FAST && m < M) {
            const int xc = (int)(m % p.OWc);
            const long r = m / p.OWc;
            const int yc = (int)(r % p.OHc);
            a_nb[j] = (int)(r / p.OHc) * p.IH;
            a_by[j] = yc * p.in_s + p.in_oy;
            a_bx[j] = xc * p.in_s + p.in_ox;
        } else {
            a_nb[j] = -1; a_by[j] = 0; a_bx[j] = 0;
        }
    }

    f32x4 ra[C::A_UNITS], rb[C::B_UNITS];
    // FAST: units that must read as zero (padding taps, columns past Cout).  The select is applied when the registers are
    // stored to LDS, NOT after the load: a select right behind the load makes the compiler wait for the global loads before
    // the MFMA phase (s_waitcnt vmcnt(0) ahead of 64 MFMAs), exposing the whole L2 / HBM latency in every K step.
    bool za[C::A_UNITS], zb[C::B_UNITS];

    auto load_tiles = [&](int k0) {
        if (FAST && padk) {
            const int kq = k0 + c4a * 4;
            const int tq = kq / Cq;
            const int ci = kq - tq * Cq;
            const bool kval = tq < p.ntaps;
            const int t = kval ? tq : 0;
            const int nin = p.Cin - ci;               // channels of this unit that exist (>= 4: whole unit)
#pragma unroll
            for (int j = 0; j < C::A_UNITS; ++j) {
                const int off = offtab[((tid >> 3) + C::A_ROWS * j) * p.ntaps + t];
                const bool ok = kval && off >= 0;
                const float* ptr = g_in + (ok ? off : 0) + (ok ? ci : 0);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (nin >= 4) {
                    const f32x4u u = *(const f32x4u*)ptr;
                    v = f32x4{u[0], u[1], u[2], u[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (e < nin) v[e] = ptr[e];
                }
                ra[j] = v;
                za[j] = !ok;
            }
#pragma unroll
            for (int j = 0; j < C::B_UNITS; ++j) {
                const int u = tid + NT * j;
                const int row = u / (BN / 4), c4 = u % (BN / 4);
                const int kb = k0 + row;
                const int tb = kb / Cq;
                const int cb = kb - tb * Cq;
                const int col = n0 + c4 * 4;
                const bool okb = tb < p.ntaps && cb < p.Cin && col < p.Cout;
                const float* wp = g_w + p.taps[okb ? tb : 0].woff + (long)(okb ? cb : 0) * p.ldb + (okb ? col : 0);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (col + 4 <= p.Cout) {
                    const f32x4u u4 = *(const f32x4u*)wp;
                    v = f32x4{u4[0], u4[1], u4[2], u4[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (col + e < p.Cout) v[e] = wp[e];
                }
                rb[j] = v;
                zb[j] = !okb;
            }
            return;
        }
        if (FAST) {
            const int t = k0 / p.Cin;                 // block-uniform
            const int ci0 = k0 - t * p.Cin;
            const float* abase = g_in + ci0 + c4a * 4;
#pragma unroll
            for (int j = 0; j < C::A_UNITS; ++j) {
                const int off = offtab[((tid >> 3) + C::A_ROWS * j) * p.ntaps + t];
                ra[j] = *(const f32x4*)(abase + (off < 0 ? 0 : off));
                za[j] = off < 0;
            }
            const float* bbase = g_w + p.taps[t].woff + (long)ci0 * p.ldb;
#pragma unroll
            for (int j = 0; j < C::B_UNITS; ++j) {
                const int u = tid + NT * j;
                const int row = u / (BN / 4), c4 = u % (BN / 4);
                const int col = n0 + c4 * 4;
                const int colc = col + 4 <= p.Cout ? col : p.Cout - 4;
                rb[j] = *(const f32x4*)(bbase + (long)row * p.ldb + colc);
                zb[j] = col + 4 > p.Cout;
            }
            return;
        }
        // ---- A: gathered activations ----
        const int k = k0 + c4a * 4;
        if (vecA) {
            const bool kval = k < K;
            const int t = kval ? k / p.Cin : 0;
            const int ci = k - t * p.Cin;
            const int dy = p.taps[t].dy, dx = p.taps[t].dx;
#pragma unroll
            for (int j = 0; j < C::A_UNITS; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kval && a_nb[j] >= 0) {
                    const int iy = ss_map_index(a_by[j] + dy, p.IH, p.reflect);
                    const int ix = ss_map_index(a_bx[j] + dx, p.IW, p.reflect);
                    if (iy >= 0 && ix >= 0)
                        v = *(const f32x4*)(g_in + ((long)(a_nb[j] + iy) * p.IW + ix) * p.in_cs + ci);
                }
                ra[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < C::A_UNITS; ++j) ra[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                if (ke < K) {
                    const int t = ke / p.Cin;
                    const int ci = ke - t * p.Cin;
                    const int dy = p.taps[t].dy, dx = p.taps[t].dx;
#pragma unroll
                    for (int j = 0; j < C::A_UNITS; ++j) {
                        if (a_nb[j] >= 0) {
                            const int iy = ss_map_index(a_by[j] + dy, p.IH, p.reflect);
                            const int ix = ss_map_index(a_bx[j] + dx, p.IW, p.reflect);
                            if (iy >= 0 && ix >= 0)
                                ra[j][e] = g_in[((long)(a_nb[j] + iy) * p.IW + ix) * p.in_cs + ci];
                        }
                    }
                }
            }
        }
        // ---- B: weights ----
#pragma unroll
        for (int j = 0; j < C::B_UNITS; ++j) {
            const int u = tid + NT * j;
            const int row = u / (BN / 4), c4 = u % (BN / 4);
            const int kb = k0 + row;
            const int col = n0 + c4 * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kb < K && col < p.Cout) {
                const int t = kb / p.Cin;
                const int ci = kb - t * p.Cin;
                const float* wp = g_w + p.taps[t].woff + (long)ci * p.ldb + col;
                if (vecB && col + 3 < p.Cout) {
                    v = *(const f32x4*)wp;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.Cout) v[e] = wp[e];
                }
            }
            rb[j] = v;
        }
    };

    auto store_tiles = [&](int buf) {
        float* Ab = As + buf * BM * C::LDA;
        float* Bb = Bs + buf * C::BK * C::LDB;
#pragma unroll
        for (int j = 0; j < C::A_UNITS; ++j)
            *(f32x4*)(Ab + ((tid >> 3) + C::A_ROWS * j) * C::LDA + c4a * 4) = (FAST && za[j]) ? f32x4{0.f, 0.f, 0.f, 0.f} : ra[j];
#pragma unroll
        for (int j = 0; j < C::B_UNITS; ++j) {
            const int u = tid + NT * j;
            const int row = u / (BN / 4), c4 = u % (BN / 4);
            *(f32x4*)(Bb + row * C::LDB + c4 * 4) = (FAST && zb[j]) ? f32x4{0.f, 0.f, 0.f, 0.f} : rb[j];
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_tiles((c + 1) * C::BK);
        const float* Ab = As + buf * BM * C::LDA + (wm * C::WM + l31) * C::LDA + 4 * lh;
        const float* Bb = Bs + buf * C::BK * C::LDB + (4 * lh) * C::LDB + wn * C::WN + l31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[C::TM];
            float b[C::TN][4];
#pragma unroll
            for (int mi = 0; mi < C::TM; ++mi) a[mi] = *(const f32x4*)(Ab + mi * 32 * C::LDA + kk * 8);
#pragma unroll
            for (int ni = 0; ni < C::TN; ++ni)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) b[ni][jj] = Bb[(kk * 8 + jj) * C::LDB + ni * 32];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < C::TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][jj], b[ni][jj], acc[mi][ni], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < C::TN; ++ni) {
        const int co = n0 + wn * C::WN + ni * 32 + l31;
        if (co >= p.Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < C::TM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int pix = pixtab[wm * C::WM + mi * 32 + row];
                if (pix < 0) continue;
                float* op = g_out + (long)pix * p.out_cs + co;
                float v = ss_apply_act(acc[mi][ni][r] + bv, p.act, p.alpha);
                if (p.accumulate) v += *op;
                *op = v;
            }
        }
    }
}

bool ss_gconv_mfma_ok(const GConvParams& p) {
    // Cout == 1 heads and degenerate reductions stay on the direct kernel
    return p.Cout >= 2 && (long)p.ntaps * p.Cin >= 1;      // K is zero-padded to the 32-wide step by the loaders
}

template <int BM, int BN, bool FAST, int NT = 256>
static int launch_gconv(const GConvParams& p, int vecA, int vecB, hipStream_t s) {
    using C = Cfg<BM, BN, NT>;
    const long M = (long)p.N * p.OHc * p.OWc;
    dim3 grid((unsigned)(((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN) * (p.nbatch > 1 ? p.nbatch : 1)));
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gconv_mfma_kernel<BM, BN, FAST, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const size_t smem = C::smem_gconv + (FAST ? (size_t)BM * (p.ntaps + 3) * sizeof(int) : 0);
    hipLaunchKernelGGL((gconv_mfma_kernel<BM, BN, FAST, NT>), grid, dim3(NT), smem, s, p, vecA, vecB);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_launch_gconv_mfma(const GConvParams& p, hipStream_t s) {
    const long M = (long)p.N * p.OHc * p.OWc;
    if (M == 0) return SS_OK;
    int vecA = (p.Cin % 4 == 0) && (p.in_cs % 4 == 0) && (((uintptr_t)p.in & 15) == 0);
    int vecB = (p.ldb % 4 == 0) && (((uintptr_t)p.w & 15) == 0);
    for (int t = 0; t < p.ntaps && vecB; ++t) vecB = (p.taps[t].woff % 4 == 0);
    const long in_elems = (long)p.N * p.IH * p.IW * p.in_cs;
    const bool uniform = vecA && vecB && (p.Cin % 32 == 0) && (p.Cout >= 4);      // aligned float4 units, one tap per K step
    const bool fast = p.ntaps >= 1 && in_elems < (1L << 31) && M < (1L << 31) && ss_tuning().gconv_fast;
    if (fast && !uniform) vecA = 0;                                                // -> padded-K loaders
    // tile choice: the largest tile that still gives every CU ~2 workgroups (256 CUs); small batches need small tiles
    auto nblocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * (p.nbatch > 1 ? p.nbatch : 1); };
    const long want = 480;
    int cfg;   // 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32
    if (p.Cout > 64) cfg = nblocks(128, 128) >= want ? 0 : (nblocks(128, 64) >= want ? 1 : 2);
    else if (p.Cout > 32) cfg = nblocks(128, 64) >= want ? 1 : 2;
    else cfg = 3;
    if (fast) {
        switch (cfg) {
            case 0: {
                if (ss_tuning().tile256 && nblocks(256, 128) >= want) return launch_gconv<256, 128, true, 512>(p, vecA, vecB, s);
                if (ss_tuning().nt512) return launch_gconv<128, 128, true, 512>(p, vecA, vecB, s);
                return launch_gconv<128, 128, true>(p, vecA, vecB, s);
            }
            case 1: return launch_gconv<128, 64, true>(p, vecA, vecB, s);
            case 2: return launch_gconv<64, 64, true>(p, vecA, vecB, s);
            default: return launch_gconv<128, 32, true>(p, vecA, vecB, s);
        }
    }
    switch (cfg) {
        case 0: return launch_gconv<128, 128, false>(p, vecA, vecB, s);
        case 1: return launch_gconv<128, 64, false>(p, vecA, vecB, s);
        case 2: return launch_gconv<64, 64, false>(p, vecA, vecB, s);
        default: return launch_gconv<128, 32, false>(p, vecA, vecB, s);
    }
}

// ------------------------------------------------------------------------------------------------
// FAST: float4-aligned operands, grid width >= 32: the pixel coordinates of this thread's rows advance
// incrementally (no divisions in the loop) and all global loads are branch-free (clamped address + select).
// X6: the contraction on the bf16 matrix cores with the exact three-piece split (conv_mfma_x6.hip's arithmetic: six products, dropped
// terms <= 2^-26) formed IN REGISTERS from the same fp32 LDS tiles -- the shapes wgrad_x6_kernel does not take (channel counts that
// are not multiples of 32 / unaligned views: the MultiResUNet's odd widths) ran on v_mfma_f32_32x32x2_f32 at 1/16 of the 16-bit rate.
// A lane's fragment for v_mfma_f32_32x32x16_bf16 is 8 consecutive k of its row / column: the same 8 LDS reads per K = 16 the fp32
// instruction needed, then 4 x ss_split3x2.
template <int BM, int BN, bool FAST, bool X6 = false>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WGradParams p, int vecA, int vecB) {
    using C = Cfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BK][LDAT]
    float* Bs = smem + 2 * C::BK * C::LDAT;    // [2][BK][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;

    const int M = p.ntaps * p.Ca;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z % p.splits;
    const int batch = blockIdx.z / p.splits;       // batched problems (Winograd transform positions)
    const float* const g_a = p.a + (long)batch * p.a_bs;
    const float* const g_b = p.b + (long)batch * p.b_bs;
    const long P = (long)p.N * p.GH * p.GW;
    const long ps = (long)split * p.pix_per_split;
    const long pe = (ps + p.pix_per_split < P) ? ps + p.pix_per_split : P;
    const int nchunks = (int)((pe - ps + C::BK - 1) / C::BK);

    // this thread's A columns m = m0 + c4*4 + e  (fixed over the pixel loop)
    const int c4a = tid % (BM / 4);
    int a_dy[4], a_dx[4], a_c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = m0 + c4a * 4 + e;
        if (m < M) {
            const int t = m / p.Ca;
            a_c[e] = m - t * p.Ca;
            a_dy[e] = p.taps[t].dy;
            a_dx[e] = p.taps[t].dx;
        } else {
            a_c[e] = -1; a_dy[e] = 0; a_dx[e] = 0;
        }
    }
    const int c4b = tid % (BN / 4);

    f32x4 ra[C::AT_UNITS], rb[C::B_UNITS];
    // FAST: per-element validity masks; the zero-selects are deferred to the LDS store (see gconv_mfma_kernel)
    unsigned char za[C::AT_UNITS], zb[C::B_UNITS];

    // FAST state: decoded coordinates of the pixel each of this thread's A rows will load next
    int f_n[C::AT_UNITS], f_y[C::AT_UNITS], f_x[C::AT_UNITS];
    if (FAST) {
#pragma unroll
        for (int j = 0; j < C::AT_UNITS; ++j) {
            const long pk = ps + (tid + 256 * j) / (BM / 4);
            f_x[j] = (int)(pk % p.GW);
            const long r = pk / p.GW;
            f_y[j] = (int)(r % p.GH);
            f_n[j] = (int)(r / p.GH);
        }
    }

    auto load_tiles = [&](long pk0) {
        if (FAST) {
            const bool mval = a_c[0] >= 0;
#pragma unroll
            for (int j = 0; j < C::AT_UNITS; ++j) {
                const int row = (tid + 256 * j) / (BM / 4);
                const bool pval = pk0 + row < pe;
                const int by = f_y[j] * p.a_s + p.a_oy, bx = f_x[j] * p.a_s + p.a_ox;
                if (vecA) {
                    const int iy = ss_map_index(by + a_dy[0], p.AH, p.reflect);
                    const int ix = ss_map_index(bx + a_dx[0], p.AW, p.reflect);
                    const bool ok = mval && pval && iy >= 0 && ix >= 0;
                    const long off = ok ? ((long)(f_n[j] * p.AH + iy) * p.AW + ix) * p.a_cs + a_c[0] : 0;
                    ra[j] = *(const f32x4*)(g_a + off);
                    za[j] = ok ? 0xF : 0;
                } else {        // odd channel counts: the 4 columns of a unit may belong to different taps -> 4 scalar gathers
                    unsigned char mk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int iy = ss_map_index(by + a_dy[e], p.AH, p.reflect);
                        const int ix = ss_map_index(bx + a_dx[e], p.AW, p.reflect);
                        const bool ok = a_c[e] >= 0 && pval && iy >= 0 && ix >= 0;
                        const long off = ok ? ((long)(f_n[j] * p.AH + iy) * p.AW + ix) * p.a_cs + a_c[e] : 0;
                        ra[j][e] = g_a[off];
                        mk |= (unsigned char)(ok ? (1 << e) : 0);
                    }
                    za[j] = mk;
                }
                // advance by one K step (32 pixels); GW >= 32 so at most one row wrap
                int x = f_x[j] + C::BK, y = f_y[j], n = f_n[j];
                if (x >= p.GW) { x -= p.GW; ++y; }
                if (y >= p.GH) { y -= p.GH; ++n; }
                f_x[j] = x; f_y[j] = y; f_n[j] = n;
            }
            const int col = n0 + c4b * 4;
#pragma unroll
            for (int j = 0; j < C::B_UNITS; ++j) {
                const int row = (tid + 256 * j) / (BN / 4);
                const long pk = pk0 + row;
                if (vecB) {
                    const int colc = col + 4 <= p.Cb ? col : (p.Cb >= 4 ? p.Cb - 4 : 0);
                    const bool ok = pk < pe && col + 4 <= p.Cb;
                    rb[j] = *(const f32x4*)(g_b + (ok ? pk : ps) * p.b_cs + colc);
                    zb[j] = ok ? 0xF : 0;
                } else {
                    unsigned char mk = 0;
                    const float* bp = g_b + (pk < pe ? pk : ps) * p.b_cs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = pk < pe && col + e < p.Cb;
                        rb[j][e] = bp[ok ? col + e : 0];
                        mk |= (unsigned char)(ok ? (1 << e) : 0);
                    }
                    zb[j] = mk;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < C::AT_UNITS; ++j) {
            const int row = (tid + 256 * j) / (BM / 4);
            const long pk = pk0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pk < pe) {
                const int xc = (int)(pk % p.GW);
                const long r = pk / p.GW;
                const int yc = (int)(r % p.GH);
                const int nb = (int)(r / p.GH) * p.AH;
                const int by = yc * p.a_s + p.a_oy, bx = xc * p.a_s + p.a_ox;
                if (vecA) {
                    if (a_c[0] >= 0) {
                        const int iy = ss_map_index(by + a_dy[0], p.AH, p.reflect);
                        const int ix = ss_map_index(bx + a_dx[0], p.AW, p.reflect);
                        if (iy >= 0 && ix >= 0)
                            v = *(const f32x4*)(g_a + ((long)(nb + iy) * p.AW + ix) * p.a_cs + a_c[0]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (a_c[e] >= 0) {
                            const int iy = ss_map_index(by + a_dy[e], p.AH, p.reflect);
                            const int ix = ss_map_index(bx + a_dx[e], p.AW, p.reflect);
                            if (iy >= 0 && ix >= 0)
                                v[e] = g_a[((long)(nb + iy) * p.AW + ix) * p.a_cs + a_c[e]];
                        }
                    }
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < C::B_UNITS; ++j) {
            const int row = (tid + 256 * j) / (BN / 4);
            const long pk = pk0 + row;
            const int col = n0 + c4b * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pk < pe && col < p.Cb) {
                const float* bp = g_b + pk * p.b_cs + col;
                if (vecB && col + 3 < p.Cb) {
                    v = *(const f32x4*)bp;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.Cb) v[e] = bp[e];
                }
            }
            rb[j] = v;
        }
    };

    auto store_tiles = [&](int buf) {
        float* Ab = As + buf * C::BK * C::LDAT;
        float* Bb = Bs + buf * C::BK * C::LDB;
#pragma unroll
        for (int j = 0; j < C::AT_UNITS; ++j) {
            const int row = (tid + 256 * j) / (BM / 4);
            f32x4 va = ra[j];
            if (FAST) {
#pragma unroll
                for (int e = 0; e < 4; ++e) va[e] = (za[j] >> e & 1) ? va[e] : 0.f;
            }
            *(f32x4*)(Ab + row * C::LDAT + c4a * 4) = va;
        }
#pragma unroll
        for (int j = 0; j < C::B_UNITS; ++j) {
            const int row = (tid + 256 * j) / (BN / 4);
            f32x4 vb = rb[j];
            if (FAST) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vb[e] = (zb[j] >> e & 1) ? vb[e] : 0.f;
            }
            *(f32x4*)(Bb + row * C::LDB + c4b * 4) = vb;
        }
    };

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (nchunks > 0) {
        load_tiles(ps);
        store_tiles(0);
    }
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_tiles(ps + (long)(c + 1) * C::BK);
        const float* Ab = As + buf * C::BK * C::LDAT + lh * C::LDAT + wm * C::WM + l31;
        const float* Bb = Bs + buf * C::BK * C::LDB + lh * C::LDB + wn * C::WN + l31;
        if constexpr (X6) {
            typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
            typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int s16 = 0; s16 < C::BK / 16; ++s16) {
                // lane (l31, lh): k = 16 s16 + 8 lh + j, j = 0..7 (Ab / Bb already point at k row lh: row 8 lh + j = lh + (7 lh + j))
                wu32x4 af[3][C::TM], bf[3][C::TN];
#pragma unroll
                for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const float v0 = Ab[(16 * s16 + 7 * lh + 2 * j2) * C::LDAT + mi * 32], v1 = Ab[(16 * s16 + 7 * lh + 2 * j2 + 1) * C::LDAT + mi * 32];
                        unsigned int h, m, l;
                        ss_split3x2(f32x2{v0, v1}, h, m, l);
                        af[0][mi][j2] = h; af[1][mi][j2] = m; af[2][mi][j2] = l;
                    }
#pragma unroll
                for (int ni = 0; ni < C::TN; ++ni)
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const float v0 = Bb[(16 * s16 + 7 * lh + 2 * j2) * C::LDB + ni * 32], v1 = Bb[(16 * s16 + 7 * lh + 2 * j2 + 1) * C::LDB + ni * 32];
                        unsigned int h, m, l;
                        ss_split3x2(f32x2{v0, v1}, h, m, l);
                        bf[0][ni][j2] = h; bf[1][ni][j2] = m; bf[2][ni][j2] = l;
                    }
                // six products, smallest terms first (conv_mfma_x6.hip): m*m, l*h, h*l, m*h, h*m, h*h
                constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < C::TN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, af[PA[q]][mi]), __builtin_bit_cast(wbf16x8, bf[PB[q]][ni]),
                                                                                  acc[mi][ni], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int s2 = 0; s2 < C::BK / 2; ++s2) {
            float a[C::TM], b[C::TN];
#pragma unroll
            for (int mi = 0; mi < C::TM; ++mi) a[mi] = Ab[(2 * s2) * C::LDAT + mi * 32];
#pragma unroll
            for (int ni = 0; ni < C::TN; ++ni) b[ni] = Bb[(2 * s2) * C::LDB + ni * 32];
#pragma unroll
            for (int mi = 0; mi < C::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < C::TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (c + 1 < nchunks) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float* part = p.part + (long)blockIdx.z * M * p.Cb;
#pragma unroll
    for (int ni = 0; ni < C::TN; ++ni) {
        const int n = n0 + wn * C::WN + ni * 32 + l31;
        if (n >= p.Cb) continue;
#pragma unroll
        for (int mi = 0; mi < C::TM; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * C::WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) part[(long)m * p.Cb + n] = acc[mi][ni][r];
            }
        }
    }
}

// dw[woff_t + ca*ldw + cb] (+)= sum_split part[split][(t,ca)][cb]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WGradParams p, float* dw, int ldw, int accumulate, int rows) {
    const long total = (long)p.ntaps * p.Ca * p.Cb;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)rows * p.Cb) return;
    const int cb = (int)(e % p.Cb);
    const long m = e / p.Cb;
    const int t = (int)(m / p.Ca);
    const int ca = (int)(m - (long)t * p.Ca);
    // 8 independent chains (8 loads in flight per thread: the 512-split reductions of the small UNet layers were latency
    // bound), combined in a fixed order -> deterministic
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= p.splits; s += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] += p.part[(long)(s + k) * total + e];
    }
    for (; s < p.splits; ++s) a8[0] += p.part[(long)s * total + e];
    const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    float* o = dw + p.taps[t].woff + (long)ca * ldw + cb;
    *o = accumulate ? (*o + acc) : acc;
}

// Many splits, few outputs (the MultiResUNet's full-resolution layers: 512 partials of a few thousand weights): one thread per
// output walks 64 dependent rounds of loads (18 us).  Here 16 lanes share an output: lane j sums the splits j, j + 16, ... (8 loads
// in flight), the 16 lane sums are combined through LDS in lane order -> deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(WGradParams p, float* dw, int ldw, int accumulate, int rows) {
    __shared__ float red[16][17];
    const long total = (long)p.ntaps * p.Ca * p.Cb;
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;          // 16 outputs x 16 split lanes per block
    const long e = (long)blockIdx.x * 16 + el;
    const bool val = e < (long)rows * p.Cb;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (val) {
        int s = sl;
        for (; s + 7 * 16 < p.splits; s += 8 * 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a8[k] += p.part[(long)(s + 16 * k) * total + e];
        }
        for (; s < p.splits; s += 16) a8[0] += p.part[(long)s * total + e];
    }
    red[sl][el] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    __syncthreads();
    if (sl == 0 && val) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += red[j][el];
        const int cb = (int)(e % p.Cb);
        const long m = e / p.Cb;
        const int t = (int)(m / p.Ca);
        const int ca = (int)(m - (long)t * p.Ca);
        float* o = dw + p.taps[t].woff + (long)ca * ldw + cb;
        *o = accumulate ? (*o + acc) : acc;
    }
}

static void launch_wgrad_reduce_any(const WGradParams& p, float* dw, int ldw, int accumulate, int rows, long total, hipStream_t s) {
    if (p.splits >= 64 && total <= (1L << 20))
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, p, dw, ldw, accumulate, rows);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, dw, ldw, accumulate, rows);
}

int ss_wgrad_mfma_splits(int64_t pixels, int M, int Cb, int* pix_per_split, int nbatch) {
    // Split the pixel (K) range so that tiles x splits fills whole "rounds" of the 512 workgroup slots
    // (256 CUs x 2 resident workgroups): time ~ ceil(tiles*s/512) * K/s.  A partial last round costs a full K/s.
    const int bn = Cb > 64 ? 128 : (Cb > 32 ? 64 : 32);
    const long tiles = (long)((M + 127) / 128) * ((Cb + bn - 1) / bn) * (nbatch > 1 ? nbatch : 1);
    long max_splits = (pixels + 255) / 256;            // >= 8 K-steps per split
    if (max_splits > 1024) max_splits = 1024;
    if (max_splits < 1) max_splits = 1;
    long best = 1;
    double best_cost = 1e30;
    // time model (us): one workgroup needs ~1.95 us per 32-pixel K step at 2 workgroups/CU; every split adds a pass over
    // its partial tiles (written once, read once at ~4 TB/s)
    const double t_full = (double)pixels / 32.0 * 1.95;
    const double t_split = (double)(nbatch > 1 ? nbatch : 1) * M * Cb * 8.0 / 4e6;
    for (long sp = 1; sp <= max_splits; ++sp) {
        const double rounds = (double)((tiles * sp + 511) / 512);
        const double cost = rounds / (double)sp * t_full + sp * t_split;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
    }
    long pps = (pixels + best - 1) / best;
    pps = (pps + 31) / 32 * 32;
    long splits = (pixels + pps - 1) / pps;
    if (splits < 1) splits = 1;
    *pix_per_split = (int)pps;
    return (int)splits;
}

template <int BM, int BN, bool FAST>
static int launch_wgrad(const WGradParams& p, int vecA, int vecB, hipStream_t s) {
    using C = Cfg<BM, BN>;
    const int M = p.ntaps * p.Ca;
    dim3 grid((M + BM - 1) / BM, (p.Cb + BN - 1) / BN, p.splits * (p.nbatch > 1 ? p.nbatch : 1));
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)wgrad_mfma_kernel<BM, BN, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::smem_wgrad);
        (void)hipFuncSetAttribute((const void*)wgrad_mfma_kernel<BM, BN, FAST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::smem_wgrad);
        return true;
    }();
    (void)attr_set;
    // p.x6 (the caller's arithmetic: AUTO / X6 with ss_config x6 = 1): six exact bf16 piece products instead of fp32 MFMA instructions
    if (p.x6 && ss_tuning().wgrad_mfma_x6) {
        SsProfScope prof(BN == 128 ? "wgrad_mfma_kernel<128,128,x6>" : (BN == 64 ? "wgrad_mfma_kernel<128,64,x6>" : "wgrad_mfma_kernel<128,32,x6>"),
                         2.0 * M * p.Cb * (double)p.N * p.GH * p.GW * (p.nbatch > 1 ? p.nbatch : 1) * 6,
                         4.0 * ((double)p.N * p.AH * p.AW * p.Ca + (double)p.N * p.GH * p.GW * p.Cb) * (p.nbatch > 1 ? p.nbatch : 1), s);
        hipLaunchKernelGGL((wgrad_mfma_kernel<BM, BN, FAST, true>), grid, dim3(256), C::smem_wgrad, s, p, vecA, vecB);
    } else {
        hipLaunchKernelGGL((wgrad_mfma_kernel<BM, BN, FAST>), grid, dim3(256), C::smem_wgrad, s, p, vecA, vecB);
    }
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_launch_wgrad_mfma(const WGradParams& p, float* dw, int ldw, int accumulate, hipStream_t s) {
    return ss_launch_wgrad_mfma_rows(p, dw, ldw, accumulate, p.ntaps * p.Ca, s);
}

int ss_launch_wgrad_mfma_rows(const WGradParams& p, float* dw, int ldw, int accumulate, int rows, hipStream_t s) {
    const long total = (long)p.ntaps * p.Ca * p.Cb;
    if (total == 0) return SS_OK;
    int rc = ss_launch_wgrad_mfma_partials(p, s);
    if (rc != SS_OK) return rc;
    launch_wgrad_reduce_any(p, dw, ldw, accumulate, rows, total, s);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// dw[woff_t + ca*ldw + cb] (+)= sum over p.splits partials part[split][(t,ca)][cb] (fixed order)
int ss_launch_wgrad_reduce(const WGradParams& p, float* dw, int ldw, int accumulate, int rows, hipStream_t s) {
    const long total = (long)rows * p.Cb;
    launch_wgrad_reduce_any(p, dw, ldw, accumulate, rows, total, s);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_launch_wgrad_mfma_partials(const WGradParams& p, hipStream_t s) {
    if (p.x6 && ss_wgrad_x6_ok(p)) return ss_launch_wgrad_x6_partials(p, s);
    {
        const int vecA = (p.Ca % 4 == 0) && (p.a_cs % 4 == 0) && (((uintptr_t)p.a & 15) == 0);
        const int vecB = (p.b_cs % 4 == 0) && (((uintptr_t)p.b & 15) == 0);
        int rc;
        // FAST = incremental pixel coordinates + deferred zero-selects; vector or (odd channel counts) scalar element loads
        const bool fast = p.GW >= 32 && p.pix_per_split % 32 == 0 && ss_tuning().gconv_fast &&
                          (long)p.N * p.AH * p.AW * p.a_cs < (1L << 31) && (long)p.N * p.GH * p.GW * p.b_cs < (1L << 31);
        if (fast) {
            if (p.Cb > 64) rc = launch_wgrad<128, 128, true>(p, vecA, vecB, s);
            else if (p.Cb > 32) rc = launch_wgrad<128, 64, true>(p, vecA, vecB, s);
            else rc = launch_wgrad<128, 32, true>(p, vecA, vecB, s);
        } else if (p.Cb > 64) rc = launch_wgrad<128, 128, false>(p, vecA, vecB, s);
        else if (p.Cb > 32) rc = launch_wgrad<128, 64, false>(p, vecA, vecB, s);
        else rc = launch_wgrad<128, 32, false>(p, vecA, vecB, s);
        if (rc != SS_OK) return rc;
    }
    return SS_OK;
}
