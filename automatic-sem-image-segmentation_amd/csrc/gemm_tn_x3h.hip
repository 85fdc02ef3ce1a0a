// Batched "TN" GEMM with the x3h arithmetic on operands that are ALREADY split into two fp16 planes, K-MAJOR:
//     C[batch][split][m][n] = sum_{k in split} A[batch][k][m] * B[batch][k][n],      a*b = ah*bh + 2^-11 (ah*bl + al*bh)
// This is the Winograd-domain weight gradient (conv_wino.hip): k = tile, m = input channel, n = output channel -- the reduction
// index is the SLOW index of both operands ([xi][tile][channel] planes written by the input / dy transforms), which is what made
// the register-staged kernel (conv_mfma_x6.hip wgrad_x6_kernel: transposing LDS stores + in-kernel split) VALU-bound.  Here the
// stage is the operands' natural image, rows of k with the channels contiguous, moved global -> LDS by LDS-DMA with no register
// pass, and the MFMA fragments come out of it with gfx950's TRANSPOSING LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane
// group column i of the 4 (k) x 16 (channel) block whose rows the group's lanes address (lane j: row j/4, columns 4 (j%4) .. +3)
// -- measured with tools/tr_probe.hip.  Two such reads give a lane the 8 k-values of its channel that v_mfma_f32_32x32x16_f16
// wants; WHICH eight k they are is free as long as both operands use the same assignment (the product sums over k).
//
// Tile 256 (M) x 128 (N) x 32 (K), 512 threads = 8 waves as 4 (M) x 2 (N), 64x64 per wave; stage = 2 planes x (32 x 256 + 32 x 128)
// halfs = 48 KiB, ring of three stages (two tiles in flight behind the one being consumed), ONE barrier per K step -- the
// schedule of gemm_x6p.hip.
// LDS image of a plane: [32 k][BM or BN channels] fp16, row = 512 B (A) / 256 B (B).  The 32 lanes of one LDS cycle of the
// transposing read touch 4 consecutive k rows x 64 B: the rows would fall on the same banks, so the 64-byte segments of a row are
// XOR-swizzled with (k & 3), on the DMA source address and on the read address alike.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int TBM = 256, TBN = 128, TBK = 32;
constexpr int TA_ROW = TBM * 2, TB_ROW = TBN * 2;            // bytes per k row
constexpr int TA_PLANE = TBK * TA_ROW, TB_PLANE = TBK * TB_ROW;      // 16 KiB, 8 KiB
// NPL = operand planes read: 2 = (h, l) with the three products of the x3h arithmetic; 1 = the leading plane alone, one product (16-bit
// activation storage under wino16_products = 1: the operand rounding of the forward pass's one-plane GEMMs; the l planes are ignored)
template <int NPL> struct TNC {
    static constexpr int STAGE = NPL * (TA_PLANE + TB_PLANE);          // 48 / 24 KiB
    static constexpr int STAGES = NPL == 2 ? 3 : 5;                     // 144 / 120 KiB in flight or in use
    static constexpr int NDMA = STAGE / 1024 / 8;                       // 6 / 3 LDS-DMA instructions per wave and K step
};

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ s16x4 tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// ILV: the transposing fragment reads between the MFMAs instead of in front of them (as gemm_x6p.hip) -- measured SLOWER here (16 reads per
// half step: 378 vs 333 us at n = 16, 215 vs 194 at n = 8, tools/tn_probe.py): kept as a measurement switch only, the launcher uses false.
// SCALAR_EPI (the default): the one-column-per-lane epilogue (64 four-byte stores per lane); false = the register-transposed one.
template <bool ILV, bool SCALAR_EPI = true, int NPL = 2>
__global__ __launch_bounds__(512, 1) void gemm_tn_x3h_kernel(TNParams p) {
    static_assert(NPL == 2 || !ILV, "the interleaved form exists for the two-plane kernel only");
    constexpr int TSTAGE = TNC<NPL>::STAGE, TSTAGES = TNC<NPL>::STAGES, TNDMA = TNC<NPL>::NDMA;
    constexpr int NQ = NPL == 2 ? 3 : 1;          // MFMA groups per K half
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int gridM = p.M / TBM, gridN = p.N / TBN;
    const int per = gridM * gridN;
    // XCD-aware order (speed only): a contiguous chunk of the tile space per XCD; the splits of one (batch, m, n) tile and the n
    // tiles of one m tile (same A rows) are neighbours.  PERSISTENT workgroups as in gemm_x6p.hip: with more tiles than workgroups
    // a workgroup walks its XCD's chunk and its operand stream runs on across the tile boundaries.
    const int total = per * p.nbatch * p.splits;
    int t_first, t_end, t_stride;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = total >> 3, r = total & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        if (nwg == total) { t_first = start + slot; t_end = t_first + 1; t_stride = 1; }
        else { t_first = start + slot; t_end = start + q + (xcd < r ? 1 : 0); t_stride = nwg >> 3; }
    }
    // per-tile set-up.  This wave's share of a stage: 6 pieces of 1 KiB.  Pieces 0..15 / 16..31: A plane h / l (2 k rows each),
    // 32..39 / 40..47: B plane h / l (4 k rows each).  Lane j of a piece lands on bytes [16 j, 16 j + 16) of the piece.
    struct TileCtx { int m0, n0, nchunks; long cbase; };
    auto setup = [&](int t, const unsigned short* (&g)[TNDMA]) -> TileCtx {
        const int bs = t / per;
        const int tl = t - bs * per;
        const int batch = bs / p.splits, split = bs - batch * p.splits;
        const int m0 = (tl / gridN) * TBM, n0 = (tl % gridN) * TBN;
        const int k_begin = split * p.k_per_split;
        const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
#pragma unroll
        for (int j = 0; j < TNDMA; ++j) {
            const int q = wave * TNDMA + j;
            if (q < NPL * 16) {
                const int pl = q >> 4, pc = q & 15;
                const int row = 2 * pc + (lane >> 5);
                const int pseg = (lane & 31) >> 2;                          // physical 64-byte segment of the row
                const int col = ((pseg ^ (row & 3)) * 64 + (lane & 3) * 16) / 2;      // logical channel offset
                g[j] = p.a + pl * p.a_plane + batch * p.a_bs + (long)(k_begin + row) * p.lda + m0 + col;
            } else {
                const int q2 = q - NPL * 16;
                const int pl = q2 >> 3, pc = q2 & 7;
                const int row = 4 * pc + (lane >> 4);
                const int pseg = (lane & 15) >> 2;
                const int col = ((pseg ^ (row & 3)) * 64 + (lane & 3) * 16) / 2;
                g[j] = p.b + pl * p.b_plane + batch * p.b_bs + (long)(k_begin + row) * p.ldb + n0 + col;
            }
        }
        return TileCtx{m0, n0, (k_end - k_begin) / TBK, ((long)batch * p.splits + split) * p.M * p.N};
    };
    int loff[TNDMA];
#pragma unroll
    for (int j = 0; j < TNDMA; ++j) {
        const int q = wave * TNDMA + j;
        loff[j] = q < NPL * 16 ? (q >> 4) * TA_PLANE + (q & 15) * 1024 : NPL * TA_PLANE + ((q - NPL * 16) >> 3) * TB_PLANE + ((q - NPL * 16) & 7) * 1024;
    }
    const unsigned short* gp[TNDMA];
    const unsigned short* gpn[TNDMA];
    if (t_first >= t_end) return;
    TileCtx cur = setup(t_first, gp);

    f32x16 acc[NPL][2][2];        // [0]: h*h, [1]: the cross terms (they carry the factor 2^-11)

    // fragment read addresses.  Lane l, read rr (0 / 1), K half kh: k row = 16 kh + 8 (l >> 5) + 4 rr + ((l & 15) >> 2),
    // channel = (wave tile) + 32 mi + 16 ((l >> 4) & 1) + 4 (l & 3).  k & 3 = (l & 15) >> 2 for every read: ONE swizzle term per lane.
    const int kr = 8 * lh + ((lane & 15) >> 2);            // + 16 kh + 4 rr
    const int sw = (lane & 15) >> 2;
    const int ca = wm * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // + 32 mi
    const int cb = wn * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    int oa[2], ob[2];            // byte offsets inside a plane for mi / ni = 0, 1 at k row kr
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a2 = (ca + 32 * i) * 2, b2 = (cb + 32 * i) * 2;
        oa[i] = kr * TA_ROW + (((a2 >> 6) ^ sw) << 6) + (a2 & 63);
        ob[i] = kr * TB_ROW + (((b2 >> 6) ^ sw) << 6) + (b2 & 63);
    }

    f16x8 a0[NPL][2], b0[NPL][2], a1[NPL][2], b1[NPL][2];          // [plane][mi / ni]
    auto frag = [&](f16x8 (&a)[NPL][2], f16x8 (&b)[NPL][2], int stage, int kh) {
        const unsigned char* sa = lds + stage * TSTAGE + kh * 16 * TA_ROW;
        const unsigned char* sb = lds + stage * TSTAGE + NPL * TA_PLANE + kh * 16 * TB_ROW;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const s16x4 x0 = tr4(sa + pl * TA_PLANE + oa[i]), x1 = tr4(sa + pl * TA_PLANE + oa[i] + 4 * TA_ROW);
                a[pl][i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                const s16x4 y0 = tr4(sb + pl * TB_PLANE + ob[i]), y1 = tr4(sb + pl * TB_PLANE + ob[i] + 4 * TB_ROW);
                b[pl][i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
    };
    auto frag_plane = [&](f16x8 (&a)[NPL][2], f16x8 (&b)[NPL][2], int stage, int kh, int pl) {
        const unsigned char* sa = lds + stage * TSTAGE + kh * 16 * TA_ROW;
        const unsigned char* sb = lds + stage * TSTAGE + NPL * TA_PLANE + kh * 16 * TB_ROW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const s16x4 x0 = tr4(sa + pl * TA_PLANE + oa[i]), x1 = tr4(sa + pl * TA_PLANE + oa[i] + 4 * TA_ROW);
            a[pl][i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
            const s16x4 y0 = tr4(sb + pl * TB_PLANE + ob[i]), y1 = tr4(sb + pl * TB_PLANE + ob[i] + 4 * TB_ROW);
            b[pl][i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    // l*h, h*l (cross accumulators), h*h; consecutive MFMAs go to different accumulators
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0}, HS[3] = {1, 1, 0};
    auto mma4 = [&](f16x8 (&a)[NPL][2], f16x8 (&b)[NPL][2], int q) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (NPL == 1) acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], b[0][ni], acc[0][mi][ni], 0, 0, 0);
                else acc[HS[q]][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], acc[HS[q]][mi][ni], 0, 0, 0);
            }
    };

    long kstep[TNDMA];          // elements per K step of the operand a piece belongs to
#pragma unroll
    for (int j = 0; j < TNDMA; ++j) kstep[j] = (long)TBK * (wave * TNDMA + j < NPL * 16 ? p.lda : p.ldb);

    // scales: the planes hold a * 2^(14 - ea) and b * 2^(14 - eb) (conv_wino.hip, same slots, same bounds)
    // a_prescaled: the A planes carry per-row scales that the producer of B folded into B's rows (ea = 14: no factor left), and B's
    // exponent counts the second slot (the maximum of the tensor A was transformed from) -- the same expression as wino_dy_kernel's
    const int ea = p.a_prescaled ? 14 : ss_amax_exp(__uint_as_float(ss_amax_load(p.amax_a, p.stripes_a))) + p.bound_a;
    const int eb = ss_amax_exp(__uint_as_float(ss_amax_load(p.amax_b, p.stripes_b))) + p.bound_b +
                   (p.amax_b2 ? ss_amax_exp(__uint_as_float(ss_amax_load(p.amax_b2, p.stripes_b2))) : 0);
    const float out_scale = ldexpf(1.f, ea - 14 + eb - 14);

    {   // pipeline fill for this workgroup's first tile (chunk indices past the end re-fetch the last chunk: fixed group count)
        const int nchunks = cur.nchunks;
#pragma unroll
        for (int t = 0; t < TSTAGES; ++t) {
            const int tt = t < nchunks ? t : nchunks - 1;
#pragma unroll
            for (int j = 0; j < TNDMA; ++j) dma16(gp[j] + tt * kstep[j], lds + t * TSTAGE + loff[j]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TSTAGES - 1) * TNDMA) : "memory");
    }
    int st = 0;
    for (int t = t_first; t < t_end; t += t_stride) {
        const int nchunks = cur.nchunks;
        const bool more = t + t_stride < t_end;
        TileCtx nxt = cur;
        if (more) nxt = setup(t + t_stride, gpn);
#pragma unroll
        for (int s = 0; s < NPL; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[s][mi][ni][r] = 0.f;
        __builtin_amdgcn_s_barrier();
        frag(a0, b0, st, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)

        for (int c = 0; c < nchunks; ++c) {
            const int st1 = st + 1 == TSTAGES ? 0 : st + 1;
            frag(a1, b1, st, 1);
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) mma4(a0, b0, q);
            if constexpr (ILV) {          // the 16 transposing reads of the other K half between this half's MFMAs instead of in a burst in front of them (gemm_x6p.hip ILV)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_assert((TSTAGES - 2) * TNDMA <= 15, "low vmcnt bits only");
            __builtin_amdgcn_s_waitcnt(0x0070 | (((TSTAGES - 2) * TNDMA) & 15));      // vmcnt(6 / 9) lgkmcnt(0): chunk c+1 landed, own reads of chunk c done
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!ILV) frag(a0, b0, st1, 0);
            // past the last chunk of a tile the stream continues with the next tile's first chunks (last tile: re-fetch, unused)
            const int ca = c + TSTAGES;
            const bool own = ca < nchunks;
            const int cn = own ? ca : (more ? ca - nchunks : nchunks - 1);
            unsigned char* dst = lds + st * TSTAGE;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if constexpr (ILV) { if (q < 2) frag_plane(a0, b0, st1, 0, q); }          // plane q of the next step's first half: 8 reads behind this group's MFMAs
                mma4(a1, b1, q);
#pragma unroll
                for (int j = 0; j < TNDMA; ++j)
                    if (j * NQ / TNDMA == q) dma16(((own || !more) ? gp[j] : gpn[j]) + cn * kstep[j], dst + loff[j]);
                if constexpr (ILV) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (q < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, TNDMA / 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            st = st1;
        }

        // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Register-transposed (ss_quad_transpose,
        // as gemm_x6p.hip): the four registers of a row quad become four consecutive columns of one row = one 16-byte store; 16 store
        // instructions per wave (8 rows x 128 B each) instead of 64 four-byte ones.  Tiles are whole (M % 256 == 0, N % 128 == 0).
        if constexpr (SCALAR_EPI) {
            float* cbase = p.c + cur.cbase + (cur.n0 + wn * 64 + l31);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = cur.m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float* crow = cbase + (long)m * p.N;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) crow[32 * ni] = (NPL == 1 ? acc[0][mi][ni][r] : fmaf(acc[NPL - 1][mi][ni][r], 4.8828125e-4f, acc[0][mi][ni][r])) * out_scale;
                }
            }
        } else {
            const bool odd = lane & 1, hi = lane & 2;
            const int m = cur.m0 + wm * 64 + 4 * lh + (lane & 3);          // + 32 mi + 8 rq
            float* g = p.c + cur.cbase + (cur.n0 + wn * 64 + (l31 & ~3)) + (long)m * p.N;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float* grow = g + (long)(mi * 32 + 8 * rq) * p.N;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float v[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) v[rr] = (NPL == 1 ? acc[0][mi][ni][rq * 4 + rr] : fmaf(acc[NPL - 1][mi][ni][rq * 4 + rr], 4.8828125e-4f, acc[0][mi][ni][rq * 4 + rr])) * out_scale;
                        *(f32x4*)(grow + 32 * ni) = ss_quad_transpose(v[0], v[1], v[2], v[3], odd, hi);
                    }
                }
            }
        }
        if (more) {
            asm volatile("s_waitcnt vmcnt(63)" ::: "memory");      // the next tile's first chunk groups are older than the stores
#pragma unroll
            for (int j = 0; j < TNDMA; ++j) gp[j] = gpn[j];
            cur = nxt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

bool ss_gemm_tn_x3h_ok(int M, int N, long K) { return M % TBM == 0 && N % TBN == 0 && K % TBK == 0 && K >= TBK; }

// CUs the persistent workgroups of this launch will occupy (the launcher's rule)
static int tn_cus() {
    static const int n_cu = [] { int v = 0; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0); return v >= 8 ? v / 8 * 8 : 256; }();
    return (ss_tuning().gemm_cus >= 8 && ss_tuning().gemm_cus < n_cu) ? ss_tuning().gemm_cus / 8 * 8 : n_cu;
}

// the largest split count ss_gemm_tn_splits can return for this problem, whatever the CU count: what a workspace is sized for
int ss_gemm_tn_splits_max(int M, int N, long K, int nbatch) {
    (void)M; (void)N; (void)nbatch;
    long m = K / TBK / 8;          // at least 8 K steps per split
    return (int)(m < 1 ? 1 : (m > 32 ? 32 : m));
}

// Splits of the K range.  The (tile, split) units are dealt to `cus` persistent workgroups: the launch takes
//     ceil(units / cus) x (K steps per unit + ~5 steps of pipeline fill and epilogue)   + ~1 step per split for the partials' reduction,
// so the count is chosen to make the units few, whole rounds (round 6; before: "enough units for ~4 rounds").  The trunk's weight
// gradients of the dual-chain step are 512 x 512 x (2048 | 4096) x 36 problems on 192 CUs: 288 tiles x 4 splits = six rounds of 32 steps
// became 288 x 2 = three rounds of 64 (-1.2 ... -1.6 ms per step in-process; forced counts 2 / 4 / 7 / 15: 144.6 / 146.5 / 154.1 / 159.1 ms
// for the CycleGAN step -- the model's ~5 steps of fill per unit is what that scan shows).  ss_config gemm_tn_rounds: 0 = the earlier
// rule, n > 1 = n splits (measurement)
int ss_gemm_tn_splits(int M, int N, long K, int nbatch, int* k_per_split) {
    const long tiles = (long)(M / TBM) * (N / TBN) * nbatch;
    const long steps = K / TBK;
    const int smax = ss_gemm_tn_splits_max(M, N, K, nbatch);
    const long cus = tn_cus();
    if (!ss_tuning().gemm_tn_rounds) {          // (measurement: the rule of rounds 2 - 5 -- units for ~4 rounds, >= 8 steps each)
        long sp = (1024 + tiles - 1) / tiles;
        if (sp > steps / 8) sp = steps / 8;
        if (sp < 1) sp = 1;
        if (sp > smax) sp = smax;
        long per = (steps + sp - 1) / sp;
        sp = (steps + per - 1) / per;
        *k_per_split = (int)(per * TBK);
        return (int)sp;
    }
    long best = 1, best_per = steps;
    double best_cost = 1e30;
    if (ss_tuning().gemm_tn_rounds > 1) {          // (measurement: a forced split count)
        long sp = ss_tuning().gemm_tn_rounds > smax ? smax : ss_tuning().gemm_tn_rounds;
        const long per = (steps + sp - 1) / sp;
        *k_per_split = (int)(per * TBK);
        return (int)((steps + per - 1) / per);
    }
    for (long sp = 1; sp <= smax; ++sp) {
        const long per = (steps + sp - 1) / sp;
        const long spe = (steps + per - 1) / per;          // the splits that are not empty
        if (spe != sp) continue;
        const long rounds = (tiles * spe + cus - 1) / cus;
        // (an uneven split -- the last one shorter -- gives up the persistent walk, whose operand stream runs on across tile boundaries)
        const double cost = ((double)rounds * (double)(per + 5) + (double)spe) * (steps % spe == 0 ? 1.0 : 1.06);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = spe; best_per = per; }
    }
    *k_per_split = (int)(best_per * TBK);
    return (int)best;
}

int ss_launch_gemm_tn_x3h(const TNParams& p, hipStream_t s) {
    if (!ss_gemm_tn_x3h_ok(p.M, p.N, p.K) || p.k_per_split % TBK || p.splits < 1 || p.lda % 8 || p.ldb % 8) return SS_ERR_UNSUPPORTED;
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gemm_tn_x3h_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_x3h_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_x3h_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_tn_x3h_kernel<false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const long tiles = (long)(p.M / TBM) * (p.N / TBN) * p.nbatch * p.splits;
    static const int n_cu = [] { int v = 0; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0); return v >= 8 ? v / 8 * 8 : 256; }();
    const int cus = (ss_tuning().gemm_cus >= 8 && ss_tuning().gemm_cus < n_cu) ? ss_tuning().gemm_cus / 8 * 8 : n_cu;
    const bool persistent = ss_tuning().gemm_persistent && tiles > cus && p.k_per_split >= 3 * TBK && p.K % p.k_per_split == 0;
    const long nwg = persistent ? cus : tiles;
    const int npl = p.one_plane ? 1 : 2;
    SsProfScope prof(p.one_plane ? "gemm_tn_x3h_kernel<1 plane>" : "gemm_tn_x3h_kernel", 2.0 * p.M * p.N * p.K * p.nbatch * (p.one_plane ? 1 : 3),
                     2.0 * npl * ((double)p.M + p.N) * p.K * p.nbatch + 4.0 * p.M * p.N * p.nbatch * p.splits, s);
    // default: burst fragment reads + the one-column-per-lane epilogue -- the two variants of gemm_x6p.hip both measure slower here
    // (tools/tn_probe.py, n = 8 / 16: interleaved reads 218 / 381 us, register-transposed epilogue 194 / 339 us, this form 188 - 193 /
    // 332 - 333 us).  tile_dbg (measurement): 4 = interleaved reads, 16 = register-transposed epilogue; all three are bit-identical
    constexpr int LDS2 = TNC<2>::STAGES * TNC<2>::STAGE, LDS1 = TNC<1>::STAGES * TNC<1>::STAGE;
    if (p.one_plane) hipLaunchKernelGGL((gemm_tn_x3h_kernel<false, true, 1>), dim3((unsigned)nwg), dim3(512), LDS1, s, p);
    else if (ss_tuning().tile_dbg & 4) hipLaunchKernelGGL((gemm_tn_x3h_kernel<true, true>), dim3((unsigned)nwg), dim3(512), LDS2, s, p);
    else if (ss_tuning().tile_dbg & 16) hipLaunchKernelGGL((gemm_tn_x3h_kernel<false, false>), dim3((unsigned)nwg), dim3(512), LDS2, s, p);
    else hipLaunchKernelGGL((gemm_tn_x3h_kernel<false, true>), dim3((unsigned)nwg), dim3(512), LDS2, s, p);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
