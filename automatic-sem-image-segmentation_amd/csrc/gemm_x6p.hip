// Batched "NT" GEMM with the x6 arithmetic (conv_mfma_x6.hip) on operands that are ALREADY split into three bf16 planes:
//     C[batch][split][m][n] = sum_{k in split} A[batch][m][k] * B[batch][n][k],      a*b = ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm
// Both operands are K-contiguous rows of bf16 (three planes each), produced once by the kernel that writes the operand (the
// Winograd input / weight / dy transforms), so the K loop carries no conversion work: every tile goes global -> LDS by
// LDS-DMA (global_load_lds_dwordx4, no staging registers, no ds_write pass) into a double-buffered stage while the matrix cores
// work on the other stage; ONE barrier per K step.
//
// Tile 256 (M) x 128 (N) x 32 (K), 512 threads = 8 waves as 4 (M) x 2 (N), 64x64 per wave = 2x2 MFMA tiles of 32x32.
// LDS stage: 3 planes x (256 + 128) rows x 64 B = 72 KiB, two stages = 144 KiB (one workgroup per CU, two waves per SIMD).
// An LDS-DMA instruction writes lane-linear (wave base + 16 B x lane) = 16 rows x 4 slots of 16 B.  Rows are 64 B with no padding;
// bank conflicts of the ds_read_b128 operand fetch (lane = row) are avoided by an XOR swizzle of the 16-B slot with bits 2..3 of
// the row, applied on the SOURCE address of the DMA (which k-octet a lane fetches) and on the read address alike.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PBM = SS_X6P_BM, PBN = SS_X6P_BN, PBK = 32;
constexpr int ROWB = PBK * 2;                  // bytes per LDS row
constexpr int A_PLANE_B = PBM * ROWB;          // 16 KiB
constexpr int B_PLANE_B = PBN * ROWB;          //  8 KiB
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NPL> struct Frag;
template <> struct Frag<3> { typedef bf16x8 T; };      // x6:  three bf16 planes, six products
template <> struct Frag<2> { typedef f16x8 T; };       // x3h: two fp16 planes (x = h + 2^-11 l), three products
template <> struct Frag<1> { typedef f16x8 T; };       // 16-bit activation storage: ONE fp16 plane per operand (the leading piece), one product

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

constexpr int x6p_stages(int npl, bool wide = false) { return npl == 1 ? 4 : (npl == 2 && !wide ? 3 : 2); }      // one plane: 24 KiB (wide: 32 KiB) stages

// WIDE (x3h only): 256 x 256 tile, 1024 threads = 16 waves as 4 (M) x 4 (N), still 64 x 64 per wave.  The kernel is paced by its
// operand stream (profiles/r02_f_*: 53 % issue stalls behind the LDS-DMA queue, matrix pipe 33 % busy), and a 256 x 128 tile moves
// 1.5 bytes per output element and K step; 256 x 256 moves 1.0.  Registers for it (four waves per SIMD: 128 VGPRs) come from ONE
// accumulator set -- the operand planes then carry the low piece at its own magnitude (x*s = h + l, as in the gather kernels;
// p.plain_l) instead of 2^11 times it -- and one fragment register set (the four resident waves cover the LDS latency); two LDS
// stages of 64 KiB.
// ILV (x3h, not wide): the fragment reads of a half step are issued BETWEEN its MFMAs (sched_group_barrier: one MFMA, one ds_read,
// ...) instead of in a burst in front of them -- after a barrier all eight waves of the CU issue their eight reads at once, the LDS
// queue backs up and the first MFMA of every wave waits behind its own reads (SQ counters, tools/x6p_sq.sh: 693 wait cycles and
// ~630 idle matrix-pipe cycles per K step even with no global memory traffic at all)
template <int NPL, bool WIDE, bool ILV = false>
__global__ __launch_bounds__(WIDE ? 1024 : 512, 1) void gemm_x6p_kernel(X6PParams p) {
    static_assert(!WIDE || NPL <= 2, "the wide tile exists for the fp16 operands (two planes with the plain low piece, or the one plane of 16-bit storage)");
    static_assert(NPL >= 1 && NPL <= 3, "one, two or three operand planes");
    constexpr int PBN = WIDE ? 256 : SS_X6P_BN;
    constexpr int B_PLANE_B = PBN * ROWB;
    constexpr int NWV = WIDE ? 16 : 8;             // waves per workgroup
    constexpr int WN = PBN / 64;                   // waves along N
    constexpr int STAGE_B = NPL * (A_PLANE_B + B_PLANE_B);
    constexpr int NDMA = STAGE_B / 1024 / NWV;     // LDS-DMA instructions per wave and K step (9 / 6; wide: 4)
    // Ring of LDS stages.  The K loop is paced by the operand stream, not by the matrix pipe: one stage in flight (two stages) is
    // 48 KiB per CU against ~2 us of loaded HBM / Infinity-Cache latency = 6 TB/s chip-wide, measured 5.96.  x3h stages are 48 KiB,
    // so three fit into the 160 KiB: two tiles in flight while the third is consumed.
    constexpr int STAGES = x6p_stages(NPL, WIDE);
    typedef typename Frag<NPL>::T FT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int gridM = (p.M + PBM - 1) / PBM, gridN = (p.N + PBN - 1) / PBN;
    const int per = gridM * gridN;
    // XCD-aware order (speed only): a contiguous chunk of the tile space per XCD, N fastest.  PERSISTENT workgroups: when the grid
    // is smaller than the tile count (one workgroup per CU, see the launcher) a workgroup walks its XCD's chunk with stride
    // (workgroups per XCD), so that the workgroups of an XCD keep working on neighbouring tiles, and its operand stream runs on
    // across the tile boundary: the last K steps of a tile already fetch the first chunks of the next one, the epilogue stores of
    // a tile overlap with that fetch, and no pipeline fill / workgroup launch is paid per tile.
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
    const int rl = lane >> 2, sl = lane & 3;
    // per-tile set-up: this wave's share of a stage = NDMA pieces of 16 rows (per plane: A 16 pieces, B 8); m0 / n0 / output base
    struct TileCtx { int m0, n0, nchunks; long cbase; };
    auto setup = [&](int t, const unsigned short* (&g)[NDMA]) -> TileCtx {
        const int bs = t / per;
        const int tl = t - bs * per;
        const int batch = bs / p.splits, split = bs - batch * p.splits;
        const int m0 = (tl / gridN) * PBM, n0 = (tl % gridN) * PBN;
        const int k_begin = split * p.k_per_split;
        const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int q = wave * NDMA + j;
            if (q < NPL * (PBM / 16)) {
                const int pl = q / (PBM / 16), rb = q % (PBM / 16);
                const int row = rb * 16 + rl;
                const int ko = sl ^ ((row >> 2) & 3);
                g[j] = p.a + pl * p.a_plane + batch * p.a_bs + (long)(m0 + row) * p.lda + k_begin + 8 * ko;
            } else {
                const int q2 = q - NPL * (PBM / 16);
                const int pl = q2 / (PBN / 16), rb = q2 % (PBN / 16);
                const int row = rb * 16 + rl;
                const int ko = sl ^ ((row >> 2) & 3);
                g[j] = p.b + pl * p.b_plane + batch * p.b_bs + (long)(n0 + row) * p.ldb + k_begin + 8 * ko;
            }
        }
        return TileCtx{m0, n0, (k_end - k_begin) / PBK, (long)batch * p.c_bs + (long)split * p.c_ss};
    };
    int loff[NDMA];
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int q = wave * NDMA + j;
        if (q < NPL * (PBM / 16)) loff[j] = (q / (PBM / 16)) * A_PLANE_B + (q % (PBM / 16)) * 1024;
        else { const int q2 = q - NPL * (PBM / 16); loff[j] = NPL * A_PLANE_B + (q2 / (PBN / 16)) * B_PLANE_B + (q2 % (PBN / 16)) * 1024; }
    }
    const unsigned short* gp[NDMA];
    const unsigned short* gpn[NDMA];
    unsigned skipmask = 0;          // measurement only (p.dbg)
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const bool isA = wave * NDMA + j < NPL * (PBM / 16);
        if ((isA && (p.dbg & 128)) || (!isA && (p.dbg & 64))) skipmask |= 1u << j;
    }
    if (t_first >= t_end) return;
    TileCtx cur = setup(t_first, gp);
    constexpr int NACC = NPL == 2 && !WIDE ? 2 : 1;          // x3h: the cross terms h*l + l*h accumulate apart (they carry the factor 2^-11); wide: plain l, one set
    f32x16 acc[NACC][2][2];

    // operand fetch addresses (stage 0): row = (wave tile base) + l31, slot = (lh + 2*ks) ^ ((row >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    const int so0 = ((lh ^ sw) << 4), so1 = so0 ^ 32;
    const unsigned char* fa = lds + (wm * 64 + l31) * ROWB;
    const unsigned char* fb = lds + NPL * A_PLANE_B + (wn * 64 + l31) * ROWB;

    // Software pipeline, ONE barrier per K step.  Two fragment register sets: F0 = (tile c, k 0..15), F1 = (tile c, k 16..31).
    //   top of step c:  F0 holds tile c / half 0 (read after the previous barrier);  stage (c+1)&1 is being filled by DMA
    //     read F1 <- tile c half 1          (in flight under the next MFMAs)
    //     24 MFMAs on F0
    //     vmcnt(0) + barrier                -> tile c+1 visible to everyone, nobody reads tile c's stage any more
    //     read F0 <- tile c+1 half 0        (in flight under the next MFMAs)
    //     24 MFMAs on F1, the 9 DMAs of tile c+2 (into tile c's stage) issued between them
    FT a0[NPL][2], b0[NPL][2], a1[NPL][2], b1[NPL][2];
    auto frag = [&](FT (&a)[NPL][2], FT (&b)[NPL][2], int stage, int so) {
        const int sb = stage * STAGE_B;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[pl][mi] = *(const FT*)(fa + sb + pl * A_PLANE_B + mi * 32 * ROWB + so);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[pl][ni] = *(const FT*)(fb + sb + pl * B_PLANE_B + ni * 32 * ROWB + so);
        }
    };
    // x6: six products, smallest terms first;  x3h: l*h, h*l (cross accumulators), h*h.  Consecutive MFMAs go to different accumulators
    constexpr int NQ = NPL == 3 ? 6 : (NPL == 2 ? 3 : 1);
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0}, HS[3] = {WIDE ? 0 : 1, WIDE ? 0 : 1, 0};
    auto mma4 = [&](FT (&a)[NPL][2], FT (&b)[NPL][2], int q) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (NPL == 3)
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][mi], b[PB[q]][ni], acc[0][mi][ni], 0, 0, 0);
                else if constexpr (NPL == 1)
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][mi], b[0][ni], acc[0][mi][ni], 0, 0, 0);
                else
                    acc[HS[q]][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], acc[HS[q]][mi][ni], 0, 0, 0);
            }
    };

    // pipeline fill for the first tile of this workgroup (chunk indices past the end re-fetch the last chunk: fixed group count)
    {
        const int nchunks = cur.nchunks;
#pragma unroll
        for (int t = 0; t < STAGES; ++t) {
            const long goff = (long)(t < nchunks ? t : nchunks - 1) * PBK;
#pragma unroll
            for (int j = 0; j < NDMA; ++j) dma16(gp[j] + goff, lds + t * STAGE_B + loff[j]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * NDMA) : "memory");      // chunk 0 landed, the others may be in flight
    }
    constexpr int NST = 16;          // store instructions of the coalesced epilogue per wave: 2 (mi) x 4 (row groups) x 2
    bool relaxed = false;            // the previous tile of this workgroup ended with exactly NST stores in THIS wave
    int st = 0;
    for (int t = t_first; t < t_end; t += t_stride) {
        const int nchunks = cur.nchunks;
        const bool more = t + t_stride < t_end;
        TileCtx nxt = cur;
        if (more) nxt = setup(t + t_stride, gpn);
#pragma unroll
        for (int s = 0; s < NACC; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[s][mi][ni][r] = 0.f;
        __builtin_amdgcn_s_barrier();
        frag(a0, b0, st, so0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)

        if constexpr (WIDE) {
            // one fragment set: half 0 (read after the previous barrier) -> 12 MFMAs -> half 1 into the same registers -> 12 MFMAs ->
            // chunk c+1 landed + barrier -> its half 0, and the DMA of chunk c+2 into the stage just released
            // (ring of STAGES: the STAGES - 2 younger chunk groups may still be in flight; the epilogue's stores are older than all of them)
            constexpr int VW = (STAGES - 2) * NDMA;
            static_assert(VW <= 63, "vmcnt is a 6-bit field");
            for (int c = 0; c < nchunks; ++c) {
                const int st1 = st + 1 == STAGES ? 0 : st + 1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) mma4(a0, b0, q);
                __builtin_amdgcn_sched_barrier(0);
                frag(a0, b0, st, so1);
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) mma4(a0, b0, q);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0070 | (VW & 15) | ((VW >> 4) << 14));      // vmcnt(VW) lgkmcnt(0): chunk c+1 landed
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                frag(a0, b0, st1, so0);
                const int ca = c + STAGES;
                const bool own = ca < nchunks;
                const int cn = own ? ca : (more ? ca - nchunks : nchunks - 1);
                const long goff = (long)cn * PBK;
                unsigned char* dst = lds + st * STAGE_B;
#pragma unroll
                for (int j = 0; j < NDMA; ++j) dma16(((own || !more) ? gp[j] : gpn[j]) + goff, dst + loff[j]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
                st = st1;
            }
        } else {
        // One K step.  VW = how many of this wave's youngest vector-memory operations may still be outstanding when chunk c+1 must
        // have landed: normally the (STAGES - 2) younger chunk groups; in the first STAGES - 1 steps of a tile that follows a tile
        // whose epilogue issued exactly NST stores (see below) those stores are younger than chunk c+1 as well and need not have
        // completed -- vmcnt counts loads and stores in issue order, and with the strict count the K loop sat out the drain of the
        // previous tile's 128 KiB of C (46 of 237 us per launch, tools/x6p_dma_probe.sh)
        auto kstep = [&](int c, auto VW) {
                constexpr int W = decltype(VW)::value;
                const int st1 = st + 1 == STAGES ? 0 : st + 1;
                frag(a1, b1, st, so1);
                if constexpr (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) mma4(a0, b0, q);
                if constexpr (ILV) {
#pragma unroll
                    for (int i = 0; i < 4 * NPL; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // one ds_read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NQ - 4 * NPL, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // chunk c+1 landed (the younger groups may still be in flight) and this wave's reads of chunk c are done: a real
                // s_waitcnt (vmcnt(W) lgkmcnt(0)), so that the compiler's own counting sees it
                __builtin_amdgcn_s_waitcnt(0x0070 | (W & 15) | ((W >> 4) << 14));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // branch-free from here to the loop end (one basic block keeps the compiler's lgkmcnt counting exact): past the last
                // chunk of the LAST tile the fragment read fetches stale LDS and the DMA re-fetches the last chunk into a free stage;
                // past the last chunk of any other tile both continue with the NEXT tile's first chunks
                if constexpr (!ILV) frag(a0, b0, st1, so0);
                const int ca = c + STAGES;
                const bool own = ca < nchunks;
                const int cn = own ? ca : (more ? ca - nchunks : nchunks - 1);
                const long goff = (long)cn * PBK;
                unsigned char* dst = lds + st * STAGE_B;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if constexpr (ILV) {          // plane q of the next half step's fragments: four reads, one behind each MFMA of this group
                        if (q < NPL) {
                            const int sb = st1 * STAGE_B;
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) a0[q][mi] = *(const FT*)(fa + sb + q * A_PLANE_B + mi * 32 * ROWB + so0);
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) b0[q][ni] = *(const FT*)(fb + sb + q * B_PLANE_B + ni * 32 * ROWB + so0);
                        }
                    }
                    mma4(a1, b1, q);
#pragma unroll
                    for (int j = 0; j < NDMA; ++j)
                        if (j * NQ / NDMA == q && !(skipmask >> j & 1)) dma16(((own || !more) ? gp[j] : gpn[j]) + goff, dst + loff[j]);
                    if constexpr (ILV) {          // this group's four MFMAs with the fragment reads of the next half step between them (the first 4 NPL MFMAs)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (q < NPL) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x020, NDMA / NQ, 0);          // then this group's LDS-DMA pieces
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // F0's reads finished long ago (24 MFMAs back): a free wait that lets the compiler start the next step without one
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
                st = st1;
        };
        {
            constexpr int STRICT = (STAGES - 2) * NDMA, RELAXED = (STAGES - 2) * NDMA + NST;
            static_assert(RELAXED <= 63, "vmcnt is a 6-bit field");
            int c = 0;
            if (relaxed)
                for (; c < STAGES - 1 && c < nchunks; ++c) kstep(c, std::integral_constant<int, RELAXED>{});
            for (; c < nchunks; ++c) kstep(c, std::integral_constant<int, STRICT>{});
        }
        }

        // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The stores are younger than
        // the next tile's first chunk groups (issued in the last K steps above): waiting for those does not wait for the stores.
        float* cb = p.c + cur.cbase + (cur.n0 + wn * 64 + l31);
        const int nrem = p.N - (cur.n0 + wn * 64 + l31);          // column ni exists iff 32 * ni < nrem
        // Coalesced epilogue: the C/D layout gives a lane ONE column and 16 scattered rows, i.e. 64 four-byte stores per lane (512
        // store instructions of 256 B per tile: 55 of the kernel's 252 us, tools/x6p_dma_probe.sh).  Each wave owns 2 KiB of LDS
        // beyond the operand ring and transposes its 64 x 64 sub-tile through it 8 rows at a time: 8 ds_write_b32, then 2
        // ds_read_b128 + 2 global_store_dwordx4 per lane, every store instruction = 4 rows x 256 contiguous bytes.
        // (wide: the register-transposed form only -- its LDS holds no scratch beyond the ring)
        const bool vec_ok = (!WIDE || !(p.dbg & 8)) && (p.N - (cur.n0 + wn * 64) >= 64) && (p.ldc % 4 == 0) && !(p.dbg & 16);
        // exactly NST store instructions leave this wave only when all its 64 rows exist (a masked-off store is branched over)
        const bool full_rows = cur.m0 + wm * 64 + 64 <= p.M;
        const bool c16 = NPL == 1 && p.c16;          // (one-plane kernels: the product as fp16 under p.c_scale)
        relaxed = !WIDE && !c16 && vec_ok && full_rows && !(p.dbg & 32);          // (the wide K loop keeps the strict count; fp16 products: 8 stores, not NST)
        typedef _Float16 c16x4 __attribute__((ext_vector_type(4)));
        if (c16 && p.c16 == 1 && vec_ok && p.ldc % 8 == 0) {          // (c16 == 2, measurement: the 8-byte stores below)
            // fp16 product: EIGHT consecutive columns per lane = one 16-byte store.  After the quad transpose a lane holds four columns of
            // one row; the lanes of two adjacent quads (i, i + 4: the same rows, columns c .. c + 3 and c + 4 .. c + 7) swap one row quad
            // each (a lane ^ 4 exchange), so that the even quad stores the pair's first row quad and the odd one the second: 8 store
            // instructions per wave of 16 B per lane instead of 16 of 8 B (whose 64-byte row segments cost the same memory transactions)
            typedef _Float16 c16x8 __attribute__((ext_vector_type(8)));
            const bool odd = lane & 1, hi = lane & 2, oddq = (lane >> 2) & 1;
            const int m = cur.m0 + wm * 64 + 4 * lh + (lane & 3);          // + 32 mi + 8 rq
            _Float16* const g16 = (_Float16*)p.c + cur.cbase + (cur.n0 + wn * 64 + (l31 & ~7)) + (long)m * p.ldc;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const int rq = 2 * rp + (oddq ? 1 : 0);
                    const bool row_ok = m + mi * 32 + 8 * rq < p.M && !(p.dbg & 32);
                    _Float16* grow = g16 + (long)(mi * 32 + 8 * rq) * p.ldc;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const f32x4 oa = ss_quad_transpose(acc[0][mi][ni][8 * rp + 0], acc[0][mi][ni][8 * rp + 1], acc[0][mi][ni][8 * rp + 2], acc[0][mi][ni][8 * rp + 3], odd, hi) * p.c_scale;
                        const f32x4 ob = ss_quad_transpose(acc[0][mi][ni][8 * rp + 4], acc[0][mi][ni][8 * rp + 5], acc[0][mi][ni][8 * rp + 6], acc[0][mi][ni][8 * rp + 7], odd, hi) * p.c_scale;
                        // every lane hands its partner (lane ^ 4) the row quad the partner stores and receives the one it stores itself
                        f32x4 recv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) recv[e] = __shfl_xor(oddq ? oa[e] : ob[e], 4, 64);
                        const f32x4 lo = oddq ? recv : oa, hi4 = oddq ? ob : recv;
                        c16x8 o8;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o8[e] = (_Float16)lo[e]; o8[4 + e] = (_Float16)hi4[e]; }
                        if (row_ok) *(c16x8*)(grow + 32 * ni) = o8;
                    }
                }
            }
        } else if (vec_ok && (!(p.dbg & 8) || c16)) {
            // Register-transposed epilogue: per (mi, ni, row quad) the four registers of a lane are four consecutive rows of its
            // column; a 4 x 4 transpose inside every group of four adjacent lanes (ss_quad_transpose, DPP) turns them into four
            // consecutive columns of ONE row = one 16-byte store: 16 store instructions per wave (8 rows x 128 B each), no LDS trip.
            const bool odd = lane & 1, hi = lane & 2;
            const int m = cur.m0 + wm * 64 + 4 * lh + (lane & 3);          // + 32 mi + 8 rq
            float* g = p.c + cur.cbase + (cur.n0 + wn * 64 + (l31 & ~3)) + (long)m * p.ldc;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const bool row_ok = m + mi * 32 + 8 * rq < p.M && !(p.dbg & 32);
                    float* grow = g + (long)(mi * 32 + 8 * rq) * p.ldc;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float v[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            v[rr] = NACC == 1 ? acc[0][mi][ni][rq * 4 + rr] : fmaf(acc[NACC - 1][mi][ni][rq * 4 + rr], 4.8828125e-4f, acc[0][mi][ni][rq * 4 + rr]);
                        const f32x4 o = ss_quad_transpose(v[0], v[1], v[2], v[3], odd, hi);
                        if (row_ok) {
                            if (c16) *(c16x4*)((_Float16*)p.c + ((grow + 32 * ni) - p.c)) = __builtin_convertvector(o * p.c_scale, c16x4);
                            else if (p.dbg & 2) __builtin_nontemporal_store(o, (f32x4*)(grow + 32 * ni));          // measurement: streaming stores
                            else *(f32x4*)(grow + 32 * ni) = o;
                        }
                    }
                }
            }
        } else if (vec_ok) {
            float* tb = (float*)(lds + STAGES * STAGE_B + wave * 2048);          // [8 rows][64 cols]
            const int rrow = lane >> 4, rcol = (lane & 15) * 4;                    // read-back: row rrow (+4), columns rcol .. rcol + 3
            float* gbase = p.c + cur.cbase + (cur.n0 + wn * 64 + rcol);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int r = rq * 4 + rr;
                            const float v = NACC == 1 ? acc[0][mi][ni][r] : fmaf(acc[NACC - 1][mi][ni][r], 4.8828125e-4f, acc[0][mi][ni][r]);
                            tb[(rr + 4 * lh) * 64 + ni * 32 + l31] = v;
                        }
                    __builtin_amdgcn_wave_barrier();             // LDS operations of one wave execute in issue order: no wait between the writes and the reads (no other wave touches this slot)
                    const f32x4 v0 = *(const f32x4*)(tb + rrow * 64 + rcol);
                    const f32x4 v1 = *(const f32x4*)(tb + (rrow + 4) * 64 + rcol);
                    const int m0r = cur.m0 + wm * 64 + mi * 32 + 8 * rq;             // rows m0r .. m0r + 7 of C
                    if (!(p.dbg & 32)) {
                        if (m0r + rrow < p.M) *(f32x4*)(gbase + (long)(m0r + rrow) * p.ldc) = v0;
                        if (m0r + rrow + 4 < p.M) *(f32x4*)(gbase + (long)(m0r + rrow + 4) * p.ldc) = v1;
                    }
                    __builtin_amdgcn_wave_barrier();             // ... nor between these reads and the next group's writes
                }
            }
        } else
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = cur.m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= p.M || (p.dbg & 32)) continue;
                float* crow = cb + (long)m * p.ldc;               // one row pointer for both column tiles
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    if (32 * ni < nrem) {
                        const float v = NACC == 1 ? acc[0][mi][ni][r] : fmaf(acc[NACC - 1][mi][ni][r], 4.8828125e-4f, acc[0][mi][ni][r]);
                        if (c16) ((_Float16*)p.c)[(crow + 32 * ni) - p.c] = (_Float16)(v * p.c_scale);
                        else crow[32 * ni] = v;
                    }
            }
        }
        if (more) {
            // chunk 0 of the next tile must have landed before the barrier at the top.  Younger than it: the other STAGES - 1 chunk
            // groups and this epilogue's stores -- NST of them when `relaxed`, an unknown number otherwise (then none is assumed)
            if (relaxed) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * NDMA + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * NDMA) : "memory");
#pragma unroll
            for (int j = 0; j < NDMA; ++j) gp[j] = gpn[j];
            cur = nxt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may still be landing when the LDS is handed to the next workgroup
}


// ---------------------------------------------------------------------------------------------------------------------------------
// PING-PONG form of the x3h kernel (two fp16 planes, three products, two accumulator sets; same tile, same LDS image, same ring of
// three 48 KiB stages, same MFMA order per accumulator => bit-identical results to gemm_x6p_kernel<2>).
//
// What the one-phase kernel above loses (tools/x6p_dma_probe.sh): its K loop without any memory traffic runs at
// 0.50 of the MFMA peak, its LDS-DMA costs +76 us and its C stores +46 us per launch ON TOP of that -- they add instead of hiding,
// because every wave carries all three instruction kinds in ONE in-order stream: a wave whose LDS-DMA waits for a slot in the CU's
// vector-memory queue cannot issue the MFMAs behind it, and both waves of a SIMD reach that point together (they run the same phase).
//
// Here the two waves of a SIMD (waves w and w + 4: a workgroup's waves go to the SIMDs cyclically) run HALF A K STEP APART.  Time is
// cut into slots by s_barrier; in every slot one wave of each SIMD is in its C slot -- 24 back-to-back MFMAs at raised priority,
// nothing else -- while its partner is in its M slot: the 16 fragment reads of its next chunk (ONE fragment register set: the wave is
// not computing), its six LDS-DMA pieces of the chunk two ahead, the counted wait for the chunk one ahead, and at a tile boundary
// the epilogue of the tile it just finished.  Memory-queue stalls land on a wave that has a whole C slot (768 matrix-pipe cycles) of
// slack; the matrix pipe of every SIMD always has exactly one feeder.
//
//   group 0 (waves 0-3):  B0 |      M(0) | C(0) | M(1) | C(1) | ...                 | barrier
//   group 1 (waves 4-7):  B0 | bar | M(0) | C(0) | M(1) | ...                | C(last) |
//
// LDS ring, chunk c in stage c % 3 (barrier numbers: group 0 leaves M(c) through #2c+1, group 1 through #2c+2):
//   * chunk c is read in M(c): group 0 between #2c and #2c+1, group 1 between #2c+1 and #2c+2;
//   * chunk c+2 is DMA-issued in M(c) into the stage of chunk c-1, whose last reads (group 1, M(c-1)) retired before #2c;
//   * a wave waits for ITS pieces of chunk c+1 at the end of M(c) (vmcnt = the pieces of chunk c+2, plus the epilogue's stores
//     when they are younger) -- group 0 before #2c+1, group 1 before #2c+2 -- and chunk c+1 is first read after #2c+2.
// The epilogue of tile T runs at the start of M(0) of tile T+1, after that slot's DMA issue (stores younger than the pieces the next
// two waits are for); its partner is in C(last) of tile T meanwhile.
__device__ unsigned long long g_x6p_dbg[32];          // measurement only (tile_dbg & 512): slot timing of waves 0 and 4 of workgroup 0
// SPLIT = how many of a wave's six LDS-DMA pieces per chunk are issued in its C slot, one behind each group of four MFMAs, instead of
// in its M slot (measured: a burst of 24 pieces from the four M-slot waves of a CU costs ~140 cycles of issue per piece)
template <int SPLIT>
__global__ __launch_bounds__(512, 1) void gemm_x6p_pp_kernel(X6PParams p) {
    constexpr int NPL = 2;
    constexpr int STAGE_B = NPL * (A_PLANE_B + B_PLANE_B);          // 48 KiB
    constexpr int NWV = 8, WN = PBN / 64;
    constexpr int NDMA = STAGE_B / 1024 / NWV;                      // 6 LDS-DMA instructions per wave and chunk
    constexpr int STAGES = 3;
    typedef f16x8 FT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;          // 0: leads, 1: one slot behind
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int gridM = (p.M + PBM - 1) / PBM, gridN = (p.N + PBN - 1) / PBN;
    const int per = gridM * gridN;
    const int total = per * p.nbatch * p.splits;
    int t_first, t_end, t_stride;          // XCD-aware persistent tile walk, as gemm_x6p_kernel
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = total >> 3, r = total & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        if (nwg == total) { t_first = start + slot; t_end = t_first + 1; t_stride = 1; }
        else { t_first = start + slot; t_end = start + q + (xcd < r ? 1 : 0); t_stride = nwg >> 3; }
    }
    // LDS-DMA pieces of a wave and chunk: 4 of A (32 = 2 planes x 16 row blocks over 8 waves: one plane, four consecutive row blocks)
    // and 2 of B (16 = 2 planes x 8).  A piece = 16 rows x 64 B; lane (rl, sl) fetches the 16-byte slot sl ^ ((row >> 2) & 3) of row
    // rl of the block -- the same lane offset for every piece (the block index does not enter bits 2..3 of the row), so a piece's
    // address is a UNIFORM base (scalar registers) plus one of two per-lane offsets.
    const int rl = lane >> 2, sl = lane & 3;
    const int ko = sl ^ ((rl >> 2) & 3);
    const unsigned laneA = (unsigned)(rl * p.lda + 8 * ko) * 2u, laneB = (unsigned)(rl * p.ldb + 8 * ko) * 2u;          // bytes
    const int qa = wave * 4, qb = wave * 2;
    const int plA = qa / (PBM / 16), rbA = qa % (PBM / 16), plB = qb / (PBN / 16), rbB = qb % (PBN / 16);
    const int loffA = plA * A_PLANE_B + rbA * 1024, loffB = NPL * A_PLANE_B + plB * B_PLANE_B + rbB * 1024;
    struct TileCtx { int m0, n0, nchunks; long cbase; long ua, ub; };          // ua / ub: element offsets of this wave's first A / B piece
    auto setup = [&](int t) __attribute__((always_inline)) -> TileCtx {
        const int bs = t / per;
        const int tl = t - bs * per;
        const int batch = bs / p.splits, split = bs - batch * p.splits;
        const int m0 = (tl / gridN) * PBM, n0 = (tl % gridN) * PBN;
        const int k_begin = split * p.k_per_split;
        const int k_end = k_begin + p.k_per_split < p.K ? k_begin + p.k_per_split : p.K;
        const long ua = plA * p.a_plane + batch * p.a_bs + (long)(m0 + rbA * 16) * p.lda + k_begin;
        const long ub = plB * p.b_plane + batch * p.b_bs + (long)(n0 + rbB * 16) * p.ldb + k_begin;
        return TileCtx{m0, n0, (k_end - k_begin) / PBK, (long)batch * p.c_bs + (long)split * p.c_ss, ua, ub};
    };
    const bool skipA = p.dbg & 128, skipB = p.dbg & 64;          // measurement only
    // the six pieces of chunk (element offset goff from the tile's first chunk) of the tile with bases (ua, ub) into stage `dst`
    // piece j of a chunk: 0..3 = A row blocks, 4..5 = B row blocks
    auto dma_piece = [&](int j, long ua, long ub, long goff, unsigned char* dst) __attribute__((always_inline)) {
        if (j < 4) {
            const char* ga = (const char*)(p.a + ua + goff);
            if (!skipA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + (long)j * 32 * p.lda + laneA),
                                                 (__attribute__((address_space(3))) void*)(dst + loffA + j * 1024), 16, 0, 0);
        } else {
            const char* gb = (const char*)(p.b + ub + goff);
            if (!skipB)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + (long)(j - 4) * 32 * p.ldb + laneB),
                                                 (__attribute__((address_space(3))) void*)(dst + loffB + (j - 4) * 1024), 16, 0, 0);
        }
    };
    auto dma_chunk = [&](long ua, long ub, long goff, unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NDMA; ++j) dma_piece(j, ua, ub, goff, dst);
    };
    if (t_first >= t_end) return;
    TileCtx cur = setup(t_first);
    f32x16 acc[2][2][2];          // [0]: h*h, [1]: the cross terms (factor 2^-11)

    const int sw = (l31 >> 2) & 3;
    const int so0 = ((lh ^ sw) << 4), so1 = so0 ^ 32;
    const unsigned char* fa = lds + (wm * 64 + l31) * ROWB;
    const unsigned char* fb = lds + NPL * A_PLANE_B + (wn * 64 + l31) * ROWB;
    FT a0[NPL][2], b0[NPL][2], a1[NPL][2], b1[NPL][2];          // ONE set: both K halves of the chunk the next C slot multiplies
    auto frag = [&](FT (&a)[NPL][2], FT (&b)[NPL][2], int stage, int so) __attribute__((always_inline)) {
        const int sb = stage * STAGE_B;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[pl][mi] = *(const FT*)(fa + sb + pl * A_PLANE_B + mi * 32 * ROWB + so);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[pl][ni] = *(const FT*)(fb + sb + pl * B_PLANE_B + ni * 32 * ROWB + so);
        }
    };
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0}, HS[3] = {1, 1, 0};          // l*h, h*l (cross accumulators), h*h
    auto mma4 = [&](FT (&a)[NPL][2], FT (&b)[NPL][2], int q) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[HS[q]][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], acc[HS[q]][mi][ni], 0, 0, 0);
    };

    constexpr int NST = 16;          // store instructions of the coalesced epilogue per wave
    // epilogue of one tile (same code as gemm_x6p_kernel: LDS-transposed 16-byte stores where whole 64-column slabs exist);
    // returns true when it issued exactly NST store instructions in this wave
    auto epilogue = [&](const TileCtx& tc) __attribute__((always_inline)) {
        float* cb = p.c + tc.cbase + (tc.n0 + wn * 64 + l31);
        const int nrem = p.N - (tc.n0 + wn * 64 + l31);
        const bool vec_ok = (p.N - (tc.n0 + wn * 64) >= 64) && (p.ldc % 4 == 0) && !(p.dbg & 16);
        if (vec_ok) {
            // register-transposed epilogue (ss_quad_transpose): 16 stores of 16 B per lane, 8 rows x 128 B each, no LDS trip
            const bool odd = lane & 1, hi = lane & 2;
            const int m = tc.m0 + wm * 64 + 4 * lh + (lane & 3);          // + 32 mi + 8 rq
            float* g = p.c + tc.cbase + (tc.n0 + wn * 64 + (l31 & ~3)) + (long)m * p.ldc;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const bool row_ok = m + mi * 32 + 8 * rq < p.M && !(p.dbg & 32);
                    float* grow = g + (long)(mi * 32 + 8 * rq) * p.ldc;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float v[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) v[rr] = fmaf(acc[1][mi][ni][rq * 4 + rr], 4.8828125e-4f, acc[0][mi][ni][rq * 4 + rr]);
                        const f32x4 o = ss_quad_transpose(v[0], v[1], v[2], v[3], odd, hi);
                        if (row_ok) *(f32x4*)(grow + 32 * ni) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = tc.m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m >= p.M || (p.dbg & 32)) continue;
                    float* crow = cb + (long)m * p.ldc;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        if (32 * ni < nrem) crow[32 * ni] = fmaf(acc[1][mi][ni][r], 4.8828125e-4f, acc[0][mi][ni][r]);
                }
            }
        }
    };

    // pipeline fill: chunks 0 and 1 of the first tile (a tile has at least two chunks: the launcher checks)
    dma_chunk(cur.ua, cur.ub, 0, lds);
    dma_chunk(cur.ua, cur.ub, PBK, lds + STAGE_B);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");          // chunk 0 landed
    __builtin_amdgcn_s_barrier();                                        // B0
    if (grp) __builtin_amdgcn_s_barrier();                               // group 1 starts one slot later
    __builtin_amdgcn_sched_barrier(0);

    int st = 0;                  // stage of the chunk the next M slot reads
    bool have_prev = false;
    TileCtx prev = cur, nxt = cur;
    int relax = 0;               // M slots left in which the epilogue's NST stores may stay outstanding
    const bool tim = (p.dbg & 512) && blockIdx.x == 0 && (wave & 3) == 0;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (tim) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (k >= 0) tacc[k] += now - tlast;
            tlast = now;
        }
    };
    stamp(-1);
    for (int t = t_first; t < t_end; t += t_stride) {
        const int nchunks = cur.nchunks;
        const bool more = t + t_stride < t_end;
        if (more) nxt = setup(t + t_stride);
        for (int c = 0; c < nchunks; ++c) {
            long d_ua, d_ub, d_goff;
            unsigned char* d_dst;
            // ---------------- M slot: DMA of chunk c + 2, [epilogue of the previous tile], fragments of chunk c
            {
                const int ca = c + 2;
                const bool own = ca < nchunks;
                const int cn = own ? ca : (more ? ca - nchunks : nchunks - 1);
                const bool mine = own || !more;
                const int sd = st + 2 >= STAGES ? st + 2 - STAGES : st + 2;
                d_ua = mine ? cur.ua : nxt.ua; d_ub = mine ? cur.ub : nxt.ub; d_goff = (long)cn * PBK; d_dst = lds + sd * STAGE_B;
#pragma unroll
                for (int j = SPLIT; j < NDMA; ++j) dma_piece(j, d_ua, d_ub, d_goff, d_dst);
            }
            if (c == 0) {
                if (have_prev) {
                    // The stores are issued behind this slot's pieces (vmcnt counts in issue order): this wait and the next one are
                    // for pieces older than the stores, which may stay outstanding when their number is known (NST: whole 64 x 64
                    // slabs); an edge tile's unknown number is simply waited for
                    const bool exact = (p.N - (prev.n0 + wn * 64) >= 64) && (p.ldc % 4 == 0) && !(p.dbg & 16) && (prev.m0 + wm * 64 + 64 <= p.M) && !(p.dbg & 32);
                    epilogue(prev);
                    // SPLIT > 0: the next chunk's C-slot pieces are issued BEHIND the stores, so only this slot's wait can skip them
                    relax = exact ? (SPLIT == 0 ? 2 : 1) : 0;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[s][mi][ni][r] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);          // DMA issue (+ epilogue, zeroing)
            frag(a0, b0, st, so0);
            frag(a1, b1, st, so1);
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);          // fragment reads (the counter read waits for them)
            // this wave's pieces of chunk c + 1 landed (younger: the NDMA pieces of chunk c + 2, and the stores while `relax`), and its
            // fragment reads are complete: vmcnt(W) lgkmcnt(0)
            constexpr int WM = NDMA - SPLIT;          // pieces of chunk c + 2 issued in this slot (the other SPLIT follow in the C slot)
            if (relax > 0) {
                --relax;
                __builtin_amdgcn_s_waitcnt(0x0070 | ((WM + NST) & 15) | (((WM + NST) >> 4) << 14));
            } else {
                __builtin_amdgcn_s_waitcnt(0x0070 | (WM & 15) | ((WM >> 4) << 14));
            }
            stamp(2);          // vmcnt wait
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);          // barrier at the end of the M slot
            // ---------------- C slot: 24 MFMAs, nothing else
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                mma4(a0, b0, q);
                if (q < SPLIT) { dma_piece(q, d_ua, d_ub, d_goff, d_dst); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                mma4(a1, b1, q);
                if (q + 3 < SPLIT) { dma_piece(q + 3, d_ua, d_ub, d_goff, d_dst); __builtin_amdgcn_sched_barrier(0); }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);          // MFMA issue
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(5);          // barrier at the end of the C slot
            st = st + 1 == STAGES ? 0 : st + 1;
        }
        prev = cur;
        have_prev = true;
        if (more) cur = nxt;
    }
    if (tim && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) g_x6p_dbg[grp * 8 + k] = tacc[k];
    }
    epilogue(prev);
    if (!grp) __builtin_amdgcn_s_barrier();          // group 1 is still one slot behind: the barrier its last C slot ends with
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may still be landing when the LDS is handed to the next workgroup
}

}  // namespace

// Operand rows must be allocated up to the tile edge: A rows padded to SS_X6P_BM, B rows to SS_X6P_BN per batch (pad rows of
// A may hold anything finite or not: they only reach C rows >= M, which are not stored; pad rows of B likewise columns >= N).
// K (and k_per_split) must be multiples of 32; pad COLUMNS of a split-K operand must be zero.
bool ss_x6p_enabled() {
    return ss_tuning().x6p && ss_tuning().x6;
}

// x3h (fp16 two-piece operands, three products): SS_X3H=0 keeps the exact three-piece bf16 arithmetic everywhere
bool ss_x3h_enabled() {
    return ss_tuning().x3h && ss_x6p_enabled();
}

// One 512-thread workgroup per CU and 256x128 tiles: worth it from about four rounds of workgroups over the 256 CUs (batch >= 8 at
// 512x512 tiles); smaller problems keep the 128x128 / 128x64 / 64x64 register-staged kernels (conv_mfma_x6.hip), which fill the
// chip with more, smaller workgroups.  SS_X6P=force: always.
bool ss_x6p_wanted(long M, int N, int nbatch) {
    const bool force = ss_tuning().x6p == 2;
    if (!ss_x6p_enabled()) return false;
    const long nwg = ((M + PBM - 1) / PBM) * ((N + PBN - 1) / PBN) * nbatch;
    return force || nwg >= 1024;
}

// The wide tile: x3h planes with the plain low piece (p.plain_l, set by the caller that wrote them so), N a multiple of 256
bool ss_x6p_wide_ok(long M, int N, int K, int nbatch) {
    if (!ss_tuning().x6p_wide || !ss_x3h_enabled() || N < 256 || N % 256 || K % PBK || K < 2 * PBK) return false;
    return ss_tuning().x6p == 2 || ((M + PBM - 1) / PBM) * (N / 256) * nbatch >= 512;      // x6p = 2 ("force"): any size (tests)
}

int ss_launch_gemm_x6p(const X6PParams& p, hipStream_t s) {
    if (p.K % PBK || p.k_per_split % PBK || p.splits < 1 || p.lda % 8 || p.ldb % 8) return SS_ERR_UNSUPPORTED;
    const bool one = p.fp16x2 == 2;          // fp16x2 == 2: ONE fp16 plane per operand, one product (16-bit activation storage)
    // one plane on the 256 x 256 tile: one accumulator set anyway, the same MFMA order per accumulator as the 256 x 128 kernel -> the
    // same bits, two thirds of the operand bytes per output element (x6p_wide1 = 0: off)
    const bool wide1 = one && ss_tuning().x6p_wide1 && p.N % 256 == 0 && p.splits == 1 && p.K >= 2 * PBK &&
                       (ss_tuning().x6p == 2 || ((p.M + PBM - 1) / PBM) * (long)(p.N / 256) * p.nbatch >= 512);
    const bool wide = (p.fp16x2 == 1 && p.plain_l) || wide1;
    X6PParams pd = p;
    pd.dbg = ss_tuning().tile_dbg & (2 | 8 | 16 | 32 | 64 | 128);          // 8: the LDS-transposed epilogue, 16: the scalar-store epilogue (A/B measurement)
    if (wide && (p.N % 256 || p.splits != 1)) return SS_ERR_UNSUPPORTED;
    if (p.c16 && !one) return SS_ERR_UNSUPPORTED;
    const int pbn = wide ? 256 : SS_X6P_BN;
    const int gridM = (p.M + PBM - 1) / PBM, gridN = (p.N + pbn - 1) / pbn;
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)gemm_x6p_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const long tiles = (long)gridM * gridN * p.nbatch * p.splits;
    // one workgroup per CU (the LDS ring fills a CU): more tiles than CUs -> persistent workgroups (a multiple of 8: one share per XCD)
    static const int n_cu = [] { int v = 0; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0); return v >= 8 ? v / 8 * 8 : 256; }();
    const int cus = (ss_tuning().gemm_cus >= 8 && ss_tuning().gemm_cus < n_cu) ? ss_tuning().gemm_cus / 8 * 8 : n_cu;
    const bool persistent = ss_tuning().gemm_persistent && tiles > cus && p.k_per_split >= 3 * PBK && p.K % p.k_per_split == 0;
    const long nwg = persistent ? cus : tiles;
    // x3h planes, every tile at least two K chunks deep: the ping-pong kernel (same results bit for bit; x6p_pp = 0 keeps the one-phase kernel)
    if (p.fp16x2 == 1 && !wide && ss_tuning().x6p_pp && p.k_per_split >= 2 * PBK && p.K % p.k_per_split == 0) {
        static const bool pp_attr = [] {
            (void)hipFuncSetAttribute((const void*)gemm_x6p_pp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)gemm_x6p_pp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)gemm_x6p_pp_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return true;
        }();
        (void)pp_attr;
        pd.dbg = ss_tuning().tile_dbg & (16 | 32 | 64 | 128 | 512);
        const long nwg_pp = (ss_tuning().gemm_persistent && tiles > n_cu) ? n_cu : tiles;
        SsProfScope prof("gemm_x6p_pp_kernel", 2.0 * p.M * p.N * p.K * p.nbatch * 3,
                         2.0 * 2 * ((double)p.M + p.N) * p.K * p.nbatch + 4.0 * p.M * p.N * p.nbatch * p.splits, s);
        const int split = ss_tuning().x6p_pp;          // 1: all pieces in the M slot, 2: three of six in the C slot, 3: all six in the C slot
        const size_t ldsb = 3 * 2 * (A_PLANE_B + B_PLANE_B) + 16384;
        if (split == 3) hipLaunchKernelGGL(gemm_x6p_pp_kernel<6>, dim3((unsigned)nwg_pp), dim3(512), ldsb, s, pd);
        else if (split == 2) hipLaunchKernelGGL(gemm_x6p_pp_kernel<3>, dim3((unsigned)nwg_pp), dim3(512), ldsb, s, pd);
        else hipLaunchKernelGGL(gemm_x6p_pp_kernel<0>, dim3((unsigned)nwg_pp), dim3(512), ldsb, s, pd);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    SsProfScope prof(wide1 ? "gemm_x6p_kernel<1,wide>" : wide ? "gemm_x6p_kernel<2,wide>" : (one ? "gemm_x6p_kernel<1>" : (p.fp16x2 ? "gemm_x6p_kernel<2>" : "gemm_x6p_kernel<3>")),
                     2.0 * p.M * p.N * p.K * p.nbatch * (one ? 1 : (p.fp16x2 ? 3 : 6)),
                     2.0 * (one ? 1 : (p.fp16x2 ? 2 : 3)) * ((double)p.M + p.N) * p.K * p.nbatch + 4.0 * p.M * p.N * p.nbatch * p.splits, s);
    if (wide1) hipLaunchKernelGGL((gemm_x6p_kernel<1, true>), dim3((unsigned)nwg), dim3(1024), x6p_stages(1, true) * (A_PLANE_B + 256 * ROWB), s, pd);
    else if (one) hipLaunchKernelGGL((gemm_x6p_kernel<1, false>), dim3((unsigned)nwg), dim3(512), x6p_stages(1) * 1 * (A_PLANE_B + B_PLANE_B) + 16384, s, pd);
    else if (wide) hipLaunchKernelGGL((gemm_x6p_kernel<2, true>), dim3((unsigned)nwg), dim3(1024), x6p_stages(2, true) * 2 * (A_PLANE_B + 256 * ROWB), s, p);
    // x3h: fragment reads interleaved with the MFMAs (ILV; bit-identical; gemm_ilv = 0 keeps the burst form)
    else if (p.fp16x2 && ss_tuning().gemm_ilv) hipLaunchKernelGGL((gemm_x6p_kernel<2, false, true>), dim3((unsigned)nwg), dim3(512), x6p_stages(2) * 2 * (A_PLANE_B + B_PLANE_B) + 16384, s, pd);
    else if (p.fp16x2) hipLaunchKernelGGL((gemm_x6p_kernel<2, false>), dim3((unsigned)nwg), dim3(512), x6p_stages(2) * 2 * (A_PLANE_B + B_PLANE_B) + 16384, s, pd);
    else hipLaunchKernelGGL((gemm_x6p_kernel<3, false>), dim3((unsigned)nwg), dim3(512), x6p_stages(3) * 3 * (A_PLANE_B + B_PLANE_B) + 16384, s, pd);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
