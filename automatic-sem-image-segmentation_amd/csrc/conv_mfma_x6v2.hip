// Gather convolution (implicit GEMM over taps x input channels) with the x3h arithmetic, second structure.
//     out[pixel][co] = act(bias + sum_{t, ci} in[map(pixel, t)][ci] * w[t][ci][co])
// conv_mfma_x6.hip (gconv_x6_kernel) keeps ONE LDS tile and alternates "compute" and "split + store" phases between two barriers
// per K step; it reaches 0.21-0.29 of the 16-bit MFMA peak on the strided / transposed / 4x4 layers.  This kernel gives the same
// problem the schedule of gemm_x6p.hip:
//   * tile 256 (pixels) x 128 (output channels) x 32 (K), 512 threads = 8 waves as 4 (M) x 2 (N), 64x64 per wave;
//   * TWO LDS stages, ONE barrier per K step: while the matrix cores work on stage s, the same waves split the next chunk of the
//     gathered fp32 activations into fp16 (h, l) pieces and store them to stage s^1 (the activations were requested two K steps
//     ahead into registers), and the weight planes of that chunk -- already split, K-contiguous rows, cached per weight version --
//     arrive by LDS-DMA with the XOR slot swizzle of gemm_x6p.hip;
//   * A rows: 80-byte stride (conflict-free ds_read_b128 fragments, 8-byte stores); zero padding / out-of-image taps load from a
//     16-byte zero page (one pointer select per load instead of a value select per element).
// Same arithmetic as gconv_x6_kernel<.., true>: x * s = h + l with one power-of-two scale per operand tensor, products l*h, h*l,
// h*h into one accumulator set, scales undone in the epilogue; results are bit-identical to it (same products, same K order).
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ float v2_zero_page16[4] = {0.f, 0.f, 0.f, 0.f};
// activation storage types: four consecutive channels <-> f32x4 (16-bit types: one 8-byte access)
typedef _Float16 v2_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 v2_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 v2_ld4(const float* q) { return *(const f32x4*)q; }
__device__ __forceinline__ f32x4 v2_ld4(const _Float16* q) { return __builtin_convertvector(*(const v2_f16x4*)q, f32x4); }
__device__ __forceinline__ f32x4 v2_ld4(const __bf16* q) { return __builtin_convertvector(*(const v2_bf16x4*)q, f32x4); }
__device__ __forceinline__ void v2_st4(float* q, f32x4 v) { *(f32x4*)q = v; }
__device__ __forceinline__ void v2_st4(_Float16* q, f32x4 v) { *(v2_f16x4*)q = __builtin_convertvector(v, v2_f16x4); }
__device__ __forceinline__ void v2_st4(__bf16* q, f32x4 v) { *(v2_bf16x4*)q = __builtin_convertvector(v, v2_bf16x4); }
template <typename T> struct V2Raw { typedef u32x2 type; };          // what a lane keeps of its 4 channels between request and split
template <> struct V2Raw<float> { typedef f32x4 type; };
__device__ __forceinline__ f32x4 v2_widen(f32x4 r, const float*) { return r; }
__device__ __forceinline__ f32x4 v2_widen(u32x2 r, const _Float16*) { return __builtin_convertvector(__builtin_bit_cast(v2_f16x4, r), f32x4); }
__device__ __forceinline__ f32x4 v2_widen(u32x2 r, const __bf16*) { return __builtin_convertvector(__builtin_bit_cast(v2_bf16x4, r), f32x4); }
constexpr bool ss_v2_scalar_epilogue = false;          // true: the one-column-per-lane stores (A/B measurement builds)

constexpr int VBM = 256, VK = 32;
constexpr int VLD = VK + 8;                       // A row stride in halfs (80 bytes)
constexpr int VA_PLANE = VBM * VLD * 2;           // bytes: 20480
constexpr int V_MAX_TAPS = 16;
// VBN = 128 (64 x 64 per wave) or 64 (64 x 32 per wave: the 64-channel layers, e.g. the last transposed convolution of a generator)
template <int VBN> struct VG {
    static constexpr int B_PLANE = VBN * VK * 2;                  // bytes: 64-byte rows, swizzled slots
    static constexpr int STAGE = 2 * VA_PLANE + 2 * B_PLANE;      // 57344 / 49152
    static constexpr int TN = VBN / 64;                           // 32-wide MFMA column tiles per wave
    static constexpr int NB = 2 * VBN / 16 / 8;                   // LDS-DMA pieces per wave and stage
};

__device__ __forceinline__ void dma16(const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// T: activation storage type.  16-bit storage (NPROD < 3): the stored value IS the operand -- one fp16 plane of A (scaled by a power of
// two, exact), no low piece: NPROD = 2 multiplies it with both weight pieces (the exact product of the stored activation and the fp32
// weight), NPROD = 1 with the leading weight piece only (ss_tuning wino16_products); half the gathered bytes, a third of the split work.
template <int VBN, typename T, int NPROD>
__global__ __launch_bounds__(512, 1) void gconv_x6v2_kernel(GConvParams p, const unsigned short* __restrict__ bpl, long plane_elems, int Npad,
                                                            int Ktot, int ilv_flag, GPhases ph) {
    constexpr bool F32 = std::is_same<T, float>::value;
    static_assert(F32 ? NPROD == 3 : NPROD <= 2, "fp32 storage: three products; 16-bit storage: one or two");
    constexpr int VB_PLANE = VG<VBN>::B_PLANE, VSTAGE = VG<VBN>::STAGE, TN = VG<VBN>::TN, NB = VG<VBN>::NB;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    int* pixtab = (int*)(lds + 2 * VSTAGE);          // [VBM]
    int* offtab = pixtab + VBM;                      // [VBM][ntaps]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const long M = (long)p.N * p.OHc * p.OWc;
    const int gridN = (p.Cout + VBN - 1) / VBN;
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
    const int gridM = (int)((M + VBM - 1) / VBM);
    const int batch = tile / (gridM * gridN);
    tile -= batch * gridM * gridN;
    const T* const g_in = (const T*)p.in + (long)batch * p.in_bs;
    T* const g_out = (T*)p.out + (long)batch * p.out_bs;
    const long m0 = (long)(tile / gridN) * VBM;
    const int n0 = (tile % gridN) * VBN;
    const int nchunks = Ktot / VK;
    const int Cq = Ktot / P_ntaps;       // = Cin (a multiple of 32: launcher)
    const int ea = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax, p.amax_stripes))), ew = ss_amax_exp(__uint_as_float(p.h_amax2[0]));
    const float a_scale = ldexpf(1.f, 14 - ea);
    const float out_scale = ldexpf(1.f, ea - 14 + ew - 14);

    {   // pixel decode (32-bit: M < 2^31) and the per-row tap offsets, as in gconv_x6_kernel
        int* rowc = offtab + VBM * P_ntaps;      // [3][VBM]
        if (tid < VBM) {
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
            rowc[VBM + tid] = ry;
            rowc[2 * VBM + tid] = rx;
        }
        __syncthreads();
        const int row = tid % VBM;
        const int rn = rowc[row], ry = rowc[VBM + row], rx = rowc[2 * VBM + row];
        for (int t = tid / VBM; t < P_ntaps; t += 512 / VBM) {
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

    // A loader: rows arow + 64 j (j < 4), channels 4 c4a .. +3 of the chunk
    const int c4a = tid & 7, arow = tid >> 3;
    // B (weight planes) by LDS-DMA: 2 x VBN / 16 pieces of 16 rows per stage, NB per wave
    const unsigned short* gb[NB];
    int lb[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = wave * NB + j;
        const int pl = q / (VBN / 16), rb = q % (VBN / 16);
        const int row = rb * 16 + (lane >> 2);
        const int ko = (lane & 3) ^ ((row >> 2) & 3);
        gb[j] = bpl + pl * plane_elems + ((long)batch * Npad + n0 + row) * Ktot + 8 * ko;
        lb[j] = 2 * VA_PLANE + pl * VB_PLANE + rb * 1024;
    }

    typename V2Raw<T>::type ra[2][4];
    const T* const zpage = (const T*)v2_zero_page16;    // zero-padding / out-of-image taps READ zeros (one pointer select per load)
    auto load_a = [&](auto setc, int chunk) {
        constexpr int S = decltype(setc)::value;
        const int k0 = chunk * VK;
        const int t = k0 / Cq;                         // block-uniform
        const T* abase = g_in + (k0 - t * Cq) + c4a * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = offtab[(arow + 64 * j) * P_ntaps + t];
            const T* pa = off < 0 ? zpage : abase + off;
            ra[S][j] = *(const typename V2Raw<T>::type*)pa;
        }
    };
    auto store_a = [&](auto setc, int stage, int j) {          // row arow + 64 j of the chunk held in register set S -> LDS stage
        constexpr int S = decltype(setc)::value;
        const f32x4 v = v2_widen(ra[S][j], (const T*)nullptr);
        unsigned int hh[2], ll[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ss_split_h2(v[2 * e] * a_scale, v[2 * e + 1] * a_scale, hh[e], ll[e]);
        }
        unsigned char* dst = lds + stage * VSTAGE + ((arow + 64 * j) * VLD + c4a * 4) * 2;
        *(u32x2*)(dst) = u32x2{hh[0], hh[1]};
        if (F32) *(u32x2*)(dst + VA_PLANE) = u32x2{ll[0], ll[1]};          // 16-bit storage: h is the (scaled) stored value, exactly
    };
    auto dma_b = [&](int chunk, int stage) {
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16(gb[j] + (long)chunk * VK, lds + stage * VSTAGE + lb[j]);
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment addresses: A row = 64 wm + 32 mi + l31, 16-byte k-octet lh + 2 ks;  B row = 64 wn + 32 ni + l31, slot (lh + 2 ks) ^ ((row >> 2) & 3)
    const unsigned char* fa = lds + ((wm * 64 + l31) * VLD + 8 * lh) * 2;
    const int sw = (l31 >> 2) & 3;
    const int so0 = (lh ^ sw) << 4, so1 = so0 ^ 32;
    const unsigned char* fb = lds + 2 * VA_PLANE + (wn * (VBN / 2) + l31) * 64;
    f16x8 a0[2][2], b0[2][TN], a1[2][2], b1[2][TN];          // [plane][mi / ni]
    auto frag = [&](f16x8 (&a)[2][2], f16x8 (&b)[2][TN], int stage, int ks) {
        const int sb = stage * VSTAGE;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if (pl == 0 || F32) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[pl][mi] = *(const f16x8*)(fa + sb + pl * VA_PLANE + mi * 32 * VLD * 2 + ks * 32);
            }
            if (pl == 0 || NPROD >= 2) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) b[pl][ni] = *(const f16x8*)(fb + sb + pl * VB_PLANE + ni * 32 * 64 + (ks ? so1 : so0));
            }
        }
    };
    // l*h, h*l, h*h (the order of gconv_x6_kernel); consecutive MFMAs go to different accumulators.  16-bit storage: h*l, h*h or h*h
    constexpr int HA[3] = {NPROD == 3 ? 1 : 0, 0, 0}, HB[3] = {NPROD == 3 ? 0 : (NPROD == 2 ? 1 : 0), NPROD == 3 ? 1 : 0, 0};
    auto mma4 = [&](f16x8 (&a)[2][2], f16x8 (&b)[2][TN], int q) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[HA[q]][mi], b[HB[q]][ni], acc[mi][ni], 0, 0, 0);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int last = nchunks - 1;
    // prologue: chunk 0 into stage 0, chunk 1 requested into register set 1
    dma_b(0, 0);
    load_a(S0{}, 0);
    load_a(S1{}, 1 < nchunks ? 1 : last);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_a(S0{}, 0, j);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frag(a0, b0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)

    // step c: stage c&1 holds chunk c; chunk c+1 sits in register set (c+1)&1 (requested one step ago) and goes to the other
    // stage under the first twelve MFMAs; chunk c+2 is requested into set c&1; the weight planes of chunk c+1 arrive by DMA
    auto step = [&](int c, auto cur, auto nxt) {
        const int s = c & 1;
        // the activations of chunk c+1 (requested one step ago) must be in their registers before the split below.  Waiting HERE,
        // where nothing younger is in flight, keeps the compiler from placing its own vmcnt(0) after this step's requests (it does
        // not count through LDS-DMA instructions): a real s_waitcnt so that its bookkeeping sees it
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        dma_b(c + 1 < nchunks ? c + 1 : last, s ^ 1);            // 2 DMA, then 4 loads: the order the wait below counts on
        __builtin_amdgcn_sched_barrier(0);
        load_a(cur, c + 2 < nchunks ? c + 2 : last);
        __builtin_amdgcn_sched_barrier(0);
        // (gemm_ilv) the other K half's fragment reads go BEHIND the first MFMA group instead of in front of it: after the barrier all
        // eight waves would issue their reads at once and every wave's first MFMA would wait behind its own reads (gemm_x6p.hip ILV)
        const bool ilv = ilv_flag != 0;
        if (!ilv) frag(a1, b1, s, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (q < NPROD) mma4(a0, b0, q);
            if (q == 0 && ilv) frag(a1, b1, s, 1);
            store_a(nxt, s ^ 1, q);
            if (q == 2) store_a(nxt, s ^ 1, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the DMA of chunk c+1 landed (the four younger activation loads may still be in flight), own LDS reads / writes done
        __builtin_amdgcn_s_waitcnt(0x0070 | 4);      // vmcnt(4) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!ilv) {
            frag(a0, b0, s ^ 1, 0);
#pragma unroll
            for (int q = 0; q < NPROD; ++q) mma4(a1, b1, q);
        } else {
            mma4(a1, b1, 0);
            __builtin_amdgcn_sched_barrier(0);
            frag(a0, b0, s ^ 1, 0);
#pragma unroll
            for (int q = 1; q < NPROD; ++q) mma4(a1, b1, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, S0{}, S1{});
        if (c + 1 < nchunks) step(c + 1, S1{}, S0{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may still be landing when the LDS is handed on

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Pixel outer, column tile inner:
    // one table read and one 32-bit offset product per output ROW (the launcher checks that the output view is below 2^31 elements)
    int co[TN];
    float bv[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        co[ni] = n0 + wn * (VBN / 2) + ni * 32 + l31;
        bv[ni] = (p.bias && co[ni] < p.Cout) ? p.bias[co[ni]] : 0.f;
    }
    // Coalesced epilogue (as gemm_x6p.hip): a lane of the C/D layout owns ONE column and 16 scattered rows = 32 four-byte stores
    // per column tile; instead every wave transposes its 64 x (32 TN) sub-tile through 2 KiB of the (now free) operand stages, 8
    // rows at a time, and stores 16 bytes per lane: 4 (TN = 2) or 8 (TN = 1) whole pixel rows of the wave's columns per instruction.
    constexpr int COLS = 32 * TN, LPR = COLS / 4, RPI = 64 / LPR, NRD = 8 / RPI;      // lanes per row, rows per read instruction, reads per group
    const int cbase = n0 + wn * (VBN / 2);
    const bool vec_ok = cbase + COLS <= p.Cout && (p.out_cs & 3) == 0 && (((uintptr_t)g_out) & 15) == 0 && !(ss_v2_scalar_epilogue);
    __builtin_amdgcn_s_barrier();              // every wave is done reading the last chunk's fragments: the stages are free (all waves: vec_ok differs per wave)
    if (vec_ok) {
        float* tb = (float*)(lds + wave * 2048);          // [8 rows][COLS]
        const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4 = *(const f32x4*)(p.bias + cbase + rcol);
        f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};          // p.stats: this lane's 4 channels over the rows it stores
        // ACC / PLAIN are compile-time inside the store loop: with `if (p.accumulate) v += load` in it the compiler put an
        // s_waitcnt vmcnt(0) around EVERY store (the conditional load's result joins the store path), i.e. each of the 16 stores
        // waited for the previous one's round trip -- and the generic activation switch sat between them.
        auto epi = [&](auto acc_c, auto plain_c) {
            constexpr bool ACC = decltype(acc_c)::value, PLAIN = decltype(plain_c)::value;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni) tb[(rr + 4 * lh) * COLS + ni * 32 + l31] = acc[mi][ni][rq * 4 + rr] * out_scale;
                    __builtin_amdgcn_wave_barrier();          // one wave's LDS operations execute in issue order
#pragma unroll
                    for (int k = 0; k < NRD; ++k) {
                        const int row = rrow + RPI * k;
                        f32x4 v = *(const f32x4*)(tb + row * COLS + rcol);
                        const int pix = pixtab[wm * 64 + mi * 32 + 8 * rq + row];
                        if (pix >= 0) {
                            T* op = g_out + pix * p.out_cs + cbase + rcol;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = PLAIN ? v[e] + b4[e] : ss_apply_act(v[e] + b4[e], p.act, p.alpha);
                            if (ACC) v += v2_ld4(op);
                            v2_st4(op, v);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { st1[e] += v[e]; st2[e] = fmaf(v[e], v[e], st2[e]); }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
        using TC = std::true_type;
        using FC = std::false_type;
        if (p.accumulate) { if (p.act == SS_ACT_NONE) epi(TC{}, TC{}); else epi(TC{}, FC{}); }
        else { if (p.act == SS_ACT_NONE) epi(FC{}, TC{}); else epi(FC{}, FC{}); }
        if (p.stats) {
            // output statistics of the tile's 256 rows (one chunk of ss_conv_desc::y_stats: the launcher lets a tile hold rows of ONE
            // sample only and every wave is on this path), fixed order: the lanes that share 4 channels, then the four M waves
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { st1[e] += __shfl_xor(st1[e], off, 64); st2[e] += __shfl_xor(st2[e], off, 64); }
            float* sst = (float*)(lds + 8 * 2048);                // [8 waves][COLS][2]
            if (lane < LPR) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { sst[(wave * COLS + rcol + e) * 2] = st1[e]; sst[(wave * COLS + rcol + e) * 2 + 1] = st2[e]; }
            }
            __syncthreads();
            if (tid < 2 * VBN) {
                const int col = tid >> 1, k = tid & 1;             // column of the tile: wave pair wn = col / COLS
                const int w_n = col / COLS, cc = col % COLS;
                const float v = (sst[((0 * 2 + w_n) * COLS + cc) * 2 + k] + sst[((1 * 2 + w_n) * COLS + cc) * 2 + k]) +
                                (sst[((2 * 2 + w_n) * COLS + cc) * 2 + k] + sst[((3 * 2 + w_n) * COLS + cc) * 2 + k]);
                const long per = (long)p.OHc * p.OWc / VBM;       // tiles (= chunks) per sample
                const long mt = m0 / VBM;
                p.stats[((mt / per * p.stats_chunks + mt % per) * p.Cout + n0 + col) * 2 + k] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int pix = pixtab[wm * 64 + mi * 32 + row];
            if (pix < 0) continue;
            T* orow = g_out + pix * p.out_cs;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                if (co[ni] >= p.Cout) continue;
                T* op = orow + co[ni];
                float v = ss_apply_act(acc[mi][ni][r] * out_scale + bv[ni], p.act, p.alpha);
                if (p.accumulate) v += (float)*op;
                *op = (T)v;
            }
        }
    }
}

}  // namespace

// shapes this structure takes: x3h, whole 32-channel chunks, aligned rows, at most 16 taps (LDS tables), enough 256 x VBN tiles
static int v2_bn(const GConvParams& p) { return p.Cout > 64 ? 128 : 64; }
bool ss_gconv_x6v2_ok(const GConvParams& p) {
    if (!ss_tuning().gconv_v2 || !p.h_amax || !p.h_amax2 || p.ntaps < 1 || p.ntaps > V_MAX_TAPS) return false;
    if (p.Cin % 32 || (p.in_cs & 3) || (((uintptr_t)p.in) & 15) || p.Cout < 48 || (p.Cout > 64 && p.Cout < 96)) return false;
    // 64 output channels with a short reduction (the sub-pixel phases of the generators' last transposed convolution: K = 128 .. 512,
    // a million pixels) are bound by their output stream; there the lighter 128 x 64 workgroups of gconv_x6_kernel measure 15-20 % faster
    if (p.Cout <= 64 && (long)p.ntaps * p.Cin < 1024) return false;
    const long M = (long)p.N * p.OHc * p.OWc;
    if (M >= (1L << 31) || (long)p.N * p.IH * p.IW * p.in_cs >= (1L << 31) || (long)p.N * p.OH * p.OW * p.out_cs >= (1L << 31)) return false;
    const int nb = p.nbatch > 1 ? p.nbatch : 1, bn = v2_bn(p);
    return ((M + VBM - 1) / VBM) * ((p.Cout + bn - 1) / bn) * nb >= 200;
}

// output statistics (GConvParams::stats): forward problems whose 256-row tiles hold rows of one sample and whole column tiles
int ss_gconv_x6v2_stats_chunks(const GConvParams& p) {
    const int bn = v2_bn(p);
    if (p.act != SS_ACT_NONE || p.accumulate || p.nbatch > 1 || p.out_s != 1 || p.out_oy || p.out_ox || p.OHc != p.OH || p.OWc != p.OW) return 0;
    if (((long)p.OHc * p.OWc) % VBM || p.Cout % bn || (p.out_cs & 3)) return 0;
    return (int)((long)p.OHc * p.OWc / VBM);
}

template <int VBN, typename T, int NPROD>
static int launch_v2(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph) {
    const long M = (long)p.N * p.OHc * p.OWc;
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)gconv_x6v2_kernel<VBN, T, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const int nph = ph && ph->count > 1 ? ph->count : 1;
    const long nwg1 = ((M + VBM - 1) / VBM) * ((p.Cout + VBN - 1) / VBN) * nb;
    if (nph > 1 && (nwg1 % 8 || p.stats)) return SS_ERR_UNSUPPORTED;          // several problems per launch: whole XCD rounds per problem
    const long nwg = nwg1 * nph;
    int mt = p.ntaps;
    for (int i = 0; i < nph && ph; ++i) mt = ph->ntaps[i] > mt ? ph->ntaps[i] : mt;
    const size_t smem = (size_t)2 * VG<VBN>::STAGE + (size_t)VBM * sizeof(int) * (4 + mt);
    char pname[64];
    if (getenv("SS_PROF_SHAPES")) snprintf(pname, sizeof(pname), "gconv_x6v2<%d> M%ld N%d K%dx%d s%d b%d", VBN, M, p.Cout, p.ntaps, p.Cin, p.in_s, nb);
    else if (std::is_same<T, float>::value) snprintf(pname, sizeof(pname), "gconv_x6v2_kernel<%d>", VBN);
    else snprintf(pname, sizeof(pname), "gconv_x6v2_kernel<%d,16-bit,%d>", VBN, NPROD);
    int taps_all = p.ntaps;
    if (nph > 1) { taps_all = 0; for (int i = 0; i < nph; ++i) taps_all += ph->ntaps[i]; }
    SsProfScope prof(pname, 2.0 * M * p.Cout * taps_all * p.Cin * nb * NPROD,
                     (double)sizeof(T) * nb * ((double)p.N * p.IH * p.IW * p.Cin + (double)M * p.Cout * nph) + 4.0 * nb * taps_all * p.Cin * p.Cout, s);
    if (p.stats && (ss_gconv_x6v2_stats_chunks(p) != p.stats_chunks || (((uintptr_t)p.out) & 15))) {
        ss_set_error("gconv_x6v2: output statistics requested for a problem whose tiles do not line up with the samples");
        return SS_ERR_UNSUPPORTED;
    }
    GPhases phv{};
    if (nph > 1) phv = *ph;
    hipLaunchKernelGGL((gconv_x6v2_kernel<VBN, T, NPROD>), dim3((unsigned)nwg), dim3(512), smem, s, p, planes, plane_elems, Npad, Ktot, ss_tuning().gemm_ilv, phv);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T, int NPROD>
static int launch_v2_bn(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph) {
    return v2_bn(p) == 128 ? launch_v2<128, T, NPROD>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_v2<64, T, NPROD>(p, planes, plane_elems, Npad, Ktot, s, ph);
}
int ss_launch_gconv_x6v2(const GConvParams& p, const unsigned short* planes, long plane_elems, int Npad, int Ktot, hipStream_t s, const GPhases* ph) {
    if (p.dtype == SS_DTYPE_F32) return launch_v2_bn<float, 3>(p, planes, plane_elems, Npad, Ktot, s, ph);
    if (p.stats) { ss_set_error("gconv_x6v2: output statistics are reported for fp32 storage only"); return SS_ERR_UNSUPPORTED; }
    const bool two = ss_tuning().wino16_products == 3;          // "fp32-grade arithmetic, only the storage is 16-bit"
    if (p.dtype == SS_DTYPE_F16) return two ? launch_v2_bn<_Float16, 2>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_v2_bn<_Float16, 1>(p, planes, plane_elems, Npad, Ktot, s, ph);
    if (p.dtype == SS_DTYPE_BF16) return two ? launch_v2_bn<__bf16, 2>(p, planes, plane_elems, Npad, Ktot, s, ph) : launch_v2_bn<__bf16, 1>(p, planes, plane_elems, Npad, Ktot, s, ph);
    return SS_ERR_UNSUPPORTED;
}
