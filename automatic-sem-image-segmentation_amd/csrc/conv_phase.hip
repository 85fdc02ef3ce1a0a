// Stride-2 data gradients / transposed convolutions: ALL FOUR sub-pixel phases of an input tile in one workgroup.
//     out[n, s yc + ry, s xc + rx, co] = act(bias + sum_{t in taps(ry,rx), ci} in[n, yc + dy_t, xc + dx_t, ci] * w_t[ci][co]),   s = 2
// (CycleGAN.py:347-358: the generators' Conv2DTranspose(3, strides 2) layers; :339-345 / :425-451: the data gradients of their and
// the discriminators' stride-2 convolutions).  conv_bwd_data launches one gather convolution per phase (ry, rx): every launch
// re-gathers the input rows of its taps, splits them into fp16 pieces again (the gather kernels are VALU-bound on exactly that:
// ~170 address + ~190 split instructions per thread and 32-deep K step against 24 MFMAs, profiles/r02_f_*) and runs a short K loop
// (1 .. 4 taps x Cin).  Here a workgroup owns a tile of 8 x 16 CLASS pixels (= 16 x 32 output pixels) and 64 output channels:
//   * per 32-channel chunk the input tile + halo (<= 10 x 18 pixels) is loaded ONCE, split ONCE into (h, l) fp16 planes and staged
//     in LDS; every tap of every phase reads its MFMA A operand from that stage at a uniform (tap) offset -- no gather, no split,
//     no per-tap address arithmetic;
//   * the weight planes of one tap and chunk (64 rows x 64 B x 2 planes, pre-split and cached per weight version: the planes of the
//     per-phase gather path, ss_launch_wprep_x6) arrive by LDS-DMA into a ring of four 8 KiB stages, two taps ahead;
//   * 4 waves, 128 accumulator registers each; 61 KiB of LDS: TWO workgroups per CU, whose load / split / multiply phases interleave by
//     themselves (a single 8-wave workgroup per CU measured slower for the gather weight gradient).  Two forms:
//     SPLIT = 0: every wave holds 32 class pixels x 64 channels of ALL FOUR phases (1.5 LDS fragment reads per MFMA triple);
//     SPLIT = 1 / 2 (ss_tuning phases_split, the default where the tap counts fit its patterns): waves 0 / 1 hold 64 pixels x 64
//     channels of two phases, waves 2 / 3 of the other two (1.0 reads per triple -- the loop is bound by the CU's LDS read rate), and
//     the weight stream interleaves the two wave pairs' taps.
// Arithmetic = gconv_x6v2_kernel's: x * 2^(14 - ea) = h + l, weights h + l under their own scale, products l*h, h*l, h*h into one fp32
// accumulator, scales undone in the epilogue.  The K order differs (channel chunk outer, tap inner), so results agree with the
// per-phase path to fp32 rounding, not bit for bit.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int PF_TH = 8, PF_TW = 16;                  // class-pixel tile: 128 pixels, 32 per wave (two rows of 16)
constexpr int PF_LD = 80;                             // bytes per staged pixel and plane: 32 channels fp16 + 16 (conflict-free b128 reads)
constexpr int PF_MAXPIX = (PF_TH + 2) * (PF_TW + 2);  // halo up to one pixel on every side
// A stage row pitch: the halo row (16 .. 18 pixels x 80 B) padded to a multiple of 256 B.  ds_read_b128 is served in four lane groups that
// MIX the two pixel rows of a fragment ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md, LDS): with rows hw * 80 B apart, hw = 17 / 18 put two
// 16-byte slots of every group on busy banks (2-way: every A read cost twice -- SQ_LDS_BANK_CONFLICT was 27 % of the LDS cycles); with the
// rows a whole number of 256-byte bank rows apart the 16 lanes of a group hit 16 different slots (5 c mod 16 over the columns).
constexpr int PF_PITCH_MAX = ((PF_TW + 2) * PF_LD + 255) / 256 * 256;          // 1536
constexpr int PF_A_PLANE = (PF_TH + 2) * PF_PITCH_MAX;                         // 15360
constexpr int PF_B_TAP = 2 * 64 * 64;                 // [2 planes][64 output channels][32 channels fp16]
constexpr int PF_RING = 4;
constexpr int PF_SMEM = 2 * PF_A_PLANE + PF_RING * PF_B_TAP;          // 63488
constexpr int PF_SLOTS = (PF_MAXPIX * 8 + 255) / 256;                 // float4 staging slots per thread: 6

__device__ float pf_zero_page16[4] = {0.f, 0.f, 0.f, 0.f};
// activation storage types (GConvParams::dtype): four consecutive channels <-> f32x4 (the 16-bit types: one 8-byte access)
typedef _Float16 pf_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pf_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pf_ld4(const float* q) { return *(const f32x4*)q; }
__device__ __forceinline__ f32x4 pf_ld4(const _Float16* q) { return __builtin_convertvector(*(const pf_f16x4*)q, f32x4); }
__device__ __forceinline__ f32x4 pf_ld4(const __bf16* q) { return __builtin_convertvector(*(const pf_bf16x4*)q, f32x4); }
__device__ __forceinline__ void pf_st4(float* q, f32x4 v) { *(f32x4*)q = v; }
__device__ __forceinline__ void pf_st4(_Float16* q, f32x4 v) { *(pf_f16x4*)q = __builtin_convertvector(v, pf_f16x4); }
__device__ __forceinline__ void pf_st4(__bf16* q, f32x4 v) { *(pf_bf16x4*)q = __builtin_convertvector(v, pf_bf16x4); }

struct PFParams {
    const float* in;
    float* out;
    const float* bias;
    int N, IH, IW, Cin, in_cs;
    int OH, OW, Cout, out_cs;
    int OHc, OWc;                  // class grid (the same for every phase)
    int in_oy, in_ox;              // input pixel of class pixel (yc, xc) and tap t: (yc + in_oy + dy_t, xc + in_ox + dx_t)
    int act;
    float alpha;
    int accumulate;
    const unsigned int* h_amax;    // max|in| (striped slot), max|w|
    const unsigned int* h_amax2;
    int amax_stripes;
    int hy0, hx0, hh, hw;          // halo tile: smallest tap offset, extents in pixels
    int pitch;                     // bytes between the halo rows of an A plane in LDS (a multiple of 256)
    int tiles_y, tiles_x, ngroups; // class-pixel tiles per sample, 64-channel output groups
    int ntaps[4], out_oy[4], out_ox[4];
    short tdy[4][4], tdx[4][4];
    const unsigned short* planes;  // ONE set of weight planes for all taps: [2 planes][Npad rows][T * Cin], k = flattened tap (phase-major) * Cin + channel
    long plane_elems;
    int KT;                        // T * Cin
    int Npad;
    int dbg;          // measurement only (tile_dbg): 1 no input loads, 2 no weight DMA, 4 no fragment reads / MFMAs, 8 no barriers per tap
    // SPLIT form (two phases per wave pair): streamed pair s holds tap s of wave pair 0 and tap s of wave pair 1 (flattened 2 s + wp)
    int sp_stages;                 // streamed pairs per chunk
    short sp_shift[2][8];          // [wave pair][s]: byte offset dy * pitch + dx * PF_LD of the tap inside the A stage
    signed char sp_lp[2][8];       // ... its local phase (0 / 1), -1: no tap (the shorter pair's last slot)
    int sp_ooy[2][2], sp_oox[2][2];          // [wave pair][local phase]: output sub-pixel of the phase
};

__device__ __forceinline__ void pf_dma16(const unsigned short* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// T: activation storage type.  fp32: NPROD = 3 (x = h + l against w = h + l without l*l).  16-bit storage: the stored value (times a power of
// two) IS the fp16 operand plane -- NPROD = 1: times the leading weight piece; 2: times both (the exact product of the stored value and the
// fp32 weight), as gconv_x6v2_kernel (ss_tuning wino16_products).
// SPLIT patterns (tap counts of the two phases of wave pair 0 | wave pair 1): 1 = {1, 4 | 2, 2} -- the 3 x 3 stride-2 layers, five streamed
// pairs; 2 = {4, 4 | 4, 4} -- the 4 x 4 stride-2 layers, eight.  Local phase of wave pair wp in streamed pair st, -1: no tap.
__host__ __device__ constexpr int pf_split_stages(int pat) { return pat == 1 ? 5 : 8; }
__host__ __device__ constexpr int pf_split_lp(int pat, int wp, int st) {
    return pat == 1 ? (wp == 0 ? (st == 0 ? 0 : 1) : (st < 2 ? 0 : (st < 4 ? 1 : -1))) : (st < 4 ? 0 : 1);
}

// SPLIT: waves 0 / 1 hold phases sp_ph[0][*] on pixel rows 0 - 3 / 4 - 7 of the tile (64 pixels each), waves 2 / 3 hold phases sp_ph[1][*]:
// acc[2 phases][2 pixel blocks][2 channel blocks] -- the same 128 registers, but one A and one B fragment per MFMA triple instead of 1.5
// (the loop is bound by the CU's LDS read rate).  Every streamed pair of taps holds one tap of each wave pair: all four SIMDs multiply in
// every stage.
template <typename T, int NPROD, int SPLIT>
__global__ __launch_bounds__(256, 2) void gconv_phases_fused_kernel(PFParams p) {
    constexpr bool F32 = std::is_same<T, float>::value;
    static_assert(F32 ? NPROD == 3 : NPROD <= 2, "fp32 storage: three products; 16-bit storage: one or two");
    const T* const g_in = (const T*)p.in;
    T* const g_out = (T*)p.out;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* const sA = lds;                        // [2 planes][hh rows at p.pitch][hw pixels][PF_LD]
    unsigned char* const sB = lds + 2 * PF_A_PLANE;       // ring of PF_RING tap stages
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware order (speed only): a contiguous chunk of the (tile, group) space per XCD, the groups of a tile next to each other
    int tile, ng;
    {
        const int total = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = total >> 3, r = total & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        tile = id / p.ngroups;
        ng = id - tile * p.ngroups;
    }
    const int n = tile / (p.tiles_y * p.tiles_x);
    const int tr = tile - n * p.tiles_y * p.tiles_x;
    const int ty0 = (tr / p.tiles_x) * PF_TH, tx0 = (tr % p.tiles_x) * PF_TW;
    const int n0 = ng * 64;
    const int nchunks = p.Cin >> 5;

    const int ea = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax, p.amax_stripes))), ew = ss_amax_exp(__uint_as_float(p.h_amax2[0]));
    const float a_scale = ldexpf(1.f, 14 - ea);
    const float out_scale = ldexpf(1.f, ea - 14 + ew - 14);

    // ---- A staging slots of this thread: slot e = tid + 256 i  <->  (halo pixel e >> 3, channels 4 (e & 7) .. +3 of the chunk) ----
    const int npix = p.hh * p.hw;
    const int hw_m = (65536 + p.hw - 1) / p.hw;
    int aoff[PF_SLOTS];          // element offset of the pixel's channel quad in chunk 0, -1: outside the image / no slot (reads zeros)
#pragma unroll
    for (int i = 0; i < PF_SLOTS; ++i) {
        const int e = tid + 256 * i;
        const int pix = e >> 3, c4 = e & 7;
        const int hy = (pix * hw_m) >> 16, hx = pix - hy * p.hw;          // pix < 512, hw in 16 .. 18: exact
        const int iy = ty0 + p.in_oy + p.hy0 + hy, ix = tx0 + p.in_ox + p.hx0 + hx;
        const bool slot = pix < npix;
        const bool ok = slot && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        aoff[i] = ok ? ((n * p.IH + iy) * p.IW + ix) * p.in_cs + c4 * 4 : -1;
    }
    f32x4 ra[PF_SLOTS];
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF_SLOTS; ++i) {
            const T* q = (aoff[i] >= 0 && !(p.dbg & 1)) ? g_in + aoff[i] + chunk * 32 : (const T*)pf_zero_page16;
            ra[i] = pf_ld4(q);
        }
    };
    auto store_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PF_SLOTS; ++i) {
            const int e = tid + 256 * i;
            if ((e >> 3) >= npix) continue;
            const int pix = e >> 3;
            const int hy = (pix * hw_m) >> 16, hx = pix - hy * p.hw;
            const int ao = hy * p.pitch + hx * PF_LD + (e & 7) * 8;
            unsigned int hh[2], ll[2];
            ss_split_h2(ra[i][0] * a_scale, ra[i][1] * a_scale, hh[0], ll[0]);
            ss_split_h2(ra[i][2] * a_scale, ra[i][3] * a_scale, hh[1], ll[1]);
            *(u32x2*)(sA + ao) = u32x2{hh[0], hh[1]};
            if (F32) *(u32x2*)(sA + PF_A_PLANE + ao) = u32x2{ll[0], ll[1]};          // 16-bit storage: h is the (scaled) stored value, exactly
        }
    };

    // ---- B (weight planes) by LDS-DMA: a tap and chunk = 8 pieces of 16 rows x 64 B (2 planes x 4 row blocks); this wave: pieces 2 wave, 2 wave + 1.
    //      Lane (row = lane >> 2 of the block, 16-byte slot lane & 3) fetches k-octet slot ^ ((row >> 2) & 3): the XOR swizzle of gemm_x6p.hip.
    int b_off[2], b_dst[2];          // element offset of this lane's 16 bytes inside the planes at k = 0; LDS offset of the piece inside a tap stage
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave * 2 + j;
        const int pl = q >> 2, rb = q & 3;
        const int row = rb * 16 + (lane >> 2);
        const int ko = (lane & 3) ^ ((row >> 2) & 3);
        b_off[j] = (int)(pl * p.plane_elems) + (n0 + row) * p.KT + 8 * ko;          // (the launcher checks: the planes are below 2^31 elements)
        b_dst[j] = pl * 4096 + rb * 1024;
    }
    // (argument arrays are read with compile-time indices only: a run-time index makes the compiler copy the struct to scratch memory)
    int nt[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) nt[ph] = p.ntaps[ph];
    const int NTAPS = nt[0] + nt[1] + nt[2] + nt[3];
    int a_shift[4][4];          // byte offset of tap (ph, t) inside the A stage
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int t = 0; t < 4; ++t) a_shift[ph][t] = p.tdy[ph][t] * p.pitch + p.tdx[ph][t] * PF_LD;
    int sp_sh[2][8];          // SPLIT: byte offset of the tap of (wave pair, streamed pair) inside the A stage
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
        for (int st = 0; st < 8; ++st) sp_sh[w2][st] = SPLIT ? p.sp_shift[w2][st] : 0;
    // Requests go out in PAIRS of taps (one barrier per pair = 24 MFMAs per wave, as gconv_x6v2): the NEXT pair starts at the even flattened
    // index d_g of chunk d_c and lands in ring stages d_stage, d_stage + 1.  A chunk with an odd tap count ends on a half pair whose second
    // request repeats the first tap (harmless: nobody reads that stage); past the last chunk the requests repeat its taps.  Every wave
    // issues exactly four DMA instructions per call, so the vmcnt bookkeeping below is the same for every wave and step.
    int d_g = 0, d_c = 0, d_stage = 0;
    auto dma_tap = [&](int g, int stage) __attribute__((always_inline)) {
        const int c = d_c < nchunks ? d_c : nchunks - 1;
        const int koff = g * p.Cin + c * 32;          // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (!(p.dbg & 2) && (NPROD >= 2 || wave < 2)) pf_dma16(p.planes + b_off[j] + koff, sB + stage * PF_B_TAP + b_dst[j]);          // (waves 2, 3 carry the low weight plane)
    };
    auto dma_pair = [&]() __attribute__((always_inline)) {
        dma_tap(d_g, d_stage);
        dma_tap(d_g + 1 < NTAPS ? d_g + 1 : d_g, d_stage + 1);
        d_stage = (d_stage + 2) & (PF_RING - 1);
        d_g += 2;
        if (d_g >= NTAPS) { d_g = 0; ++d_c; }
    };

    const int wp = wave >> 1, wi = wave & 1;          // SPLIT: wave pair (phases), pixel half
    // SPLIT: the rest of the kernel exists once per wave pair (one uniform branch at the very end of this function picks the copy): each copy
    // is straight-line code with its own accumulators -- a branch per streamed pair, or per chunk with the accumulators live across it, made the
    // register allocator spill accumulator tiles.  Both copies execute the same barriers.
    auto body = [&](auto wpc) __attribute__((always_inline)) {
    constexpr int WP = decltype(wpc)::value;
    f32x16 acc[8];          // [phase][channel block]; SPLIT: [local phase][pixel block][channel block]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int row0 = SPLIT ? 4 * wi : 2 * wave;       // first class-pixel row of this wave inside the tile

    // fragment addresses.  A: this lane's class pixel (row 2 wave + (l31 >> 4), column l31 & 15) at halo coordinates (- hy0, - hx0),
    // k-octet lh (+ 2 ks); a tap adds the uniform offset dy_t * pitch + dx_t * PF_LD.  B: row ni * 32 + l31, slot (lh + 2 ks) ^ ((row >> 2) & 3).
    const int a_lane = (row0 + (l31 >> 4) - p.hy0) * p.pitch + ((l31 & 15) - p.hx0) * PF_LD + lh * 16;
    const int a_mi = 2 * p.pitch;          // SPLIT: second pixel block = two rows down
    const int sw = (l31 >> 2) & 3;
    const int so0 = (lh ^ sw) << 4, so1 = so0 ^ 32;
    const int b_lane = l31 * 64;

    // prologue: first chunk's input tile in registers, the first two taps' weight planes requested
    load_a(0);
    __builtin_amdgcn_sched_barrier(0);
    dma_pair();
    __builtin_amdgcn_sched_barrier(0);
    int r_stage = 0;          // ring stage of the tap being multiplied

    for (int c = 0; c < nchunks; ++c) {
        // every wave is done with the previous chunk's A stage (it passed the last tap's MFMAs); the registers hold chunk c
        if (c > 0) {
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): own fragment reads done (no vector-memory drain: the DMA stays in flight)
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        store_a();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = c + 1 < nchunks;
        if (more) load_a(c + 1);          // into the same registers: the stores above have consumed them
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SPLIT != 0) {
            // Straight-line code per wave pair (one uniform branch per chunk; the barriers are the same ones): the local phase of every
            // streamed pair is a compile-time constant, so are the accumulator indices.  (A per-pair branch on a run-time local phase made
            // the register allocator spill one accumulator tile per pair.)
            {
                auto stage = [&](auto stc) __attribute__((always_inline)) {
                    constexpr int st = decltype(stc)::value;
                    if constexpr (st < pf_split_stages(SPLIT)) {
                    // the planes of this streamed pair have landed (see the plain form below for the counts)
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && st == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);          // vmcnt(6)
                    else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    if (!(p.dbg & 8)) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    dma_pair();
                    __builtin_amdgcn_sched_barrier(0);
                    constexpr int lp = pf_split_lp(SPLIT, WP, st);
                    if constexpr (lp >= 0) {
                        constexpr int base = lp * 4;
                        const unsigned char* const ap = sA + a_lane + sp_sh[WP][st];
                        const unsigned char* const bp = sB + ((r_stage + WP) & (PF_RING - 1)) * PF_B_TAP + b_lane;
#pragma unroll
                        for (int ks = 0; ks < ((p.dbg & 4) ? 0 : 2); ++ks) {
                            f16x8 ah[2], al[2];
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) {
                                ah[mi] = *(const f16x8*)(ap + mi * a_mi + ks * 32);
                                al[mi] = ah[mi];
                                if (F32) al[mi] = *(const f16x8*)(ap + mi * a_mi + PF_A_PLANE + ks * 32);
                            }
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                const unsigned char* bq = bp + ni * 2048 + (ks ? so1 : so0);
                                const f16x8 bh = *(const f16x8*)bq;
                                f16x8 bl = bh;
                                if (NPROD >= 2) bl = *(const f16x8*)(bq + 4096);
#pragma unroll
                                for (int mi = 0; mi < 2; ++mi) {
                                    if (NPROD == 3) acc[base + mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh, acc[base + mi * 2 + ni], 0, 0, 0);
                                    if (NPROD >= 2) acc[base + mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl, acc[base + mi * 2 + ni], 0, 0, 0);
                                    acc[base + mi * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh, acc[base + mi * 2 + ni], 0, 0, 0);
                                }
                            }
                        }
                    }
                    r_stage = (r_stage + 2) & (PF_RING - 1);
                    }
                };
                stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
                stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
                stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{});
            }
        } else {
        int g = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= nt[ph]) continue;
                if ((g & 1) == 0) {
                    // the planes of this pair of taps have landed: requests are answered in order, and behind this pair's four DMA
                    // instructions only the six loads of the next chunk's input tile can be in flight (first pair of a chunk)
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && g == 0) __builtin_amdgcn_s_waitcnt(0x0F70 | 6);          // vmcnt(6)
                    else __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0)
                    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): own LDS stores (A stage) / fragment reads of the previous pair
                    if (!(p.dbg & 8)) __builtin_amdgcn_s_barrier();          // ... for every wave: the pair's planes and the A stage are visible, the stages the next request overwrites are free
                    __builtin_amdgcn_sched_barrier(0);
                    dma_pair();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const unsigned char* const ap = sA + a_lane + a_shift[ph][t];
                const unsigned char* const bp = sB + r_stage * PF_B_TAP + b_lane;
#pragma unroll
                for (int ks = 0; ks < ((p.dbg & 4) ? 0 : 2); ++ks) {
                    const f16x8 ah = *(const f16x8*)(ap + ks * 32);
                    f16x8 al = ah;
                    if (F32) al = *(const f16x8*)(ap + PF_A_PLANE + ks * 32);
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const unsigned char* bq = bp + ni * 2048 + (ks ? so1 : so0);
                        const f16x8 bh = *(const f16x8*)bq;
                        if (NPROD == 3) acc[ph * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[ph * 2 + ni], 0, 0, 0);
                        if (NPROD >= 2) {
                            const f16x8 bl = *(const f16x8*)(bq + 4096);
                            acc[ph * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[ph * 2 + ni], 0, 0, 0);
                        }
                        acc[ph * 2 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[ph * 2 + ni], 0, 0, 0);
                    }
                }
                r_stage = (r_stage + 1) & (PF_RING - 1);
                ++g;
            }
        }
        r_stage = (r_stage + (g & 1)) & (PF_RING - 1);          // an odd tap count: skip the half pair's second stage
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no DMA may still be landing when the workgroup ends

    // ---- epilogue.  C/D layout of the 32x32 MFMA: column (output channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
    // stored directly, an instruction writes 2 x 128 bytes and a wave issues 128 of them (measured: 1074 MB of output at 2.4 TB/s, 40 % of
    // the kernel).  Each wave passes its 32 pixels x 64 channels of a phase through a private LDS block (the operand stages are free now)
    // and stores 16 bytes per lane: 16 lanes = one pixel's 256 bytes, 4 pixels per instruction, 8 instructions per phase.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();          // every wave is done with the operand stages
    constexpr int ES = 68;                 // floats per pixel row of the scratch (64 + 4: the 16-byte reads of 4 pixels hit different banks)
    float* const tb = (float*)(lds + wave * (32 * ES * 4));
    const bool vec_ok = n0 + 64 <= p.Cout && (p.out_cs & 3) == 0 && (((uintptr_t)p.out) & (4 * sizeof(T) - 1)) == 0;
    const int q4 = (lane & 15) * 4, prow = lane >> 4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && vec_ok) b4 = *(const f32x4*)(p.bias + n0 + q4);
    const bool plain = p.act == SS_ACT_NONE && !p.accumulate;
    // one block of 32 pixels x 64 channels: accumulators acc[base], acc[base + 1], pixel rows r0 + {0, 1} of the tile, output sub-pixel (ooy, oox)
    auto epilogue = [&](auto basec, const int r0, const int ooy, const int oox, const bool has) __attribute__((always_inline)) {
        constexpr int base = decltype(basec)::value;          // (compile-time: the accumulators stay in registers)
        if (!has || (p.dbg & 16)) return;
        if (vec_ok) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + ni * 32 + l31] = acc[base + ni][r] * out_scale;
            __builtin_amdgcn_wave_barrier();          // one wave's LDS operations execute in issue order
            // (one uniform branch picks the plain form -- no activation, no accumulation: every layer that is followed by a norm -- whose
            // store loop holds no case analysis; the generic form keeps the run-time switches)
            if (plain) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = 4 * k + prow;
                    f32x4 v = *(const f32x4*)(tb + m * ES + q4);
                    const int yc = ty0 + r0 + (m >> 4), xc = tx0 + (m & 15);
                    const int oy = yc * 2 + ooy, ox = xc * 2 + oox;
                    if (yc >= p.OHc || xc >= p.OWc || oy >= p.OH || ox >= p.OW) continue;
                    T* op = g_out + ((n * p.OH + oy) * p.OW + ox) * p.out_cs + n0 + q4;          // (the launcher checks: below 2^31 elements)
                    pf_st4(op, v + b4);
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < 8; ++k) {
                    const int m = 4 * k + prow;
                    f32x4 v = *(const f32x4*)(tb + m * ES + q4);
                    const int yc = ty0 + r0 + (m >> 4), xc = tx0 + (m & 15);
                    const int oy = yc * 2 + ooy, ox = xc * 2 + oox;
                    if (yc >= p.OHc || xc >= p.OWc || oy >= p.OH || ox >= p.OW) continue;
                    T* op = g_out + ((n * p.OH + oy) * p.OW + ox) * p.out_cs + n0 + q4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ss_apply_act(v[e] + b4[e], p.act, p.alpha);
                    if (p.accumulate) v += pf_ld4(op);
                    pf_st4(op, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
            return;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int co = n0 + ni * 32 + l31;
            if (co >= p.Cout) continue;
            const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int yc = ty0 + r0 + (m >> 4), xc = tx0 + (m & 15);
                if (yc >= p.OHc || xc >= p.OWc) continue;
                const int oy = yc * 2 + ooy, ox = xc * 2 + oox;
                if (oy >= p.OH || ox >= p.OW) continue;
                T* op = g_out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs + co;
                float v = ss_apply_act(acc[base + ni][r] * out_scale + bv, p.act, p.alpha);
                if (p.accumulate) v += (float)*op;
                *op = (T)v;
            }
        }
    };
    if constexpr (SPLIT != 0) {
        const int oy0 = p.sp_ooy[WP][0], ox0 = p.sp_oox[WP][0];
        const int oy1 = p.sp_ooy[WP][1], ox1 = p.sp_oox[WP][1];
        epilogue(std::integral_constant<int, 0>{}, row0, oy0, ox0, true);
        epilogue(std::integral_constant<int, 2>{}, row0 + 2, oy0, ox0, true);
        epilogue(std::integral_constant<int, 4>{}, row0, oy1, ox1, true);
        epilogue(std::integral_constant<int, 6>{}, row0 + 2, oy1, ox1, true);
    } else {
        epilogue(std::integral_constant<int, 0>{}, row0, p.out_oy[0], p.out_ox[0], nt[0] != 0);
        epilogue(std::integral_constant<int, 2>{}, row0, p.out_oy[1], p.out_ox[1], nt[1] != 0);
        epilogue(std::integral_constant<int, 4>{}, row0, p.out_oy[2], p.out_ox[2], nt[2] != 0);
        epilogue(std::integral_constant<int, 6>{}, row0, p.out_oy[3], p.out_ox[3], nt[3] != 0);
    }
    };          // body
    if constexpr (SPLIT != 0) {
        if (wp == 0) body(std::integral_constant<int, 0>{});
        else body(std::integral_constant<int, 1>{});
    } else {
        body(std::integral_constant<int, 0>{});
    }
}

}  // namespace

// The `count` problems are the sub-pixel phases of ONE stride-2 data gradient / transposed convolution as conv_bwd_data builds them
// (same input, output, class grid; out_oy / out_ox and the taps differ) and the fused kernel takes them.
bool ss_gconv_phases_fused_ok(const GConvParams* ps, int count) {
    if (!ss_tuning().phases_fused || count != 4) return false;
    const GConvParams& p0 = ps[0];
    if (p0.c1_dtype != SS_DTYPE_F32 || !p0.h_amax || !p0.h_amax2 || p0.in_s != 1 || p0.out_s != 2 || p0.nbatch > 1 || p0.reflect ||
        p0.Cin % 32 || p0.Cin < 64 || p0.Cout % 64 || (p0.in_cs & 3) || (((uintptr_t)p0.in) & (p0.dtype == SS_DTYPE_F32 ? 15 : 7)) || p0.stats)
        return false;
    if ((long)p0.N * p0.IH * p0.IW * p0.in_cs >= (1L << 31) || (long)p0.N * p0.OH * p0.OW * p0.out_cs >= (1L << 31)) return false;
    int y0 = 1 << 20, y1 = -(1 << 20), x0 = 1 << 20, x1 = -(1 << 20), taps = 0;
    for (int i = 0; i < count; ++i) {
        const GConvParams& q = ps[i];
        // (class grids may differ by one row / column between the phases of an odd-sized output: the kernel walks the largest and masks)
        if (q.ntaps < 1 || q.ntaps > SS_MAX_PHASE_TAPS || abs(q.OHc - p0.OHc) > 1 || abs(q.OWc - p0.OWc) > 1 || q.N != p0.N || q.Cin != p0.Cin || q.Cout != p0.Cout || q.in != p0.in ||
            q.out != p0.out || q.in_s != p0.in_s || q.out_s != p0.out_s || q.in_oy != p0.in_oy || q.in_ox != p0.in_ox || q.nbatch != p0.nbatch || q.h_amax != p0.h_amax ||
            q.h_amax2 != p0.h_amax2 || q.accumulate != p0.accumulate || q.act != p0.act || q.dtype != p0.dtype || q.stats || q.bias != p0.bias || q.out_oy < 0 ||
            q.out_oy > 1 || q.out_ox < 0 || q.out_ox > 1)
            return false;
        for (int t = 0; t < q.ntaps; ++t) {
            const int dy = q.taps[t].dy, dx = q.taps[t].dx;
            y0 = dy < y0 ? dy : y0; y1 = dy > y1 ? dy : y1; x0 = dx < x0 ? dx : x0; x1 = dx > x1 ? dx : x1;
        }
        taps += q.ntaps;
    }
    if (y1 - y0 > 2 || x1 - x0 > 2) return false;
    // (no floor on the workgroup count: measured faster than four per-phase launches from n = 1 up -- 132 -> 80 us for the generators' last
    // transposed convolution at 256 x 256 -- where the launches themselves are what costs)
    int ohc = 0, owc = 0;
    for (int i = 0; i < count; ++i) { ohc = ps[i].OHc > ohc ? ps[i].OHc : ohc; owc = ps[i].OWc > owc ? ps[i].OWc : owc; }
    const long tiles = (long)p0.N * ((ohc + PF_TH - 1) / PF_TH) * ((owc + PF_TW - 1) / PF_TW) * (p0.Cout / 64);
    const long floor_wgs = ss_tuning().phases_fused >= 2 ? ss_tuning().phases_fused : 1;          // phases_fused = n >= 2: a floor on the workgroup count (measurement)
    return tiles >= floor_wgs && tiles < (1L << 30) && taps >= 4;
}

namespace {
// SPLIT form: the four phase slots in two pairs whose tap counts differ by at most one (the longer first); flattened tap 2 s + wp = tap s
// of pair wp (its first phase's taps, then its second's)
struct PFSplit { int slot[2][2]; int n[2][2]; int stages; int pat; const GConvParams* q[4]; };
bool pf_split_schedule(const GConvParams* ps, PFSplit* sc) {
    if (!ss_tuning().phases_split) return false;
    const GConvParams* q[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) {
        const int k = 2 * ps[i].out_oy + ps[i].out_ox;
        if (k < 0 || k > 3 || q[k]) return false;
        q[k] = &ps[i];
    }
    static const int pairs[3][4] = {{0, 3, 1, 2}, {0, 1, 2, 3}, {0, 2, 1, 3}};
    int best = -1, bestd = 1 << 20;
    for (int i = 0; i < 3; ++i) {
        const int a = q[pairs[i][0]]->ntaps + q[pairs[i][1]]->ntaps, b = q[pairs[i][2]]->ntaps + q[pairs[i][3]]->ntaps;
        const int d = a > b ? a - b : b - a;
        if (d < bestd) { bestd = d; best = i; }
    }
    if (best < 0 || bestd > 1) return false;
    const int a = q[pairs[best][0]]->ntaps + q[pairs[best][1]]->ntaps, b = q[pairs[best][2]]->ntaps + q[pairs[best][3]]->ntaps;
    const int first = a >= b ? 0 : 2;
    for (int wp = 0; wp < 2; ++wp) {
        int s0 = pairs[best][(wp == 0 ? first : 2 - first)], s1 = pairs[best][(wp == 0 ? first : 2 - first) + 1];
        if (q[s0]->ntaps > q[s1]->ntaps) { const int t = s0; s0 = s1; s1 = t; }          // the shorter phase first
        sc->slot[wp][0] = s0; sc->slot[wp][1] = s1;
        sc->n[wp][0] = q[s0]->ntaps; sc->n[wp][1] = q[s1]->ntaps;
    }
    sc->stages = sc->n[0][0] + sc->n[0][1];
    for (int k = 0; k < 4; ++k) sc->q[k] = q[k];
    // the kernel's compile-time patterns (pf_split_lp)
    const int (&m)[2][2] = sc->n;
    if (m[0][0] == 1 && m[0][1] == 4 && m[1][0] == 2 && m[1][1] == 2) sc->pat = 1;
    else if (m[0][0] == 4 && m[0][1] == 4 && m[1][0] == 4 && m[1][1] == 4) sc->pat = 2;
    else return false;
    return sc->stages == pf_split_stages(sc->pat);
}
// tap s of pair wp: (phase problem, tap index), false past the pair's end
bool pf_split_tap(const PFSplit& sc, int wp, int s, const GConvParams** q, int* t, int* lp) {
    if (s < sc.n[wp][0]) { *q = sc.q[sc.slot[wp][0]]; *t = s; *lp = 0; return true; }
    if (s < sc.n[wp][0] + sc.n[wp][1]) { *q = sc.q[sc.slot[wp][1]]; *t = s - sc.n[wp][0]; *lp = 1; return true; }
    return false;
}
}  // namespace

// The weight planes of the fused kernel: ONE problem whose tap list is the phases' taps one after the other, phase slot 2 ry + rx major
// (the order of the kernel's multiply loop) -- what ss_launch_wprep_x6 / ss_gconv_x6_planes_bytes take.
bool ss_gconv_phases_fused_wprob(const GConvParams* ps, int count, GConvParams* w) {
    if (!ss_gconv_phases_fused_ok(ps, count)) return false;
    *w = ps[0];
    w->ntaps = 0;
    PFSplit sc;
    if (pf_split_schedule(ps, &sc)) {          // the SPLIT form's stream order (a different tap list = a different cache identity)
        for (int st = 0; st < sc.stages; ++st)
            for (int wp = 0; wp < 2; ++wp) {
                const GConvParams* q; int t, lp;
                if (!pf_split_tap(sc, wp, st, &q, &t, &lp)) continue;
                if (w->ntaps >= SS_MAX_TAPS) return false;
                w->taps[w->ntaps++] = q->taps[t];
            }
        return true;
    }
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i) {
            if (2 * ps[i].out_oy + ps[i].out_ox != k) continue;
            for (int t = 0; t < ps[i].ntaps; ++t) {
                if (w->ntaps >= SS_MAX_TAPS) return false;
                w->taps[w->ntaps++] = ps[i].taps[t];
            }
        }
    return true;
}

int ss_launch_gconv_phases_fused(const GConvParams* ps, const unsigned short* planes, int count, hipStream_t s) {
    if (!ss_gconv_phases_fused_ok(ps, count)) return SS_ERR_UNSUPPORTED;
    const GConvParams& p0 = ps[0];
    PFParams f{};
    f.in = p0.in; f.out = p0.out; f.bias = p0.bias;
    f.N = p0.N; f.IH = p0.IH; f.IW = p0.IW; f.Cin = p0.Cin; f.in_cs = p0.in_cs;
    f.OH = p0.OH; f.OW = p0.OW; f.Cout = p0.Cout; f.out_cs = p0.out_cs;
    f.OHc = 0; f.OWc = 0;
    for (int i = 0; i < 4; ++i) { f.OHc = ps[i].OHc > f.OHc ? ps[i].OHc : f.OHc; f.OWc = ps[i].OWc > f.OWc ? ps[i].OWc : f.OWc; }
    f.in_oy = p0.in_oy; f.in_ox = p0.in_ox;
    f.act = p0.act; f.alpha = p0.alpha; f.accumulate = p0.accumulate;
    f.h_amax = p0.h_amax; f.h_amax2 = p0.h_amax2; f.amax_stripes = p0.amax_stripes;
    int y0 = 1 << 20, y1 = -(1 << 20), x0 = 1 << 20, x1 = -(1 << 20), taps = 0;
    f.Npad = ss_x6_npad(p0.Cout);
    for (int i = 0; i < 4; ++i) {
        // phase slot = 2 * out_oy + out_ox (any order of the caller's list)
        const GConvParams& q = ps[i];
        const int k = 2 * q.out_oy + q.out_ox;
        if (f.ntaps[k] != 0) return SS_ERR_UNSUPPORTED;
        f.ntaps[k] = q.ntaps; f.out_oy[k] = q.out_oy; f.out_ox[k] = q.out_ox;
        for (int t = 0; t < q.ntaps; ++t) {
            const int dy = q.taps[t].dy, dx = q.taps[t].dx;
            f.tdy[k][t] = (short)dy; f.tdx[k][t] = (short)dx;
            y0 = dy < y0 ? dy : y0; y1 = dy > y1 ? dy : y1; x0 = dx < x0 ? dx : x0; x1 = dx > x1 ? dx : x1;
        }
        taps += q.ntaps;
    }
    f.planes = planes;
    f.KT = taps * p0.Cin;
    f.plane_elems = (long)f.Npad * f.KT;
    if (2 * f.plane_elems >= (1L << 31)) return SS_ERR_UNSUPPORTED;
    f.hy0 = y0; f.hx0 = x0; f.hh = PF_TH + (y1 - y0); f.hw = PF_TW + (x1 - x0);
    f.pitch = (f.hw * PF_LD + 255) / 256 * 256;
    f.tiles_y = (f.OHc + PF_TH - 1) / PF_TH; f.tiles_x = (f.OWc + PF_TW - 1) / PF_TW;
    f.ngroups = p0.Cout / 64;
    const long nwg = (long)p0.N * f.tiles_y * f.tiles_x * f.ngroups;
    char pname[64];
    const long M = (long)p0.N * f.OHc * f.OWc;
    if (getenv("SS_PROF_SHAPES")) snprintf(pname, sizeof(pname), "gconv_phases_fused M%ld N%d K%dx%d", M, p0.Cout, taps, p0.Cin);
    else snprintf(pname, sizeof(pname), "gconv_phases_fused_kernel");
    const int esz = p0.dtype == SS_DTYPE_F32 ? 4 : 2;
    SsProfScope prof(pname, 2.0 * M * p0.Cout * taps * p0.Cin * (p0.dtype == SS_DTYPE_F32 ? 3 : (ss_tuning().wino16_products == 3 ? 2 : 1)),
                     (double)esz * ((double)p0.N * p0.IH * p0.IW * p0.Cin + 4.0 * M * p0.Cout) + 4.0 * taps * p0.Cin * p0.Cout, s);
    f.dbg = ss_tuning().tile_dbg;
    PFSplit sc;
    const bool split = pf_split_schedule(ps, &sc);
    if (split) {
        f.sp_stages = sc.stages;
        for (int wp = 0; wp < 2; ++wp) {
            for (int st = 0; st < 8; ++st) {
                const GConvParams* q; int t, lp;
                if (st < sc.stages && pf_split_tap(sc, wp, st, &q, &t, &lp)) {
                    f.sp_shift[wp][st] = (short)(q->taps[t].dy * f.pitch + q->taps[t].dx * PF_LD);
                    f.sp_lp[wp][st] = (signed char)lp;
                } else {
                    f.sp_shift[wp][st] = 0;
                    f.sp_lp[wp][st] = -1;
                }
            }
            for (int lp = 0; lp < 2; ++lp) { f.sp_ooy[wp][lp] = sc.slot[wp][lp] >> 1; f.sp_oox[wp][lp] = sc.slot[wp][lp] & 1; }
        }
    }
    auto go = [&](auto tc, auto npc, auto spc) {
        typedef decltype(tc) T;
        constexpr int NPROD = decltype(npc)::value;
        constexpr int SPLIT = decltype(spc)::value;
        static const bool attr_set = [] {
            (void)hipFuncSetAttribute((const void*)gconv_phases_fused_kernel<T, NPROD, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return true;
        }();
        (void)attr_set;
        hipLaunchKernelGGL((gconv_phases_fused_kernel<T, NPROD, SPLIT>), dim3((unsigned)nwg), dim3(256), PF_SMEM, s, f);
    };
    auto go2 = [&](auto tc, auto npc) {
        if (split && sc.pat == 1) go(tc, npc, std::integral_constant<int, 1>{});
        else if (split) go(tc, npc, std::integral_constant<int, 2>{});
        else go(tc, npc, std::integral_constant<int, 0>{});
    };
    const bool two = ss_tuning().wino16_products == 3;          // "fp32-grade arithmetic, only the storage is 16-bit"
    if (p0.dtype == SS_DTYPE_F32) go2(0.f, std::integral_constant<int, 3>{});
    else if (p0.dtype == SS_DTYPE_F16) { if (two) go2((_Float16)0, std::integral_constant<int, 2>{}); else go2((_Float16)0, std::integral_constant<int, 1>{}); }
    else { if (two) go2((__bf16)0, std::integral_constant<int, 2>{}); else go2((__bf16)0, std::integral_constant<int, 1>{}); }
    SS_LAUNCH_CHECK();
    return SS_OK;
}
