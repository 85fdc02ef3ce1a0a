// Weight gradient of the stride-2 gather layers (3 x 3 down / transposed up convolutions of the generators, 4 x 4 PatchGAN layers) with the
// operands STAGED ONCE per spatial tile and every tap served from LDS.
//
//     part[split][(t, ca)][cb] = sum over the split's output pixels of  x[n, 2 y + oy + dy_t, 2 x + ox + dx_t, ca] * dy[n, y, x, cb]
//
// Why: wgrad_x6_kernel (conv_mfma_x6.hip) is an implicit GEMM over M = (tap, channel) tiles of 128 rows; every M tile gathers its own
// copy of the input rows and re-reads dy -- 2.7 GB of L2 -> CU traffic for 0.8 GB of operands on the 64 -> 128 layer, and the kernel is
// paced by exactly that (tools/wgrad_phase_probe.py: loads alone 309 of 449 us).  Here a workgroup owns a block of 32 input channels x
// CBB output channels for ALL taps: per 4 x 16 tile of output pixels (K = 64) it stages the (6 + kh) x (30 + kw) input halo tile and
// the dy tile ONCE -- fp32 from global memory, split into the two fp16 pieces of the x3h arithmetic (per-tensor power-of-two scales,
// as wgrad_x6_kernel), stored in the operands' natural image [pixel][channel] -- and forms every tap's A fragment with gfx950's
// transposing LDS read (ds_read_b64_tr_b16: gemm_tn_x3h.hip) at a tap-dependent row offset.  The halo tile is stored with the input
// columns de-interleaved by parity, so that the 4 consecutive output pixels a 16-lane group addresses are 4 consecutive 64-byte rows
// whatever the tap (conflict-free); the dy rows' 64-byte segments are XOR-swizzled with the pixel index (as gemm_tn_x3h).
// 512 threads = 8 waves: wave -> (32-column tile of the CBB output channels, a residue class of taps); one B fragment per K = 16 step
// serves all of the wave's taps.  Global loads of tile i + 1 are in flight (registers) while tile i is multiplied, and split / stored into the second LDS stage behind a wave's own
// MFMAs: one barrier per tile.
// Splits (contiguous tile ranges) write partials; ss_launch_wgrad_reduce sums them in fixed order (deterministic).
//
// 16-bit activation storage (T = _Float16 / __bf16): the stored values ARE the leading fp16 piece (under the tensor's power-of-two
// scale; bf16's 8 significand bits fit fp16's 11), so ONE plane per operand, ONE product = the exact product of the stored values (as
// wgrad_x6_kernel does for these types): half the global bytes, half the LDS, a third of the matrix work, no split arithmetic.
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));

// storage type -> the 4-channel unit as loaded, the number of operand planes
template <typename T> struct Stg;
template <> struct Stg<float> { typedef f32x4 V; static constexpr int NPL = 2; };
template <> struct Stg<_Float16> { typedef wf16x4 V; static constexpr int NPL = 1; };
template <> struct Stg<__bf16> { typedef wbf16x4 V; static constexpr int NPL = 1; };
__device__ __forceinline__ f32x4 to_f32x4(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 to_f32x4(wf16x4 v) { return __builtin_convertvector(v, f32x4); }
__device__ __forceinline__ f32x4 to_f32x4(wbf16x4 v) { return __builtin_convertvector(v, f32x4); }
// two scaled values -> one packed fp16 pair (round to nearest even; exact for stored 16-bit values in fp16's normal range)
__device__ __forceinline__ unsigned int pack_h2(float x0, float x1) {
    const ss_f2 v = {x0, x1};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ss_h2));
}

constexpr int TH = 4, TW = 16, TK = TH * TW;          // output pixels of a tile = the K of one LDS stage
constexpr int CAB = 32;                               // input channels per workgroup
constexpr int XC2 = TW + 1;                           // entries per (row, parity) of the de-interleaved halo tile
constexpr int XROWB = CAB * 2;                        // bytes per halo entry and plane (32 fp16)

struct WSGeom {
    int kh, kw, R, C;          // tap box; halo rows (6 + kh) and columns (30 + kw)
    int tiles_x, tiles_y, tiles_total, tiles_per_split;
    int nca, ncb;              // channel blocks
};

__device__ __forceinline__ s16x4 tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <int NTAPS, int CBB, typename T = float>
__global__ __launch_bounds__(512, 1) void wgrad_stage_kernel(WGradParams p, WSGeom g) {
    typedef typename Stg<T>::V UV;                    // one unit = 4 channels as stored
    constexpr int NPL = Stg<T>::NPL;                  // operand planes in LDS: (h, l) or h alone
    constexpr int NJ = CBB / 32;                      // 32-column tiles of the output-channel block
    constexpr int NG = 8 / NJ;                        // tap residue classes (waves per column tile)
    constexpr int TPW = (NTAPS + NG - 1) / NG;        // taps per wave
    constexpr int BROWB = CBB * 2;                    // bytes per dy row and plane
    constexpr int XU = NTAPS == 9 ? 5 : 6;            // halo units (16 bytes of fp32 = 4 channels of one entry) per thread: 9 x 33 x 8 <= 5 x 512, 10 x 34 x 8 <= 6 x 512
    constexpr int BU = TK * (CBB / 4) / 512;          // dy units per thread
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int xplane = g.R * 2 * XC2 * XROWB;
    constexpr int bplane = TK * BROWB;
    const int stage_b = NPL * xplane + NPL * bplane;  // TWO stages: tile i + 1 is split and stored while other waves still multiply tile i
    // stage: [2 planes][R][2 parities][XC2][32 ch] fp16 | [2 planes][TK][CBB] fp16, 64-byte segments swizzled with the row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wj = wave % NJ, wg = wave / NJ;

    // workgroup -> (split, channel blocks): the AB workgroups of a split read the same tiles; they are neighbours on ONE XCD (own L2)
    const int AB = g.nca * g.ncb;
    int split, sub;
    {
        const int L = blockIdx.x;
        const int xcd = L & 7, q = L >> 3;
        split = (q / AB) * 8 + xcd;
        sub = q % AB;
    }
    if (split >= p.splits) return;
    const int ca0 = (sub % g.nca) * CAB, cb0 = (sub / g.nca) * CBB;
    const int t_begin = split * g.tiles_per_split;
    const int t_end = t_begin + g.tiles_per_split < g.tiles_total ? t_begin + g.tiles_per_split : g.tiles_total;

    const int ea = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax, p.amax_stripes))), eb = ss_amax_exp(__uint_as_float(ss_amax_load(p.h_amax2, p.amax2_stripes)));
    const float a_scale = ldexpf(1.f, 14 - ea), b_scale = ldexpf(1.f, 14 - eb), out_scale = ldexpf(1.f, ea - 14 + eb - 14);

    // ---- staging: this thread's units.  Halo unit id -> (row r, column c, channel quad cq); dy unit id -> (pixel k, channel quad)
    const int xunits = g.R * g.C * (CAB / 4);
    int xrc[XU], xl[XU];                              // (row << 8) | column, LDS byte offset (h plane); xl < 0: no such unit
#pragma unroll
    for (int i = 0; i < XU; ++i) {
        const int id = tid + 512 * i;
        if (id < xunits) {
            const int e = id >> 3, cq = id & 7;
            const int r = e / g.C, c = e - r * g.C;
            xrc[i] = (r << 8) | c;
            xl[i] = ((r * 2 + (c & 1)) * XC2 + (c >> 1)) * XROWB + cq * 8;
        } else { xrc[i] = 0; xl[i] = -1; }
    }
    const int xcq = tid & 7;
    // dy unit i of this thread: pixel bk0 + (512 / (CBB / 4)) i (same swizzle term: the step is a multiple of 4), channel quad bq
    constexpr int BSTEP = 512 / (CBB / 4);
    const int bk0 = tid / (CBB / 4), bq = tid % (CBB / 4);
    const int bl0 = bk0 * BROWB + (((bq >> 3) ^ (NJ == 4 ? (bk0 & 3) : ((bk0 >> 1) & 1))) << 6) + (bq & 7) * 8;
    const T* const ga = (const T*)p.a + ca0 + 4 * xcq;
    const T* const gb = (const T*)p.b + cb0;

    UV rx[XU], rb[BU];
    auto load_tile = [&](int t) {
        if (p.dbg & 1) {          // measurement: no global loads
#pragma unroll
            for (int i = 0; i < XU; ++i) rx[i] = UV{(T)1.f, (T)2.f, (T)3.f, (T)4.f};
#pragma unroll
            for (int i = 0; i < BU; ++i) rb[i] = UV{(T)1.f, (T)1.f, (T)1.f, (T)1.f};
            return;
        }
        const int tx = t % g.tiles_x;
        const int r0 = t / g.tiles_x;
        const int ty = r0 % g.tiles_y, n = r0 / g.tiles_y;
        const int iy0 = ty * TH * 2 + p.a_oy, ix0 = tx * TW * 2 + p.a_ox;
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            const int iy = iy0 + (xrc[i] >> 8), ix = ix0 + (xrc[i] & 255);
            const bool ok = xl[i] >= 0 && iy >= 0 && iy < p.AH && ix >= 0 && ix < p.AW;
            rx[i] = ok ? *(const UV*)(ga + ((long)(n * p.AH + iy) * p.AW + ix) * p.a_cs) : UV{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
        }
#pragma unroll
        for (int i = 0; i < BU; ++i) {
            const int bk = bk0 + BSTEP * i;
            const int oy = ty * TH + (bk >> 4), ox = tx * TW + (bk & 15);
            // ragged grids ('valid' PatchGAN layers: 255 / 126 / 62 pixels a side): pixels beyond the grid contribute nothing
            const bool ok = oy < p.GH && ox < p.GW;
            rb[i] = ok ? *(const UV*)(gb + ((long)(n * p.GH + oy) * p.GW + ox) * p.b_cs + 4 * bq) : UV{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
        }
    };
    // the split + LDS stores of the NEXT tile (halo units, dy units) into the other stage
    auto store_x = [&](int st) {
        if (p.dbg & 2) return;          // measurement: no split, no LDS stores
        unsigned char* const sx = lds + st * stage_b;
#pragma unroll
        for (int i = 0; i < XU; ++i) {
            if (xl[i] < 0) continue;
            const f32x4 v = to_f32x4(rx[i]);
            if constexpr (NPL == 1) {
                *(u32x2*)(sx + xl[i]) = u32x2{pack_h2(v[0] * a_scale, v[1] * a_scale), pack_h2(v[2] * a_scale, v[3] * a_scale)};
            } else {
                unsigned int h0, l0, h1, l1;
                ss_split_h2s(v[0] * a_scale, v[1] * a_scale, h0, l0);
                ss_split_h2s(v[2] * a_scale, v[3] * a_scale, h1, l1);
                *(u32x2*)(sx + xl[i]) = u32x2{h0, h1};
                *(u32x2*)(sx + xplane + xl[i]) = u32x2{l0, l1};
            }
        }
    };
    auto store_b = [&](int st) {
        if (p.dbg & 2) return;
        unsigned char* const sb = lds + st * stage_b + NPL * xplane;
#pragma unroll
        for (int i = 0; i < BU; ++i) {
            const f32x4 v = to_f32x4(rb[i]);
            if constexpr (NPL == 1) {
                *(u32x2*)(sb + bl0 + i * BSTEP * BROWB) = u32x2{pack_h2(v[0] * b_scale, v[1] * b_scale), pack_h2(v[2] * b_scale, v[3] * b_scale)};
            } else {
                unsigned int h0, l0, h1, l1;
                ss_split_h2s(v[0] * b_scale, v[1] * b_scale, h0, l0);
                ss_split_h2s(v[2] * b_scale, v[3] * b_scale, h1, l1);
                *(u32x2*)(sb + bl0 + i * BSTEP * BROWB) = u32x2{h0, h1};
                *(u32x2*)(sb + bplane + bl0 + i * BSTEP * BROWB) = u32x2{l0, l1};
            }
        }
    };

    // ---- fragment addresses.  Lane l, read rr, K = 16 step ks: pixel k = 16 ks + 8 lh + 4 rr + q, q = (l & 15) >> 2; the lane's 8 bytes are
    // channels 16 ((l >> 4) & 1) + 4 (l & 3) .. + 3 of that pixel's row; the read hands lane i of each 16-lane group channel i of the group
    const int q = (lane & 15) >> 2;
    const int chb = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);          // byte offset of the lane's 4 channels inside a 32-channel row
    // B: row k, logical segment wj (64 bytes = the wave's 32 columns), swizzled with the row
    int boff[2];          // rr = 0, 1 at ks = 0 (ks adds 16 rows: k & 3 and (k >> 1) & 1 keep their values)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int k = 8 * lh + 4 * rr + q;
        const int sw = NJ == 4 ? (k & 3) : ((k >> 1) & 1);
        boff[rr] = k * BROWB + ((wj ^ sw) << 6) + chb;
    }
    // A: pixel k = 16 ty + tx -> halo entry of tap (dy, dx): row 2 ty + dy, parity dx & 1, column tx + (dx >> 1); ks adds one tile row (ty)
    int aoff[TPW][2];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wg + NG * i;
        const int tt = t < NTAPS ? t : 0;
        const int dy = tt / g.kw, dx = tt - dy * g.kw;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int txp = 8 * lh + 4 * rr + q;          // tx of the pixel at ks = 0 (ty = 0)
            aoff[i][rr] = ((dy * 2 + (dx & 1)) * XC2 + txp + (dx >> 1)) * XROWB + chb;
        }
    }
    const int a_ks = 2 * 2 * XC2 * XROWB;          // one tile row further = two halo rows further

    f32x16 acc[TPW], accx[NPL == 2 ? TPW : 1];          // (one plane: no cross terms)
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; if (NPL == 2) accx[i][r] = 0.f; }

    // One tile's matrix work: every tap of this wave against the tile's dy fragment, K = 16 at a time
    auto mma_tile = [&](const unsigned char* sx, const unsigned char* sb) {
#pragma unroll
        for (int ks = 0; ks < TK / 16; ++ks) {
            if (p.dbg & 4) break;          // measurement: no fragment reads, no MFMAs
            f16x8 bh, bl2;
            {
                const unsigned char* pb = sb + ks * 16 * BROWB;
                const s16x4 y0 = tr4(pb + boff[0]), y1 = tr4(pb + boff[1]);
                bh = __builtin_bit_cast(f16x8, __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (NPL == 2) {
                    const s16x4 z0 = tr4(pb + bplane + boff[0]), z1 = tr4(pb + bplane + boff[1]);
                    bl2 = __builtin_bit_cast(f16x8, __builtin_shufflevector(z0, z1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (wg + NG * i >= NTAPS) continue;
                const unsigned char* pa = sx + ks * a_ks;
                const s16x4 x0 = tr4(pa + aoff[i][0]), x1 = tr4(pa + aoff[i][1]);
                const f16x8 ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (NPL == 2) {
                    const s16x4 w0 = tr4(pa + xplane + aoff[i][0]), w1 = tr4(pa + xplane + aoff[i][1]);
                    const f16x8 al = __builtin_bit_cast(f16x8, __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7));
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[i], 0, 0, 0);
                    accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl2, accx[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
            }
        }
    };
    // Tile t + 1 (in this thread's load registers since the previous iteration) -> the other LDS stage, then the loads of tile t + 2.
    auto stage_next = [&](int t, int st) {
        if (t + 1 < t_end) {
            store_x(st ^ 1);
            store_b(st ^ 1);
            if (t + 2 < t_end) load_tile(t + 2);
        }
    };
    // Workgroup barrier that orders LDS traffic only (conv_tile.hip): __syncthreads() would also drain the prefetch loads in flight across it.
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // STAGGERED halves: the two waves of a SIMD (w and w + 4) run a tile's two phases in opposite order -- one multiplies tile t while the
    // other splits / stores tile t + 1, then they swap -- so the matrix pipe works while the VALU splits and vice versa.  In lockstep (all
    // eight waves multiply, then all eight split: wgrad_stage = 2, measurement) the split phase was the exposed term: MFMAs + fragment reads alone
    // 206 us, split + stores alone 91 us, together 330 (64 -> 128 layer, tools/wgrad_phase_probe.py).  Loads are issued one tile further
    // ahead than they are consumed, right after the registers they fill have been stored.
    const bool early = wave >= 4 && !(p.dbg & 16);
    if (t_begin < t_end) {
        load_tile(t_begin);
        store_x(0);
        store_b(0);
        if (t_begin + 1 < t_end) load_tile(t_begin + 1);
    }
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int st = (t - t_begin) & 1;
        const unsigned char* const sx = lds + st * stage_b;
        const unsigned char* const sb = sx + NPL * xplane;
        // (stage st ^ 1 was last read in iteration t - 1: every wave has passed that barrier)
        if (early) stage_next(t, st);
        mma_tile(sx, sb);
        if (!early) stage_next(t, st);
        lds_barrier();
    }

    // ---- partials: part[split][(t, ca)][cb]; the wave's tiles: tap t, rows ca0 + 0..31, columns cb0 + 32 wj + 0..31
    const int M = NTAPS * p.Ca;
    float* const part = p.part + (long)split * M * p.Cb + cb0 + 32 * wj + l31;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wg + NG * i;
        if (t >= NTAPS) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = t * p.Ca + ca0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if constexpr (NPL == 2) part[(long)m * p.Cb] = (acc[i][r] + accx[i][r] * (1.f / 2048.f)) * out_scale;
            else part[(long)m * p.Cb] = acc[i][r] * out_scale;
        }
    }
}

bool geom(const WGradParams& p, WSGeom* g, int* cbb) {
    if (p.ntaps != 9 && p.ntaps != 16) return false;
    const int kw = p.ntaps == 9 ? 3 : 4, kh = kw;
    for (int t = 0; t < p.ntaps; ++t)
        if (p.taps[t].dy != t / kw || p.taps[t].dx != t % kw) return false;          // the full tap box in row-major order
    *cbb = p.ntaps == 9 ? 128 : 64;
    if (p.a_s != 2 || p.reflect || p.nbatch > 1) return false;
    if (p.dtype != SS_DTYPE_F32 && p.dtype != SS_DTYPE_F16 && p.dtype != SS_DTYPE_BF16) return false;
    if (p.Ca % CAB || p.Cb % *cbb || p.GH < TH || p.GW < TW) return false;
    const uintptr_t amask = p.dtype == SS_DTYPE_F32 ? 15 : 7;          // one unit = 4 channels = one 16- / 8-byte access
    if (p.a_cs % 4 || p.b_cs % 4 || (((uintptr_t)p.a) & amask) || (((uintptr_t)p.b) & amask)) return false;
    if ((long)p.N * p.AH * p.AW * p.a_cs >= (1L << 31) || (long)p.N * p.GH * p.GW * p.b_cs >= (1L << 31)) return false;
    g->kh = kh; g->kw = kw;
    g->R = 2 * (TH - 1) + kh;
    g->C = 2 * (TW - 1) + kw;
    g->tiles_x = (p.GW + TW - 1) / TW; g->tiles_y = (p.GH + TH - 1) / TH;
    g->tiles_total = p.N * g->tiles_x * g->tiles_y;
    g->nca = p.Ca / CAB; g->ncb = p.Cb / *cbb;
    return g->R * g->C * (CAB / 4) <= (p.ntaps == 9 ? 5 : 6) * 512 && g->tiles_total >= 512;
}

int splits_of(const WSGeom& g) {
    static const int n_cu = [] { int v = 0; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0); return v >= 8 ? v / 8 * 8 : 256; }();
    const int AB = g.nca * g.ncb;
    int s = n_cu / AB / 8 * 8;          // one workgroup per CU where the channel blocks allow it; a multiple of 8: one share per XCD
    if (s < 8) s = 8;
    while (s > 8 && g.tiles_total / s < 8) s -= 8;
    return s;
}

}  // namespace

bool ss_wgrad_stage_ok(const WGradParams& p) {
    WSGeom g;
    int cbb;
    return ss_tuning().wgrad_stage && p.x6 && ss_x3h_enabled() && geom(p, &g, &cbb);
}

int ss_wgrad_stage_splits(const WGradParams& p) {
    WSGeom g;
    int cbb;
    return geom(p, &g, &cbb) ? splits_of(g) : 0;
}

// partials only (p.part sized for ss_wgrad_stage_splits(p) splits, p.splits = that number, p.h_amax / h_amax2 set)
int ss_launch_wgrad_stage_partials(const WGradParams& p, hipStream_t s) {
    WSGeom g;
    int cbb;
    if (!geom(p, &g, &cbb) || !p.h_amax || !p.h_amax2 || p.splits != splits_of(g)) return SS_ERR_UNSUPPORTED;
    g.tiles_per_split = (g.tiles_total + p.splits - 1) / p.splits;
    const int AB = g.nca * g.ncb;
    const int npl = p.dtype == SS_DTYPE_F32 ? 2 : 1;
    const size_t smem = 2 * ((size_t)npl * g.R * 2 * XC2 * XROWB + (size_t)npl * TK * cbb * 2);          // two stages
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<9, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<9, 128, _Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<16, 64, _Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<9, 128, __bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_stage_kernel<16, 64, __bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const long P = (long)p.N * p.GH * p.GW;
    const bool f32 = p.dtype == SS_DTYPE_F32;
    SsProfScope prof(f32 ? (p.ntaps == 9 ? "wgrad_stage_kernel<9,128>" : "wgrad_stage_kernel<16,64>") : (p.ntaps == 9 ? "wgrad_stage_kernel<9,128,16-bit>" : "wgrad_stage_kernel<16,64,16-bit>"),
                     2.0 * p.ntaps * p.Ca * p.Cb * (double)P * (f32 ? 3 : 1), (f32 ? 4.0 : 2.0) * ((double)p.N * p.AH * p.AW * p.Ca + (double)P * p.Cb), s);
    const unsigned nwg = (unsigned)(p.splits * AB);
    WGradParams pd = p;
    pd.dbg = (ss_tuning().tile_dbg & 7) | (ss_tuning().wgrad_stage == 2 ? 16 : 0);          // measurement only (phase skipping, lockstep phases; 0 in every product path)
    if (p.dtype == SS_DTYPE_F16) {
        if (p.ntaps == 9) hipLaunchKernelGGL((wgrad_stage_kernel<9, 128, _Float16>), dim3(nwg), dim3(512), smem, s, pd, g);
        else hipLaunchKernelGGL((wgrad_stage_kernel<16, 64, _Float16>), dim3(nwg), dim3(512), smem, s, pd, g);
    } else if (p.dtype == SS_DTYPE_BF16) {
        if (p.ntaps == 9) hipLaunchKernelGGL((wgrad_stage_kernel<9, 128, __bf16>), dim3(nwg), dim3(512), smem, s, pd, g);
        else hipLaunchKernelGGL((wgrad_stage_kernel<16, 64, __bf16>), dim3(nwg), dim3(512), smem, s, pd, g);
    } else if (p.ntaps == 9) hipLaunchKernelGGL((wgrad_stage_kernel<9, 128>), dim3(nwg), dim3(512), smem, s, pd, g);
    else hipLaunchKernelGGL((wgrad_stage_kernel<16, 64>), dim3(nwg), dim3(512), smem, s, pd, g);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
