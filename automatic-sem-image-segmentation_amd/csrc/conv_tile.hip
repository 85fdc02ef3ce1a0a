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
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 tr4(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

constexpr int TW = 32;                 // tile width = one 32-row MFMA block per tile row
constexpr int T_THREADS = 256;         // 4 waves
constexpr int MAXQ = 96;               // k-groups (8 channels of one tap) per problem: 9 taps x 64 channels = 72 (+ padding)

struct TileGeom {
    int cg;                 // channel groups of 8 per tap (CinPad / 8)
    int cgp;                // K steps (16 channels = two groups) per tap: (cg + 1) / 2; the odd group of the last step has zero weights
    int nq;                 // k-groups in the weight planes: ntaps * cgp * 2
    int nb;                 // 32-row output-channel blocks
    int pf;                 // float4 staging slots per thread: ceil(hh * hw * 2 * cg / 256)
    unsigned m_c4n, m_cg, m_hw, m_cout, m_pertap;   // ceil(2^32 / d): x / d == umulhi(x, m) for x * d < 2^32 (d = 2*cg, cg, hw, Cout; d == 1 handled apart)
    int wgs_per_cu;
    int stagger;            // start delay between co-resident workgroups, in units of 2048 cycles
    int dbg;                // measurement only ("tile_dbg"): 1 skip global loads, 2 skip the conversion, 4 skip the MFMAs, 8 skip the stores
    int hy0, hx0;           // smallest tap offset (in_oy + dy, in_ox + dx)
    int hh, hw;             // halo tile extents (pixels)
    int psb;                // bytes per staged pixel: cg * 32 (+16 so that it is an odd multiple of 16 -> conflict-free b128 reads)
    int ksb;                // bytes per weight row in LDS: nq * 16 (+16, same rule)
    size_t in_bytes, w_bytes, smem;
};

constexpr int TH = 4;                  // tile rows: one per wave

inline int odd16(int bytes) { return ((bytes / 16) % 2 == 0) ? bytes + 16 : bytes; }
inline unsigned magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + d - 1) / d); }
__device__ __forceinline__ int fdiv(int x, int d, unsigned m) { return d == 1 ? x : (int)__umulhi((unsigned)x, m); }

bool tile_geom(const GConvParams& p, TileGeom* g) {
    if (p.ntaps < 1 || p.nbatch > 1 || p.in_s != 1 || p.out_s != 1 || p.out_oy || p.out_ox || p.OHc != p.OH || p.OWc != p.OW) return false;
    int y0 = 1 << 20, y1 = -(1 << 20), x0 = 1 << 20, x1 = -(1 << 20);
    for (int t = 0; t < p.ntaps; ++t) {
        const int oy = p.in_oy + p.taps[t].dy, ox = p.in_ox + p.taps[t].dx;
        y0 = oy < y0 ? oy : y0; y1 = oy > y1 ? oy : y1; x0 = ox < x0 ? ox : x0; x1 = ox > x1 ? ox : x1;
    }
    if (y1 - y0 > 2 || x1 - x0 > 2) return false;               // 1x1 .. 3x3 footprints
    g->dbg = ss_tuning().tile_dbg;
    g->stagger = ss_tuning().tile_stagger;
    g->cg = (p.Cin + 7) / 8;
    g->cgp = (g->cg + 1) / 2;
    g->nq = p.ntaps * g->cgp * 2;
    g->nb = (p.Cout + 31) / 32;
    if (g->nq > MAXQ || g->nb > 4) return false;
    g->hy0 = y0; g->hx0 = x0;
    g->hw = TW + (x1 - x0);
    g->hh = TH + (y1 - y0);
    g->m_c4n = magic(2 * g->cg); g->m_cg = magic(g->cg); g->m_hw = magic(g->hw); g->m_cout = magic(p.Cout); g->m_pertap = magic(p.Cin * p.Cout);
    g->pf = (g->hh * g->hw * 2 * g->cg + T_THREADS - 1) / T_THREADS;
    if (g->pf > 13) return false;
    g->psb = odd16(g->cg * 32);
    g->ksb = odd16(g->nq * 16);
    g->w_bytes = (size_t)2 * g->nb * 32 * g->ksb;
    g->in_bytes = (size_t)g->hh * g->hw * g->psb;
    g->smem = g->in_bytes + g->w_bytes + 16 + 128 * sizeof(float);
    const int per = (int)((160 * 1024) / (g->smem + 1024));
    const int pfc = g->pf <= 4 ? 4 : (g->pf <= 8 ? 8 : 13);
    const int cap = (g->nb <= 1 && pfc <= 4) ? 4 : 2;           // VGPR budget of the kernel variant: 128 / 256 registers
    g->wgs_per_cu = per > cap ? cap : per;
    return g->wgs_per_cu >= 1;
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

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter (its release fence waits for
// every outstanding global store, and with them, in order, for the NEXT tile's prefetch loads), which serialises the phases of a tile.
// The tile loop's barriers protect LDS reuse alone; global stores and the prefetch stay in flight across them.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// activation storage types (ss_dtype): fp32, fp16, bf16 -- four consecutive channels <-> f32x4
// Four consecutive channels move as ONE global access at ELEMENT alignment (gfx950 global memory takes multi-dword accesses at any
// dword -- for the 16-bit types any 2-byte -- address): channel slices of concatenated tensors (odd offsets, odd strides) stay on the
// wide path instead of four scalar accesses.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_u __attribute__((ext_vector_type(4), aligned(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_u __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ f32x4 ld4(const float* q) { return *(const f32x4_u*)q; }
__device__ __forceinline__ f32x4 ld4(const _Float16* q) { return __builtin_convertvector((f16x4)(*(const f16x4_u*)q), f32x4); }
__device__ __forceinline__ f32x4 ld4(const __bf16* q) { return __builtin_convertvector((bf16x4)(*(const bf16x4_u*)q), f32x4); }
__device__ __forceinline__ void st4(float* q, f32x4 v) { *(f32x4_u*)q = v; }
__device__ __forceinline__ void st4(_Float16* q, f32x4 v) { *(f16x4_u*)q = __builtin_convertvector(v, f16x4); }
__device__ __forceinline__ void st4(__bf16* q, f32x4 v) { *(bf16x4_u*)q = __builtin_convertvector(v, bf16x4); }

// NBT: output-channel blocks held in registers (>= g.nb); PF: staging slots per thread (>= g.pf); NP: piece products (3: x = h + l
// times w = h + l without l*l, fp32 activations; 1: 16-bit activations times the leading weight piece, plain mixed precision)
template <typename TI, typename TO, int NBT, int PF, int NP>
__global__ __launch_bounds__(T_THREADS, ((NBT == 1 && PF <= 4) ? 4 : 2)) void tconv_kernel(GConvParams p, TileGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sIn = smem;                               // [hh*hw pixels][psb]: per channel group 16 B of h then 16 B of l
    unsigned char* sW = smem + g.in_bytes;                   // [2 planes][nb*32][ksb]
    float* red = (float*)(sW + g.w_bytes);                  // [4] block-reduction slots, then [128] the bias vector (zeros without one):
    float* sBias = red + 4;                                  // the epilogue must not issue global loads (they would drain the prefetch)
    const TI* const gin = (const TI*)p.in;
    TO* const gout = (TO*)p.out;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- once per workgroup: the weight tensor -> two fp16 planes in LDS under one power-of-two scale ----
    int ew;
    {
        for (int i = tid; i < (int)(g.w_bytes / 16); i += T_THREADS) *(u32x4*)(sW + (long)i * 16) = u32x4{0u, 0u, 0u, 0u};
        if (tid < 128) sBias[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
        // the (small) weight tensor is read twice (maximum, then split); loads are issued in independent batches of 8 per thread: a
        // loop that consumes each load right away pays the full memory latency per element (it was 20 us per launch)
        const int per_tap = p.Cin * p.Cout;
        const int total = p.ntaps * per_tap;
        const long wplane = (long)g.nb * 32 * g.ksb;
        float m = 0.f;
        for (int base = 0; base < total; base += 8 * T_THREADS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * T_THREADS + tid;
                const int ee = e < total ? e : 0;
                const int t = fdiv(ee, per_tap, g.m_pertap), r = ee - t * per_tap;
                const int ci = fdiv(r, p.Cout, g.m_cout), co = r - ci * p.Cout;
                v[u] = p.w[p.taps[t].woff + (long)ci * p.ldb + co];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, (base + u * T_THREADS + tid < total) ? fabsf(v[u]) : 0.f);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        ew = ss_amax_exp(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        const float sw = ldexpf(1.f, 14 - ew);
        for (int base = 0; base < total; base += 8 * T_THREADS) {
            float v[8];
            int dsto[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + u * T_THREADS + tid;
                const int ee = e < total ? e : 0;
                const int t = fdiv(ee, per_tap, g.m_pertap), r = ee - t * per_tap;
                const int ci = fdiv(r, p.Cout, g.m_cout), co = r - ci * p.Cout;
                v[u] = p.w[p.taps[t].woff + (long)ci * p.ldb + co];
                const int k = ((t * g.cgp + (ci >> 4)) * 2 + ((ci >> 3) & 1)) * 8 + (ci & 7);
                dsto[u] = e < total ? co * g.ksb + k * 2 : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (dsto[u] < 0) continue;
                const float x = v[u] * sw;
                const _Float16 h = (_Float16)x;
                const _Float16 l = (_Float16)(x - (float)h);
                *(_Float16*)(sW + dsto[u]) = h;
                *(_Float16*)(sW + dsto[u] + wplane) = l;
            }
        }
        __syncthreads();
    }

    const int tiles_x = (p.OW + TW - 1) / TW, tiles_y = (p.OH + TH - 1) / TH;
    const int ntiles = p.N * tiles_y * tiles_x;
    int first, stride, end;
    tile_walk(ntiles, first, stride, end);
    // a 4-wide window that straddles the end of a pixel's channels reads into the next pixel(s) (valid memory, masked below) -- except
    // where it would run past the LAST element of the view (the last pixel; with fewer than 4 channels per pixel the last few
    // pixels: a batch-1 input image ending on a page boundary faulted here), where the elements are fetched one by one
    const long in_end = ((long)p.N * p.IH * p.IW - 1) * p.in_cs + p.Cin;
    const int hp = g.hh * g.hw;                              // halo pixels
    const int c4n = g.cg * 2;                                // float4 slots per pixel (padded channels)
    const int nslots = hp * c4n;

    // tile-invariant decomposition of this thread's staging slots: slot i <-> (halo row, halo column, float4 index)
    // packed: halo row | halo column << 8 | float4 index << 16 ; -1 = no slot (one register per slot)
    int s_pk[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int e = tid + i * T_THREADS;
        const int hpix = fdiv(e, c4n, g.m_c4n), c4 = e - hpix * c4n;
        const int hy = fdiv(hpix, g.hw, g.m_hw);
        s_pk[i] = e < nslots ? (hy | ((hpix - hy * g.hw) << 8) | (c4 << 16)) : -1;
    }
    // Prefetch registers.  issue_loads is BRANCH-FREE: every lane always loads four elements (from the tensor's first element when its
    // slot is padding / outside the image), and nothing looks at the data until the tile is consumed one iteration later -- a load
    // whose result is tested right away costs its full latency per slot (s_waitcnt vmcnt(0) after each one in the first version).
    f32x4 pf[PF];
    unsigned ok_mask = 0, edge_mask = 0;      // per tile: slot holds real data / slot is the tensor's last, partial 4-wide window
    auto issue_loads = [&](int tile) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        ok_mask = 0; edge_mask = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int pk = s_pk[i] < 0 ? 0 : s_pk[i];
            const int iy = ss_map_index(oy0 + g.hy0 + (pk & 255), p.IH, p.reflect);
            const int ix = ss_map_index(ox0 + g.hx0 + ((pk >> 8) & 255), p.IW, p.reflect);
            const int c = (pk >> 16) * 4;
            // (iy / ix >= size: far overhang of an edge tile under reflection: feeds no stored output)
            const bool ok = s_pk[i] >= 0 && iy >= 0 && ix >= 0 && iy < p.IH && ix < p.IW && c < p.Cin && !(g.dbg & 1);
            const long off = ((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs;
            const bool edge = ok && off + c + 4 > in_end;
            ok_mask |= (ok && !edge) ? (1u << i) : 0u;
            edge_mask |= edge ? (1u << i) : 0u;
            pf[i] = ld4(gin + ((ok && !edge) ? off + c : 0L));
        }
    };
    // slot i of the CURRENT tile (n0, y0, x0 = its image / origin): the four channels with padding, out-of-image and the partial
    // last window resolved
    auto slot_value = [&](int i, int n0, int y0, int x0) {
        f32x4 v = pf[i];
        const int c = (s_pk[i] >> 16) * 4;
        if (!((ok_mask >> i) & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c + 1 >= p.Cin) v[1] = 0.f;
        if (c + 2 >= p.Cin) v[2] = 0.f;
        if (c + 3 >= p.Cin) v[3] = 0.f;
        if ((edge_mask >> i) & 1) {           // one pixel of the whole tensor: element-wise
            const int iy = ss_map_index(y0 + g.hy0 + (s_pk[i] & 255), p.IH, p.reflect);
            const int ix = ss_map_index(x0 + g.hx0 + ((s_pk[i] >> 8) & 255), p.IW, p.reflect);
            const TI* src = gin + ((long)(n0 * p.IH + iy) * p.IW + ix) * p.in_cs + c;
            v[0] = (float)src[0];
            if (c + 1 < p.Cin) v[1] = (float)src[1];
            if (c + 2 < p.Cin) v[2] = (float)src[2];
        }
        return v;
    };

    // The phases of a tile use different units (LDS, matrix pipe, memory) but are serial inside a workgroup; the co-resident
    // workgroups of a CU start together and would stay in lockstep (measured: phase times ADD).  Workgroups 256 apart share a CU
    // (round-robin dispatch): stagger their start by a fraction of a tile time.
    if (!(g.dbg & 16)) {
        const int slot = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < slot * g.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    int tile = first;
    if (tile < end) issue_loads(tile);
    const unsigned char* abase = sW + (long)l31 * g.ksb + lh * 16;                       // weight fragment: row = output channel
    const unsigned char* bbase = sIn + (long)(wave * g.hw + l31) * g.psb;               // input fragment: column = pixel of this wave's row
    const long wplane = (long)g.nb * 32 * g.ksb;

    for (; tile < end; tile += stride) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- registers -> LDS as fp32 (in the final 32-byte units), max |x| of the tile ----
        float vmax = 0.f;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (i < g.pf && s_pk[i] >= 0) {
                const f32x4 v = slot_value(i, n, oy0, ox0);
                vmax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), vmax);
                *(f32x4*)(sIn + ((s_pk[i] & 255) * g.hw + ((s_pk[i] >> 8) & 255)) * g.psb + (s_pk[i] >> 16) * 16) = v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        if (lane == 0) red[wave] = vmax;
        lds_barrier();
        vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int ex = ss_amax_exp(vmax);
        const float sx = ldexpf(1.f, 14 - ex);
        // ---- in place: every 32-byte unit (8 channels, fp32) -> 16 B of h + 16 B of l ----
        for (int e = tid; e < hp * g.cg && !(g.dbg & 2); e += T_THREADS) {
            const int hpix = fdiv(e, g.cg, g.m_cg), c = e - hpix * g.cg;
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
            if (NP > 1) *(f16x8*)(u + 16) = l;
        }
        lds_barrier();
        // the next tile's global loads fly during the contraction and the stores of this one
        if (tile + stride < end) issue_loads(tile + stride);

        // ---- contraction: rows = output channels (weights), columns = the 32 pixels of this wave's tile row ----
        f32x16 acc[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        // K steps = (tap, 16 channels): lanes 0..31 take the step's first channel group, lanes 32..63 its second (clamped to the
        // last group when cg is odd: its weight rows are zero).  Fragments of step s+1 are read before the MFMAs of step s issue.
        const int nsteps = (g.dbg & 4) ? 0 : p.ntaps * g.cgp;
        int t = 0, c2 = 0;
        auto frag_addr = [&](int tt, int cc) {
            const int tapoff = ((p.in_oy + p.taps[tt].dy - g.hy0) * g.hw + (p.in_ox + p.taps[tt].dx - g.hx0)) * g.psb;
            int cgi = 2 * cc + lh;
            cgi = cgi < g.cg ? cgi : g.cg - 1;
            return bbase + tapoff + cgi * 32;
        };
        f16x8 xh, xl, wh[NBT], wl[NBT];
        auto read_frags = [&](int step, int tt, int cc) {
            const unsigned char* bp = frag_addr(tt, cc);
            xh = *(const f16x8*)bp;
            if (NP > 1) xl = *(const f16x8*)(bp + 16);
            const unsigned char* ap = abase + step * 32;
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb)
                if (nb < g.nb) {
                    wh[nb] = *(const f16x8*)(ap + (long)nb * 32 * g.ksb);
                    if (NP > 1) wl[nb] = *(const f16x8*)(ap + (long)nb * 32 * g.ksb + wplane);
                }
        };
        if (nsteps > 0) read_frags(0, 0, 0);
        for (int step = 0; step < nsteps; ++step) {
            const f16x8 cxh = xh, cxl = xl;
            f16x8 cwh[NBT], cwl[NBT];
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb) { cwh[nb] = wh[nb]; cwl[nb] = wl[nb]; }
            if (++c2 == g.cgp) { c2 = 0; ++t; }
            if (step + 1 < nsteps) read_frags(step + 1, t, c2);
#pragma unroll
            for (int nb = 0; nb < NBT; ++nb)
                if (nb < g.nb) {
                    if (NP > 1) {
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwl[nb], cxh, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwh[nb], cxl, acc[nb], 0, 0, 0);
                    }
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cwh[nb], cxh, acc[nb], 0, 0, 0);
                }
        }

        // ---- epilogue: lane = pixel, registers = output channels in groups of 4 consecutive ones: 16-byte stores ----
        const float oscale = ldexpf(1.f, (ex - 14) + (ew - 14));
        const int oy = oy0 + wave, ox = ox0 + l31;
        if (oy < p.OH && ox < p.OW && !(g.dbg & 8)) {
            TO* opix = gout + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs;
            // PLAIN (no activation, no accumulation: every BatchNorm-followed layer) is a compile-time path: the generic activation
            // switch costs ~14 scalar instructions and taken branches PER ELEMENT, 16 NB elements per lane -- a third of the
            // tile's instruction count
            auto stores = [&](auto plain_c) {
                constexpr bool PLAIN = decltype(plain_c)::value;
#pragma unroll
                for (int nb = 0; nb < NBT; ++nb) {
                    if (nb >= g.nb) continue;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int co = nb * 32 + 8 * q4 + 4 * lh;
                        if (co >= p.Cout) continue;
                        f32x4 v = {acc[nb][4 * q4] * oscale, acc[nb][4 * q4 + 1] * oscale, acc[nb][4 * q4 + 2] * oscale, acc[nb][4 * q4 + 3] * oscale};
                        const int nv = p.Cout - co < 4 ? p.Cout - co : 4;
                        const f32x4 b4 = *(const f32x4*)(sBias + co);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = PLAIN ? v[k] + b4[k] : ss_apply_act(v[k] + b4[k], p.act, p.alpha);
                        if (nv == 4) {
                            if (!PLAIN && p.accumulate) v += ld4(opix + co);
                            st4(opix + co, v);
                        } else {
                            for (int k = 0; k < nv; ++k) {
                                float o = v[k];
                                if (!PLAIN && p.accumulate) o += (float)opix[co + k];
                                opix[co + k] = (TO)o;
                            }
                        }
                    }
                }
            };
            if (p.act == SS_ACT_NONE && !p.accumulate) stores(std::true_type{}); else stores(std::false_type{});
        }
        lds_barrier();          // the next tile overwrites sIn
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
    // (the x3h kernel's rows are tap * CaP + ca with CaP = Ca rounded up to 4: whole 4-channel units per transposing read)
    const int M = p.ntaps * (ss_tuning().twgrad_x3h && ss_x3h_enabled() ? ((p.Ca + 3) & ~3) : p.Ca);
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
        const int per = (int)((160 * 1024) / (g->smem + 1024 + 64));
        g->wgs_per_cu = per > 2 ? 2 : per;
        if (g->wgs_per_cu >= 2) return true;
    }
    return g->wgs_per_cu >= 1;
}

template <typename TI, int MBW, int NB>      // activation storage type; 32-row blocks per wave, 32-wide column blocks
__global__ __launch_bounds__(T_THREADS, 2) void twgrad_kernel(WGradParams p, WTileGeom g) {
    const TI* const ga = (const TI*)p.a;
    const TI* const gb = (const TI*)p.b;
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
    const bool va = (p.Ca % 4 == 0);         // whole 4-channel groups: one element-aligned wide access (see ld4)
    const bool vb = (p.Cb % 4 == 0);
    const int a4 = g.psa / 4, b4 = g.psb / 4;

    for (int tile = first; tile < end; tile += stride) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int gy0 = ty * g.th, gx0 = tx * TW;
        // stage: a wave per halo row of a / per tile row of b
        for (int hy = wave; hy < g.hh; hy += 4) {
            int iy = ss_map_index(gy0 + g.hy0 + hy, p.AH, p.reflect);
            if (iy >= p.AH) iy = -1;
            const TI* rowp = ga + (long)(n * p.AH + (iy < 0 ? 0 : iy)) * p.AW * p.a_cs;
            float* drow = sA + (long)hy * g.hw * g.psa;
            for (int e = lane; e < g.hw * a4; e += 64) {
                const int hx = a4 == 1 ? e : (int)__umulhi((unsigned)e, g.m_a4), c = (e - hx * a4) * 4;
                int ix = ss_map_index(gx0 + g.hx0 + hx, p.AW, p.reflect);
                if (ix >= p.AW) ix = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && ix >= 0 && c < p.Ca) {
                    const TI* src = rowp + (long)ix * p.a_cs + c;
                    if (va) v = ld4(src);
                    else {
                        v[0] = (float)src[0];
                        if (c + 1 < p.Ca) v[1] = (float)src[1];
                        if (c + 2 < p.Ca) v[2] = (float)src[2];
                        if (c + 3 < p.Ca) v[3] = (float)src[3];
                    }
                }
                *(f32x4*)(drow + (long)hx * g.psa + c) = v;
            }
        }
        for (int py = wave; py < g.th; py += 4) {
            const int gy = gy0 + py;
            const TI* rowp = gb + (long)(n * p.GH + (gy < p.GH ? gy : 0)) * p.GW * p.b_cs;
            float* drow = sB + (long)py * TW * g.psb;
            for (int e = lane; e < TW * b4; e += 64) {
                const int px = b4 == 1 ? e : (int)__umulhi((unsigned)e, g.m_b4), c = (e - px * b4) * 4;
                const int gx = gx0 + px;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy < p.GH && gx < p.GW && c < p.Cb) {          // pixels outside the grid contribute zero (b = 0)
                    const TI* src = rowp + (long)gx * p.b_cs + c;
                    if (vb) v = ld4(src);
                    else {
                        v[0] = (float)src[0];
                        if (c + 1 < p.Cb) v[1] = (float)src[1];
                        if (c + 2 < p.Cb) v[2] = (float)src[2];
                        if (c + 3 < p.Cb) v[3] = (float)src[3];
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the fp16 matrix cores ("x3h": 5.3x the fp32 matrix rate; the fp32 kernel above spends most of its time
// in v_mfma_f32_32x32x2_f32: 2 pixels per instruction).  K = the pixels of a tile: both operands are K-major in LDS ([pixel][channel]),
// so the MFMA fragments come from the TRANSPOSING LDS read ds_read_b64_tr_b16 (lane j of a 16-lane group addresses 4 consecutive
// channels of pixel j / 4; lane i receives the 4 pixels of channel i -- gemm_tn_x3h.hip, tools/tr_probe.hip).
//  * staging as above in fp32, the tile maxima of a (halo tile) and b taken on the way; then IN PLACE every 16-byte unit (4 channels,
//    fp32) becomes 8 B of h + 8 B of l:  a * sA = h + l  with sA = 2^(14 - eA) the tile's own power-of-two scale (values within
//    2^-17 of the TILE maximum keep 22 significand bits, as in tconv_kernel), products hh + hl + lh into ONE accumulator set.
//  * one accumulator set lives across all tiles of the workgroup, in the unit 2^(E - 28) where E = the largest eA + eB seen so far:
//    a tile with a larger exponent rescales the accumulators by 2^(E_old - E_new) (a power of two: exact) and becomes the new unit; a
//    smaller tile shifts ITS b pieces down by 2^(e - E) while they are split, so that its products land in the running unit (what it
//    loses there is below 2^-38 of the largest tile's products: noise against the sum it is added to).  All-zero tiles are skipped.
//  * rows are the PADDED index m' = tap * CaP + ca with CaP = Ca rounded up to 4 (a 16-lane group reads whole 4-channel units);
//    padding rows / columns compute on zero-filled staging and are not stored.
template <typename TI, int MBW, int NB>
__global__ __launch_bounds__(T_THREADS, 2) void twgrad_x3h_kernel(WGradParams p, WTileGeom g) {
    const TI* const ga = (const TI*)p.a;
    const TI* const gb = (const TI*)p.b;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                // [hh*hw][psa * 4 bytes]
    unsigned char* sB = smem + g.a_bytes;                    // [th*32][psb * 4 bytes]
    float* red = (float*)(smem + g.a_bytes + g.b_bytes);     // [8]: wave maxima of a, of b
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int CaP = (p.Ca + 3) & ~3, CbP = (p.Cb + 3) & ~3;
    const int MP = p.ntaps * CaP;                            // padded row count
    const int wm = wave % g.mw, wk = wave / g.mw;
    const int pa_b = g.psa * 4, pb_b = g.psb * 4;            // bytes per staged pixel

    // fragment addressing (bytes): this lane addresses unit `quad` of k row kq of its 16-lane group; group = (m half mh, k half lh)
    const int kq = (lane & 15) >> 2, quad = lane & 3, mh = (lane >> 4) & 1;
    int a_base[MBW];
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int blk = wm + g.mw * i;
        int mq = blk * 32 + 16 * mh + 4 * quad;
        if (mq >= MP) mq = 0;                                // rows past the end: any valid address (their results are not stored)
        const int t = mq / CaP, ca = mq - t * CaP;
        a_base[i] = ((p.a_oy + p.taps[t].dy - g.hy0) * g.hw + (p.a_ox + p.taps[t].dx - g.hx0) + 8 * lh + kq) * pa_b + (ca >> 2) * 16;
    }
    int b_base[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        int cq = j * 32 + 16 * mh + 4 * quad;
        if (cq >= CbP) cq = 0;
        b_base[j] = (8 * lh + kq) * pb_b + (cq >> 2) * 16;
    }

    f32x16 acc[MBW][NB];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int E = -(1 << 20);                                       // exponent of the running unit: none yet

    const int tiles_x = (p.GW + TW - 1) / TW, tiles_y = (p.GH + g.th - 1) / g.th;
    const int ntiles = p.N * tiles_y * tiles_x;
    int first, stride, end;
    tile_walk(ntiles, first, stride, end);
    const bool va = (p.Ca % 4 == 0), vb = (p.Cb % 4 == 0);
    const int a4 = g.psa / 4, b4 = g.psb / 4;
    const int na4 = g.hh * g.hw * a4, nb4 = g.th * TW * b4;

    for (int tile = first; tile < end; tile += stride) {
        const int tx = tile % tiles_x, r1 = tile / tiles_x, ty = r1 % tiles_y, n = r1 / tiles_y;
        const int gy0 = ty * g.th, gx0 = tx * TW;
        float ma = 0.f, mbv = 0.f;
        for (int hy = wave; hy < g.hh; hy += 4) {
            int iy = ss_map_index(gy0 + g.hy0 + hy, p.AH, p.reflect);
            if (iy >= p.AH) iy = -1;
            const TI* rowp = ga + (long)(n * p.AH + (iy < 0 ? 0 : iy)) * p.AW * p.a_cs;
            unsigned char* drow = sA + (long)hy * g.hw * pa_b;
            for (int e = lane; e < g.hw * a4; e += 64) {
                const int hx = a4 == 1 ? e : (int)__umulhi((unsigned)e, g.m_a4), c = (e - hx * a4) * 4;
                int ix = ss_map_index(gx0 + g.hx0 + hx, p.AW, p.reflect);
                if (ix >= p.AW) ix = -1;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (iy >= 0 && ix >= 0 && c < p.Ca) {
                    const TI* src = rowp + (long)ix * p.a_cs + c;
                    if (va) v = ld4(src);
                    else {
                        v[0] = (float)src[0];
                        if (c + 1 < p.Ca) v[1] = (float)src[1];
                        if (c + 2 < p.Ca) v[2] = (float)src[2];
                        if (c + 3 < p.Ca) v[3] = (float)src[3];
                    }
                }
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *(f32x4*)(drow + (long)hx * pa_b + c * 4) = v;
            }
        }
        for (int py = wave; py < g.th; py += 4) {
            const int gy = gy0 + py;
            const TI* rowp = gb + (long)(n * p.GH + (gy < p.GH ? gy : 0)) * p.GW * p.b_cs;
            unsigned char* drow = sB + (long)py * TW * pb_b;
            for (int e = lane; e < TW * b4; e += 64) {
                const int px = b4 == 1 ? e : (int)__umulhi((unsigned)e, g.m_b4), c = (e - px * b4) * 4;
                const int gx = gx0 + px;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy < p.GH && gx < p.GW && c < p.Cb) {
                    const TI* src = rowp + (long)gx * p.b_cs + c;
                    if (vb) v = ld4(src);
                    else {
                        v[0] = (float)src[0];
                        if (c + 1 < p.Cb) v[1] = (float)src[1];
                        if (c + 2 < p.Cb) v[2] = (float)src[2];
                        if (c + 3 < p.Cb) v[3] = (float)src[3];
                    }
                }
                mbv = fmaxf(mbv, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                *(f32x4*)(drow + (long)px * pb_b + c * 4) = v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, off)); mbv = fmaxf(mbv, __shfl_xor(mbv, off)); }
        if (lane == 0) { red[wave] = ma; red[4 + wave] = mbv; }
        __syncthreads();
        ma = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        mbv = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        const bool live = ma > 0.f && mbv > 0.f;             // uniform over the workgroup
        if (live) {
            const int ea = ss_amax_exp(ma), eb = ss_amax_exp(mbv);
            const int et = ea + eb;
            int shift = 0;                                   // exponent taken off this tile's b pieces (<= 0)
            if (et > E) {
                if (E > -(1 << 19)) {
                    const float f = ldexpf(1.f, (E - et) < -126 ? -126 : (E - et));
#pragma unroll
                    for (int i = 0; i < MBW; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
                }
                E = et;
            } else {
                shift = et - E;
                if (shift < -60) shift = -60;
            }
            const float sa = ldexpf(1.f, 14 - ea), sb = ldexpf(1.f, 14 - eb + shift);
            // in place: 16-byte fp32 units -> 8 B of h + 8 B of l
            for (int e = tid; e < na4; e += T_THREADS) {
                unsigned char* u = sA + (long)e * 16;
                const f32x4 v = *(const f32x4*)u;
                f16x4 h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float x = v[k] * sa; h[k] = (_Float16)x; l[k] = (_Float16)(x - (float)h[k]); }
                *(f16x4*)u = h;
                *(f16x4*)(u + 8) = l;
            }
            for (int e = tid; e < nb4; e += T_THREADS) {
                unsigned char* u = sB + (long)e * 16;
                const f32x4 v = *(const f32x4*)u;
                f16x4 h, l;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float x = v[k] * sb; h[k] = (_Float16)x; l[k] = (_Float16)(x - (float)h[k]); }
                *(f16x4*)u = h;
                *(f16x4*)(u + 8) = l;
            }
        }
        __syncthreads();
        if (live) {
            // K steps of 16 pixels: (tile row py, half ks of the 32-pixel row); this wave's share of the rows: py = wk, wk + kw, ...
            for (int py = wk; py < g.th; py += g.kw) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const unsigned char* ap = sA + (long)(py * g.hw + 16 * ks) * pa_b;
                    const unsigned char* bp = sB + (long)(py * TW + 16 * ks) * pb_b;
                    f16x8 bh[NB], bl[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const unsigned char* q = bp + b_base[j];
                        const s16x4 h0 = tr4(q), h1 = tr4(q + 4 * pb_b), l0 = tr4(q + 8), l1 = tr4(q + 4 * pb_b + 8);
                        bh[j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        bl[j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int i = 0; i < MBW; ++i) {
                        const unsigned char* q = ap + a_base[i];
                        const s16x4 h0 = tr4(q), h1 = tr4(q + 4 * pa_b), l0 = tr4(q + 8), l1 = tr4(q + 4 * pa_b + 8);
                        const f16x8 ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        const f16x8 al = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // back to real units: the accumulators hold sum a * b * 2^(28 - E)
    {
        const float f = E > -(1 << 19) ? ldexpf(1.f, E - 28) : 0.f;
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
    }
    // join the kw pixel-row groups (fixed order), then one partial per workgroup (as twgrad_kernel)
    if (g.kw > 1) {
        float* redj = (float*)smem;
        constexpr int PER = MBW * NB * 16;
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) redj[((long)wave * PER + (i * NB + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r];
                        for (int k = 1; k < g.kw; ++k) v += redj[((long)(wm + k * g.mw) * PER + (i * NB + j) * 16 + r) * 64 + lane];
                        acc[i][j][r] = v;
                    }
        }
    }
    if (wk != 0) return;
    const int M = p.ntaps * p.Ca;
    float* part = p.part + (long)blockIdx.x * M * p.Cb;
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int blk = wm + g.mw * i;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int cb = j * 32 + l31;
            if (cb >= p.Cb) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mp = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;          // padded row -> (tap, ca)
                if (mp >= MP) continue;
                const int t = mp / CaP, ca = mp - t * CaP;
                if (ca < p.Ca) part[(long)(t * p.Ca + ca) * p.Cb + cb] = acc[i][j][r];
            }
        }
    }
}

// x3h (fp16 matrix cores) unless switched off or the fp32-MFMA-only mode is on; its row blocks count PADDED rows (see the kernel)
bool twgrad_use_x3h(const WGradParams& p) {
    (void)p;          // (the tile kernels are only taken in the AUTO / X6 modes: conv_api.hip twgrad_takes)
    return ss_tuning().twgrad_x3h && ss_x3h_enabled();
}

template <typename TI, int MBW, int NB>
int launch_twgrad_t(const WGradParams& p, const WTileGeom& g, int nwg, hipStream_t s) {
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)twgrad_kernel<TI, MBW, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)twgrad_x3h_kernel<TI, MBW, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    const bool x3h = twgrad_use_x3h(p);
    char name[64];
    snprintf(name, sizeof(name), x3h ? "twgrad_x3h_kernel<%d,%d>" : "twgrad_kernel<%d,%d>", MBW, NB);
    const double pix = (double)p.N * p.GH * p.GW;
    SsProfScope prof(name, 2.0 * p.ntaps * p.Ca * p.Cb * pix * (x3h ? 3 : 1), (double)sizeof(TI) * pix * (p.Ca + p.Cb), s);
    if (x3h) hipLaunchKernelGGL((twgrad_x3h_kernel<TI, MBW, NB>), dim3(nwg), dim3(T_THREADS), g.smem + 64, s, p, g);
    else hipLaunchKernelGGL((twgrad_kernel<TI, MBW, NB>), dim3(nwg), dim3(T_THREADS), g.smem, s, p, g);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <int MBW, int NB>
int launch_twgrad(const WGradParams& p, const WTileGeom& g, int nwg, hipStream_t s) {
    switch (p.dtype) {
        case SS_DTYPE_F32: return launch_twgrad_t<float, MBW, NB>(p, g, nwg, s);
        case SS_DTYPE_F16: return launch_twgrad_t<_Float16, MBW, NB>(p, g, nwg, s);
        case SS_DTYPE_BF16: return launch_twgrad_t<__bf16, MBW, NB>(p, g, nwg, s);
    }
    return SS_ERR_INVALID;
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

size_t ss_tconv_ws(const GConvParams&) { return 0; }       // the weight planes are formed in LDS by every workgroup

namespace {
template <typename TI, typename TO, int NBT, int PF, int NP>
int launch_tconv(const GConvParams& p, const TileGeom& g, int nwg, hipStream_t s) {
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)tconv_kernel<TI, TO, NBT, PF, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)attr_set;
    char name[64];
    snprintf(name, sizeof(name), "tconv_kernel<%d,%d,%d> %s", NBT, PF, NP, NP == 3 ? "x3h" : "f16");
    const double pix = (double)p.N * p.OH * p.OW;
    SsProfScope prof(name, 2.0 * pix * p.Cout * p.ntaps * p.Cin * NP, (double)sizeof(TI) * pix * p.Cin + (double)sizeof(TO) * pix * p.Cout * (p.accumulate ? 2 : 1), s);
    hipLaunchKernelGGL((tconv_kernel<TI, TO, NBT, PF, NP>), dim3(nwg), dim3(T_THREADS), g.smem, s, p, g);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
template <typename TI, typename TO, int NP>
int dispatch_tconv(const GConvParams& p, const TileGeom& g, int nwg, hipStream_t s) {
    const int nbt = g.nb <= 1 ? 1 : (g.nb == 2 ? 2 : 4);
    const int pfc = g.pf <= 4 ? 4 : (g.pf <= 8 ? 8 : 13);
#define TC(NBT, PF) if (nbt == NBT && pfc == PF) return launch_tconv<TI, TO, NBT, PF, NP>(p, g, nwg, s);
    TC(1, 4) TC(1, 8) TC(1, 13) TC(2, 4) TC(2, 8) TC(2, 13) TC(4, 4) TC(4, 8) TC(4, 13)
#undef TC
    return SS_ERR_UNSUPPORTED;
}
}  // namespace

int ss_launch_tconv(const GConvParams& p, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)ws; (void)ws_bytes;
    TileGeom g;
    if (!tile_geom(p, &g)) return SS_ERR_UNSUPPORTED;
    const int tiles = p.N * ((p.OH + TH - 1) / TH) * ((p.OW + TW - 1) / TW);
    const int nwg = tile_nwg(tiles, g.wgs_per_cu);
    switch (p.dtype) {
        case SS_DTYPE_F32: return dispatch_tconv<float, float, 3>(p, g, nwg, s);
        // 16-bit stored activations: exactly representable as ONE fp16 piece under the tile scale (bf16's 8 significand bits fit
        // fp16's 11), multiplied with the leading fp16 piece of the fp32 master weights: plain mixed precision, one product
        case SS_DTYPE_F16: return dispatch_tconv<_Float16, _Float16, 1>(p, g, nwg, s);
        case SS_DTYPE_BF16: return dispatch_tconv<__bf16, __bf16, 1>(p, g, nwg, s);
    }
    return SS_ERR_INVALID;
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
