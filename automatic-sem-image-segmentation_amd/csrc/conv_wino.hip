// Winograd F(RxR, 3x3), R = 2 or 4, for the stride-1 3x3 convolutions with wide channels (the generator trunk: 85 % of
// the CycleGAN FLOPs), all in fp32:  2.25x (R=2) / 4x (R=4) fewer multiply-adds than the direct implicit GEMM.
//   V = B^T d B   ((R+2)^2 input patch per RxR output tile; reflect / zero padding applied in the gather)
//   U = G g G^T   (weights, per call)
//   M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co]      (R+2)^2 independent GEMMs -> ONE batched gconv_mfma launch
//   Y = A^T M A
// weight gradient (F(3x3, RxR)):  S_xi[ci][co] = sum_tiles V_xi[tile][ci] * E_xi[tile][co],  E = A e A^T,  dW = G^T S G.
// The transforms are streaming (HBM-bound) kernels: one thread = one tile x VW channels (float4 for R=2, float2 for R=4).
// fp32 error vs fp64 (rel-L2, K = 128 channels, measured on CPU): direct 2.2e-7, F(2,3) 3.5e-7, F(4,3) 2.3e-6.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int R> struct WT;
template <> struct WT<2> { typedef f32x4 T; static constexpr int VW = 4; };
template <> struct WT<4> { typedef f32x2 T; static constexpr int VW = 2; };

template <class T> __device__ __forceinline__ T zero_v() { T z; for (int k = 0; k < (int)(sizeof(T) / 4); ++k) z[k] = 0.f; return z; }
template <class T> __device__ __forceinline__ T ldz(const float* base, long off, bool ok) {
    const T v = *(const T*)(base + (ok ? off : 0));
    return ok ? v : zero_v<T>();
}

// VW consecutive channels of an activation view in its storage type <-> fp32 vector (16-bit storage: BASELINE configs 2 / 5)
template <class T, class TI> __device__ __forceinline__ T ldz_t(const TI* base, long off, bool ok) {
    constexpr int N = (int)(sizeof(T) / 4);
    T v = zero_v<T>();
    if (sizeof(TI) == 4) {
        v = *(const T*)((const float*)base + (ok ? off : 0));
    } else {
        typedef TI TV __attribute__((ext_vector_type(N)));
        const TV q = *(const TV*)(base + (ok ? off : 0));
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = (float)q[k];
    }
    return ok ? v : zero_v<T>();
}
template <class T, class TO> __device__ __forceinline__ T ld_t(const TO* p) { return ldz_t<T, TO>(p, 0, true); }
template <class T, class TO> __device__ __forceinline__ void st_t(TO* p, const T& v) {
    constexpr int N = (int)(sizeof(T) / 4);
    if (sizeof(TO) == 4) {
        *(T*)p = v;
    } else {
        typedef TO TV __attribute__((ext_vector_type(N)));
        TV q;
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = (TO)v[k];
        *(TV*)p = q;
    }
}

// fp32 -> (hi, lo) bf16 planes, round-to-nearest-even (finite inputs)
__device__ __forceinline__ unsigned short bf16_rne(float v) {
    unsigned int u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void split_bf16(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf16_rne(v);
    lo = bf16_rne(v - __uint_as_float((unsigned int)hi << 16));
}
template <class T> __device__ __forceinline__ void store_split(unsigned short* ph, unsigned short* pl, const T& v) {
    constexpr int N = (int)(sizeof(T) / 4);
    unsigned short h[N], l[N];
#pragma unroll
    for (int k = 0; k < N; ++k) split_bf16(v[k], h[k], l[k]);
    if (N == 2) {
        *(unsigned int*)ph = (unsigned int)h[0] | ((unsigned int)h[1] << 16);
        *(unsigned int*)pl = (unsigned int)l[0] | ((unsigned int)l[1] << 16);
    } else {
#pragma unroll
        for (int k = 0; k < N; k += 2) {
            ((unsigned int*)ph)[k / 2] = (unsigned int)h[k] | ((unsigned int)h[k + 1] << 16);
            ((unsigned int*)pl)[k / 2] = (unsigned int)l[k] | ((unsigned int)l[k + 1] << 16);
        }
    }
}

// 1-D transforms (applied to rows, then columns)
template <int R, class T> __device__ __forceinline__ void t_in(const T* d, T* v) {      // B^T d
    if (R == 2) {
        v[0] = d[0] - d[2]; v[1] = d[1] + d[2]; v[2] = d[2] - d[1]; v[3] = d[1] - d[3];
    } else {
        // explicit fused multiply-adds: the compiler's own contraction of "4 d0 - 5 d2 + d4" may differ between two instantiations of
        // the calling kernel (it did between the plain and the fused-normalisation input transform), and those must agree bit for bit
        constexpr int N = (int)(sizeof(T) / 4);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            v[0][k] = __builtin_fmaf(4.f, d[0][k], __builtin_fmaf(-5.f, d[2][k], d[4][k]));
            v[1][k] = __builtin_fmaf(-4.f, d[1][k] + d[2][k], d[3][k] + d[4][k]);
            v[2][k] = __builtin_fmaf(4.f, d[1][k] - d[2][k], d[4][k] - d[3][k]);
            v[3][k] = __builtin_fmaf(2.f, d[3][k] - d[1][k], d[4][k] - d[2][k]);
            v[4][k] = __builtin_fmaf(2.f, d[1][k] - d[3][k], d[4][k] - d[2][k]);
            v[5][k] = __builtin_fmaf(4.f, d[1][k], __builtin_fmaf(-5.f, d[3][k], d[5][k]));
        }
    }
}
template <int R> __device__ __forceinline__ void t_w(const float* g, float* u) {        // G g
    if (R == 2) {
        u[0] = g[0]; u[1] = 0.5f * (g[0] + g[1] + g[2]); u[2] = 0.5f * (g[0] - g[1] + g[2]); u[3] = g[2];
    } else {
        u[0] = 0.25f * g[0];
        u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
        u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
        u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
        u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
        u[5] = g[2];
    }
}
template <int R, class T> __device__ __forceinline__ void t_out(const T* m, T* y) {     // A^T m
    if (R == 2) {
        y[0] = m[0] + m[1] + m[2]; y[1] = m[1] - m[2] - m[3];
    } else {
        y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
        y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
        y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
        y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
    }
}
template <int R, class T> __device__ __forceinline__ void t_dy(const T* e, T* a) {      // A e
    if (R == 2) {
        a[0] = e[0]; a[1] = e[0] + e[1]; a[2] = e[0] - e[1]; a[3] = -e[1];
    } else {
        a[0] = e[0];
        a[1] = e[0] + e[1] + e[2] + e[3];
        a[2] = e[0] - e[1] + e[2] - e[3];
        a[3] = e[0] + 2.f * e[1] + 4.f * e[2] + 8.f * e[3];
        a[4] = e[0] - 2.f * e[1] + 4.f * e[2] - 8.f * e[3];
        a[5] = e[3];
    }
}
template <int R> __device__ __forceinline__ void t_dw(const float* s, float* o) {       // G^T s
    if (R == 2) {
        o[0] = s[0] + 0.5f * (s[1] + s[2]); o[1] = 0.5f * (s[1] - s[2]); o[2] = 0.5f * (s[1] + s[2]) + s[3];
    } else {
        o[0] = 0.25f * s[0] - (s[1] + s[2]) * (1.f / 6.f) + (s[3] + s[4]) * (1.f / 24.f);
        o[1] = (s[2] - s[1]) * (1.f / 6.f) + (s[3] - s[4]) * (1.f / 12.f);
        o[2] = -(s[1] + s[2]) * (1.f / 6.f) + (s[3] + s[4]) * (1.f / 6.f) + s[5];
    }
}

// max of a non-negative value over the block -> atomic max of its bit pattern (order-independent: deterministic); all 256 threads
__device__ __forceinline__ void block_amax(float v, unsigned int* out) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __shared__ float bm[4];
    if ((threadIdx.x & 63) == 0) bm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(bm[0], bm[1]), fmaxf(bm[2], bm[3]))));
}

// Workgroups are dealt to the 8 XCDs round-robin (MI355X_MICROARCH.md): block b of the launch takes the b / 8-th block of the
// (b % 8)-th EIGHTH of the tile list, so that the tiles one L2 serves are neighbours (a whole image at batch 8) and the two halo
// rows / columns a 6 x 6 patch shares with the next tile are L2 hits instead of fabric reads.  Leftover blocks (grid % 8) keep their index.
__device__ __forceinline__ unsigned ss_xcd_block(unsigned b, unsigned nb) {
    const unsigned per = nb >> 3;
    return b < (per << 3) ? (b & 7u) * per + (b >> 3) : b;
}

// V[xi][tile][c], tile = (n, ty, tx); patch d[i][j] = in[n, map(R*ty + i - pt), map(R*tx + j - pl), c]
// BF = 0: fp32 V;  1: two bf16 planes (SS_PRECISION=bf16x3);  2: the three bf16 planes of the x6 arithmetic,
// [plane][xi][tile rows padded to Mpad][c] -- the A operand of gemm_x6p.hip, no conversion left for the GEMM
// BF = 5 (16-bit activation storage, TI = _Float16 / __bf16): ONE fp16 plane under the per-tile scale of BF = 3 -- a stored 16-bit
// value has 8 / 11 significand bits, its transform is carried with 11: plain mixed precision, one product in the GEMM
// FN: fused input normalisation (a separate instantiation: the plain one pays nothing for it).  It holds ~165 VGPRs: three workgroups
// per CU (170 registers); under a cap of 128 it spilled 37 of them and moved 1.9 x its algorithmic bytes (132 -> 94 us on the trunk)
template <int R, int BF, bool FN = false, typename TI = float>
__global__ __launch_bounds__(256, FN ? 3 : 1) void wino_input_kernel(const TI* __restrict__ in, int in_cs, int N, int H, int W, int C,
                                                         int TH, int TW, int pt, int pl, int reflect, float* __restrict__ V, long Mpad = 0,
                                                         float* __restrict__ tile_inv = nullptr, unsigned int* __restrict__ amax_out = nullptr,
                                                         const unsigned int* __restrict__ amax_in = nullptr, int amax_stripes = 0, int bound = 0,
                                                         int plain_l = 0, InNorm nm = InNorm()) {
    typedef typename WT<R>::T T;
    constexpr int VW = WT<R>::VW, P = R + 2;
    const int CV = C / VW;
    const long tiles = (long)N * TH * TW;
    const long e = (long)ss_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (e >= tiles * CV) return;
    const int c = (int)(e % CV) * VW;
    const long tile = e / CV;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    int iy[P], ix[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        iy[i] = ss_map_index(R * ty + i - pt, H, reflect);
        ix[i] = ss_map_index(R * tx + i - pl, W, reflect);
    }
    // fused input normalisation (ss_conv_desc::in_norm_*; BF 3 / 4 / 5 launchers only): `in` is the PRE-norm tensor, every element is
    // normalised as it is loaded with norm_apply_kernel's expression (norm.hip), padding zeros stay zero.  16-bit storage: the value is
    // rounded to the storage type exactly where norm_apply_kernel would have stored it -- the two routes agree bit for bit
    constexpr bool fused = FN && (BF == 3 || BF == 4 || BF == 5);
    T n_mu = zero_v<T>(), n_k = zero_v<T>(), n_bt = zero_v<T>();
    if (fused) {
        const long gi = (nm.groups > 1 ? (long)n * C : 0L) + c;
        n_mu = *(const T*)(nm.mean + gi);
        n_k = *(const T*)(nm.rstd + gi);
        n_bt = *(const T*)(nm.beta + c);
        if (nm.gamma) {
            const T gm = *(const T*)(nm.gamma + c);
#pragma unroll
            for (int k = 0; k < VW; ++k) n_k[k] = n_k[k] * gm[k];
        } else {
#pragma unroll
            for (int k = 0; k < VW; ++k) n_k[k] = n_k[k] * 1.f;
        }
    }
    float n_am = 0.f;
    const float n_slope = nm.act == SS_ACT_RELU ? 0.f : (nm.act == SS_ACT_LRELU ? nm.alpha : 1.f);
    T t[P][P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        T d[P], v[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const bool ok = iy[i] >= 0 && ix[j] >= 0;
            d[i] = ldz_t<T, TI>(in, ((long)(n * H + iy[i]) * W + ix[j]) * in_cs + c, ok);
            if (fused) {
                // act in {none, relu, lrelu} (the launcher checks): t > 0 ? t : slope * t with slope 1 / 0 / alpha -- one compare, one
                // multiply, one select per element instead of a switch (this loop runs 36 x VW times per thread)
#pragma unroll
                for (int k = 0; k < VW; ++k) {
                    const float t = __builtin_fmaf(d[i][k] - n_mu[k], n_k[k], n_bt[k]);          // = norm_apply_kernel's expression (norm.hip)
                    const float o = t > 0.f ? t : t * n_slope;
                    d[i][k] = ok ? (float)(TI)o : 0.f;
                    n_am = fmaxf(n_am, fabsf(d[i][k]));
                }
            }
        }
        t_in<R, T>(d, v);
#pragma unroll
        for (int i = 0; i < P; ++i) t[i][j] = v[i];
    }
    const long xs = tiles * C;     // stride between transform positions
    float* o = V + tile * C + c;
    float x3h_scale = 1.f;
    if (BF == 3 || BF == 5) {
        // largest |V| of this tile: thread maximum over its 36 x VW values, then over the C/VW threads of the tile
        // (C/VW in {64, 128, 256}: whole waves; 256-thread blocks hold whole tiles)
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            T v[P];
            t_in<R, T>(t[i], v);
#pragma unroll
            for (int j = 0; j < P; ++j)
#pragma unroll
                for (int k = 0; k < VW; ++k) mx = fmaxf(mx, fabsf(v[j][k]));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        __shared__ float wmax[4];
        const int wave = threadIdx.x >> 6, wpg = CV >> 6;          // waves per tile
        if ((threadIdx.x & 63) == 0) wmax[wave] = mx;
        __syncthreads();
        const int w0 = wave / wpg * wpg;
        mx = wmax[w0];
        for (int k = 1; k < wpg; ++k) mx = fmaxf(mx, wmax[w0 + k]);
        const int ex = ss_amax_exp(mx);                            // mx = f * 2^ex, f in [0.5, 1)
        x3h_scale = ldexpf(1.f, 14 - ex);
        if (c == 0) tile_inv[tile] = ldexpf(1.f, ex - 14);
    }
    if (BF == 4)        // weight gradient: ONE scale for the whole transformed tensor, from max|in| and the transform's gain bound 2^bound
        x3h_scale = ldexpf(1.f, 14 - (ss_amax_exp(__uint_as_float(ss_amax_load(amax_in, amax_stripes))) + bound));
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        T v[P];
        t_in<R, T>(t[i], v);
        if (BF == 5) {
            const long xs3 = Mpad * C;
            unsigned int* o3 = (unsigned int*)((unsigned short*)V + tile * C + c);
#pragma unroll
            for (int j = 0; j < P; ++j) {
#pragma unroll
                for (int k = 0; k < VW; k += 2) {
                    const ss_f2 sv = {v[j][k] * x3h_scale, v[j][k + 1] * x3h_scale};
                    o3[(long)(i * P + j) * xs3 / 2 + k / 2] = __builtin_bit_cast(unsigned int, __builtin_convertvector(sv, ss_h2));
                }
            }
        } else if (BF == 3 || BF == 4) {
            // x3h: v*s = h + 2^-11 l with h, l fp16 and s = 2^e per TILE (all 36 positions, all channels) such that the largest
            // |v*s| of the tile lies in [2^13, 2^14): fp16 keeps 11 bits for everything within 2^-28 of the tile's maximum.
            // 1/s goes to tile_inv[tile]; the output transform multiplies it back (the GEMM is linear in each A row).
            const long xs3 = Mpad * C;
            unsigned int* o3 = (unsigned int*)((unsigned short*)V + tile * C + c);
            const long pl32 = (long)P * P * xs3 / 2;
#pragma unroll
            for (int j = 0; j < P; ++j) {
#pragma unroll
                for (int k = 0; k < VW; k += 2) {
                    unsigned int* d = o3 + (long)(i * P + j) * xs3 / 2 + k / 2;
                    // plain_l (BF == 3, the wide GEMM tile): the low piece at its own magnitude, v*s = h + l
                    if (BF == 3 && plain_l) ss_split_h2(v[j][k] * x3h_scale, v[j][k + 1] * x3h_scale, d[0], d[pl32]);
                    else ss_split_h2s(v[j][k] * x3h_scale, v[j][k + 1] * x3h_scale, d[0], d[pl32]);
                }
            }
        } else if (BF == 2) {
            const long xs3 = Mpad * C;                       // elements between transform positions
            unsigned int* o3 = (unsigned int*)((unsigned short*)V + tile * C + c);
            const long pl32 = (long)P * P * xs3 / 2;         // u32 between planes
#pragma unroll
            for (int j = 0; j < P; ++j) {
#pragma unroll
                for (int k = 0; k < VW; k += 2) {
                    unsigned int h, m, l;
                    ss_split3x2(f32x2{v[j][k], v[j][k + 1]}, h, m, l);
                    unsigned int* d = o3 + (long)(i * P + j) * xs3 / 2 + k / 2;
                    d[0] = h;
                    d[pl32] = m;
                    d[2 * pl32] = l;
                }
            }
        } else if (BF == 1) {       // two bf16 planes [xi][tile][c] (hi plane, then lo plane) in the space of the fp32 V
            unsigned short* oh = (unsigned short*)V + tile * C + c;
            unsigned short* ol = oh + (long)P * P * xs;
#pragma unroll
            for (int j = 0; j < P; ++j) store_split<T>(oh + (long)(i * P + j) * xs, ol + (long)(i * P + j) * xs, v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                *(T*)(o + (long)(i * P + j) * xs) = v[j];
                if (BF == 0) {
#pragma unroll
                    for (int k = 0; k < VW; ++k) vmax = fmaxf(vmax, fabsf(v[j][k]));
                }
            }
        }
    }
    if (BF == 0 && amax_out) block_amax(vmax, amax_out);      // launcher: whole blocks only
    if ((BF == 3 || BF == 5) && fused && nm.amax_out) ss_block_amax_to_slot(n_am, nm.amax_out);      // max|normalised x| for the weight gradient's scale
}

// E[xi][tile][c] = (A e A^T)_xi for the RxR tile e of dy (zero outside the dy extent)
// PL = 1: E as two fp16 planes [plane][xi][tile][c] of E * 2^(14 - (exponent(max|dy|) + bound)) (weight gradient on pre-split planes)
template <int R, int PL = 0, typename TI = float>
__global__ __launch_bounds__(256) void wino_dy_kernel(const TI* __restrict__ dy, int dy_cs, int N, int OH, int OW, int C,
                                                      int TH, int TW, float* __restrict__ E, unsigned int* __restrict__ amax_out = nullptr,
                                                      const unsigned int* __restrict__ amax_in = nullptr, int amax_stripes = 0, int bound = 0,
                                                      const float* __restrict__ row_scale = nullptr,
                                                      const unsigned int* __restrict__ amax2 = nullptr, int amax2_stripes = 0, int row_bound = 0) {
    typedef typename WT<R>::T T;
    constexpr int VW = WT<R>::VW, P = R + 2;
    const int CV = C / VW;
    const long tiles = (long)N * TH * TW;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tiles * CV) return;
    const int c = (int)(e % CV) * VW;
    const long tile = e / CV;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    T a[P][R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        T v[R], w[P];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int oy = R * ty + i, ox = R * tx + j;
            v[i] = ldz_t<T, TI>(dy, ((long)(n * OH + oy) * OW + ox) * dy_cs + c, oy < OH && ox < OW);
        }
        t_dy<R, T>(v, w);
#pragma unroll
        for (int i = 0; i < P; ++i) a[i][j] = w[i];
    }
    const long xs = tiles * C;
    float* o = E + tile * C + c;
    float emax = 0.f;
    float sc = 1.f;
    if (PL == 1) sc = ldexpf(1.f, 14 - (ss_amax_exp(__uint_as_float(ss_amax_load(amax_in, amax_stripes))) + bound));
    if (PL == 1 && row_scale) {
        // The other operand of the weight-gradient GEMM is the FORWARD pass's V planes, which carry 1 / tile_inv[tile] per row
        // (conv_wino.hip fwd_impl, ss_conv_desc::saved_operand): its factor goes into this row, E' = E * tile_inv[tile] -- a power of
        // two, exact -- and the common scale covers max|E'| <= max|dy| 2^bound * max tile_inv, tile_inv <= 2^(e_x + row_bound - 14)
        // (|V| <= 2^row_bound max|x|).  The clamp only matters for an all-zero tile (tile_inv = 1, V = 0: any finite factor does).
        const int ex = ss_amax_exp(__uint_as_float(ss_amax_load(amax2, amax2_stripes))) + row_bound - 14;
        sc = ldexpf(sc, -ex) * fminf(row_scale[tile], ldexpf(1.f, ex));
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        T w[P];
        t_dy<R, T>(a[i], w);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (PL == 1) {
                unsigned int* d = (unsigned int*)((unsigned short*)E + tile * C + c) + (long)(i * P + j) * xs / 2;
                const long pl32 = (long)P * P * xs / 2;
#pragma unroll
                for (int k = 0; k < VW; k += 2) {
                    ss_split_h2s(w[j][k] * sc, w[j][k + 1] * sc, d[k / 2], d[pl32 + k / 2]);
                }
            } else {
                *(T*)(o + (long)(i * P + j) * xs) = w[j];
#pragma unroll
                for (int k = 0; k < VW; ++k) emax = fmaxf(emax, fabsf(w[j][k]));
            }
        }
    }
    if (PL == 0 && amax_out) block_amax(emax, amax_out);      // launcher: whole blocks only
}

// U[xi][kr][no] = (G g G^T)_xi with g[kh][kw] = w[kh'][kw'][..]; flip = 0: (kr,no) = (ci,co); flip = 1 (backward-data):
// (kh',kw') = (2-kh,2-kw), (kr,no) = (co,ci).  w is the Keras (3,3,cin,cout) kernel.
template <int R, bool BF>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int Cin, int Cout, int flip, float* __restrict__ U) {
    constexpr int P = R + 2;
    const int KR = flip ? Cout : Cin, NO = flip ? Cin : Cout;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)KR * NO) return;
    const int no = (int)(e % NO), kr = (int)(e / NO);
    float t[P][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        float g[3], u[P];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int kh = flip ? 2 - a : a, kw = flip ? 2 - b : b;
            const int ci = flip ? no : kr, co = flip ? kr : no;
            g[a] = w[((long)(kh * 3 + kw) * Cin + ci) * Cout + co];
        }
        t_w<R>(g, u);
#pragma unroll
        for (int i = 0; i < P; ++i) t[i][b] = u[i];
    }
    const long xs = (long)KR * NO;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        float u[P];
        t_w<R>(t[i], u);
        if (BF) {       // transposed planes U^T[xi][no][kr] (K-contiguous rows for the split-bf16 GEMM): hi plane, then lo plane
            unsigned short* uh = (unsigned short*)U + (long)no * KR + kr;
            unsigned short* ul = uh + (long)P * P * xs;
#pragma unroll
            for (int j = 0; j < P; ++j) split_bf16(u[j], uh[(long)(i * P + j) * xs], ul[(long)(i * P + j) * xs]);
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) U[(long)(i * P + j) * xs + e] = u[j];
        }
    }
}

// Same transform, emitted directly as the x6 GEMM's B operand: three K-contiguous bf16 planes
// planes[pl][xi][no (padded to Npad, zero rows)][kr].  A block owns a 32 (kr) x 32 (no) tile of the 3x3 kernel: the 9 taps are
// staged in LDS with loads that are contiguous in memory whichever index that is (co: `no` for flip = 0, `kr` for flip = 1 --
// a thread-per-output version read the 9 MiB kernel tensor at 2 KB strides: 8x over-fetch, 20 us), then one thread = two
// consecutive kr of one no: a packed bf16 pair per store, consecutive threads -> consecutive kr.
// F16 (x3h): two fp16 planes U*s = h + 2^-11 l with ONE power-of-two scale s per launch, from the largest |w| of the kernel tensor
// (|U| <= max|w|: every row of G has absolute sum <= 1); 1/s goes to *w_inv for the output transform.
__global__ __launch_bounds__(256) void amax_bits_kernel(const float* __restrict__ v, long n, unsigned int* __restrict__ out) {
    unsigned int m = 0;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 t = ((const f32x4*)v)[i];               // kernel tensors are 16-byte aligned (parameter arena)
        m = max(max(m, __float_as_uint(fabsf(t[0]))), max(__float_as_uint(fabsf(t[1])), max(__float_as_uint(fabsf(t[2])), __float_as_uint(fabsf(t[3])))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(fabsf(v[(n4 << 2) + threadIdx.x])));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));   // non-negative floats order like their bits
    __shared__ unsigned int wm[4];          // ONE atomic per block (thousands of atomics on one address serialise in L2)
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(wm[0], wm[1]), max(wm[2], wm[3])));      // order-independent: deterministic
}

template <int R, bool F16 = false>
__device__ __forceinline__ void wino_weight_x6_body(const float* __restrict__ w, int Cin, int Cout, int flip, int Npad,
                                                    unsigned short* __restrict__ planes,
                                                    const unsigned int* __restrict__ amax_bits, float* __restrict__ w_inv,
                                                    int plain_l, int bx, int by, int bz) {
    constexpr int P = R + 2, HP = P / 2;
    __shared__ float tl[9][32][17];        // [logical tap a*3+b][kr][no]
    const int KR = flip ? Cout : Cin, NO = flip ? Cin : Cout;
    const int K2 = KR / 2;
    const int kr0 = bx * 32, no0 = by * 16;
    const int i0 = bz * HP;        // this block's rows of the (R+2) x (R+2) transform
    const int tid = threadIdx.x;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int a = t / 3, b2 = t % 3;
        const int kh = flip ? 2 - a : a, kw = flip ? 2 - b2 : b2;
        const float* wt = w + (long)(kh * 3 + kw) * Cin * Cout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // the memory-contiguous index (co) goes to the fast lanes: flip = 0: (kr, no) = (ci, co);  flip = 1: (kr, no) = (co, ci)
            const int e = tid + 256 * i;
            const int krl = flip ? (e & 31) : (e >> 4), nol = flip ? (e >> 5) : (e & 15);
            const int kr = kr0 + krl, no = no0 + nol;
            float v = 0.f;
            if (no < NO) {
                const int ci = flip ? no : kr, co = flip ? kr : no;
                v = wt[(long)ci * Cout + co];
            }
            tl[t][krl][nol] = v;
        }
    }
    __syncthreads();
    const int k2l = tid & 15, nol = tid >> 4;
    const int no = no0 + nol;
    const long plane_u32 = (long)P * P * Npad * KR / 2;     // u32 (bf16 pair) stride between the three planes
    const long xs2 = (long)Npad * KR / 2;                   // ... between transform positions
    float u2[2][HP][P];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float t[P][3];
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
            float g[3], u[P];
#pragma unroll
            for (int a = 0; a < 3; ++a) g[a] = tl[a * 3 + b2][2 * k2l + half][nol];
            t_w<R>(g, u);
#pragma unroll
            for (int i = 0; i < P; ++i) t[i][b2] = u[i];
        }
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            float row[3];
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) row[b2] = i0 ? t[HP + i][b2] : t[i][b2];
            t_w<R>(row, u2[half][i]);
        }
    }
    unsigned int* dst = (unsigned int*)planes + (long)no * K2 + kr0 / 2 + k2l + (long)(i0 * P) * xs2;
    float sc = 1.f;
    if (F16) {
        const float am = __uint_as_float(*amax_bits);
        const int ex = ss_amax_exp(am);
        sc = ldexpf(1.f, 14 - ex);
        if (bx == 0 && by == 0 && bz == 0 && tid == 0) *w_inv = ldexpf(1.f, ex - 14);
    }
#pragma unroll
    for (int i = 0; i < HP; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const long o = (long)(i * P + j) * xs2;
            if (F16) {
                if (plain_l) ss_split_h2(u2[0][i][j] * sc, u2[1][i][j] * sc, dst[o], dst[o + plane_u32]);
                else ss_split_h2s(u2[0][i][j] * sc, u2[1][i][j] * sc, dst[o], dst[o + plane_u32]);
            } else {
                unsigned int h, m, l;
                ss_split3x2(f32x2{u2[0][i][j], u2[1][i][j]}, h, m, l);      // rows no >= NO were staged as zeros
                dst[o] = h;
                dst[o + plane_u32] = m;
                dst[o + 2 * plane_u32] = l;
            }
        }
}

template <int R, bool F16 = false>
__global__ __launch_bounds__(256) void wino_weight_x6_kernel(const float* __restrict__ w, int Cin, int Cout, int flip, int Npad,
                                                             unsigned short* __restrict__ planes,
                                                             const unsigned int* __restrict__ amax_bits = nullptr, float* __restrict__ w_inv = nullptr,
                                                             int plain_l = 0) {
    wino_weight_x6_body<R, F16>(w, Cin, Cout, flip, Npad, planes, amax_bits, w_inv, plain_l, blockIdx.x, blockIdx.y, blockIdx.z);
}
// the x3h Winograd weight planes of a recorded plan in ONE launch (wprep_batch.hip): a: w_cin, b: w_cout, c: flip, d: Npad, e: plain_l
__global__ __launch_bounds__(256) void wino_weight_x6_batch_kernel(const SsWJob* __restrict__ jobs, const int* __restrict__ map) {
    const SsWJob& j = jobs[map[blockIdx.x]];
    const int l = blockIdx.x - j.blk0;
    wino_weight_x6_body<4, true>(j.src, j.a, j.b, j.c, j.d, (unsigned short*)j.dst, j.amax, (float*)j.dst2, j.e,
                                 l % j.gx, (l / j.gx) % j.gy, l / (j.gx * j.gy));
}

// record (ss_wprep_*) what the x3h weight fill launches: the maximum of the kernel tensor into w_inv[1], then the planes
inline void wino_record_weight_jobs(const float* w, int w_cin, int w_cout, int flip, int Npad, int kr, unsigned short* planes, float* w_inv, int plain_l) {
    SsWJob a{};
    a.src = w; a.n = (long)9 * w_cin * w_cout; a.dst = w_inv + 1;
    a.type = SS_WJ_AMAX; a.gx = (int)(a.n / 16384 < 4 ? 4 : (a.n / 16384 > 64 ? 64 : a.n / 16384)); a.gy = 1; a.gz = 1;
    ss_wrec_push(a);
    SsWJob j{};
    j.type = SS_WJ_WINO_H; j.gx = kr / 32; j.gy = Npad / 16; j.gz = 2;
    j.src = w; j.a = w_cin; j.b = w_cout; j.c = flip; j.d = Npad; j.e = plain_l;
    j.dst = planes; j.dst2 = w_inv; j.amax = (const unsigned int*)(w_inv + 1);
    ss_wrec_push(j);
}

// y[n, R*ty+i, R*tx+j, c] (+)= act(bias + (A^T M A)_ij)
// TM: type the Winograd-domain product is stored in (float; _Float16 under the scale 1 / m_scale: the one-plane 16-bit path, wino16_m16)
template <int R, typename TO = float, typename TM = float>
__global__ __launch_bounds__(256) void wino_output_kernel(const TM* __restrict__ Mx, int N, int OH, int OW, int C, int TH, int TW,
                                                          const float* __restrict__ bias, int act, float alpha,
                                                          TO* __restrict__ y, int y_cs, int accumulate, int FH, int FW,
                                                          const float* __restrict__ tile_inv = nullptr, const float* __restrict__ w_inv = nullptr,
                                                          float* __restrict__ stats = nullptr, float m_scale = 1.f) {
    // FH > 0 ("reflect fold", data gradient of reflect-pad(1) + 3x3 valid conv): the OH x OW grid is the PADDED gradient shifted by
    // one (virtual o' = P + 1, P in [0, FH+1]); padded pixel P lands on dx[reflect(P - 1)].  With FH % R == 0 the two padded
    // rows that fold onto the same dx row (P = 0,2 and P = FH-1,FH+1) sit in ONE tile, i.e. one thread: they are summed in
    // registers and every dx pixel is written exactly once -- no padded scratch tensor, no separate fold pass.
    typedef typename WT<R>::T T;
    constexpr int VW = WT<R>::VW, P = R + 2;
    const int CV = C / VW;
    const long tiles = (long)N * TH * TW;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tiles * CV) return;
    const int c = (int)(e % CV) * VW;
    const long tile = e / CV;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    const long xs = tiles * C;
    const TM* m = Mx + tile * C + c;
    const float unscale = (tile_inv ? tile_inv[tile] * w_inv[0] : 1.f) * m_scale;      // x3h operands carry power-of-two scales (exact to undo)
    T s[R][P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        T col[P], o[R];
#pragma unroll
        for (int i = 0; i < P; ++i) col[i] = ld_t<T, TM>(m + (long)(i * P + j) * xs) * unscale;
        t_out<R, T>(col, o);
#pragma unroll
        for (int i = 0; i < R; ++i) s[i][j] = o[i];
    }
    T bv = zero_v<T>();
    if (bias) bv = *(const T*)(bias + c);
    if (FH > 0) {
        T o[R][R];
        int dy_[R], dx_[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            t_out<R, T>(s[i], o[i]);
            const int py = R * ty + i - 1, px = R * tx + i - 1;          // padded coordinates of row i / column i
            dy_[i] = (py < 0 || py > FH + 1) ? -1 : (py == 0 ? 1 : (py == FH + 1 ? FH - 2 : py - 1));
            dx_[i] = (px < 0 || px > FW + 1) ? -1 : (px == 0 ? 1 : (px == FW + 1 ? FW - 2 : px - 1));
        }
#pragma unroll
        for (int i = 1; i < R; ++i)
#pragma unroll
            for (int i2 = 0; i2 < i; ++i2) {
                if (dy_[i] >= 0 && dy_[i2] == dy_[i]) {                 // rows folding onto the same dx row
#pragma unroll
                    for (int j = 0; j < R; ++j) o[i2][j] += o[i][j];
                    dy_[i] = -1;
                }
                if (dx_[i] >= 0 && dx_[i2] == dx_[i]) {
#pragma unroll
                    for (int k = 0; k < R; ++k) o[k][i2] += o[k][i];
                    dx_[i] = -1;
                }
            }
        // `accumulate` as a compile-time constant inside the store loop: a conditional `v += load` there makes the compiler wait for
        // vmcnt(0) around every store (16 serial round trips per thread)
        auto fold_stores = [&](auto acc_c) {
            constexpr bool ACC = decltype(acc_c)::value;
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (dy_[i] < 0 || dx_[j] < 0) continue;
                    T v = o[i][j];
                    TO* dst = y + ((long)(n * FH + dy_[i]) * FW + dx_[j]) * y_cs + c;
                    if (ACC) v += ld_t<T, TO>(dst);
                    st_t<T, TO>(dst, v);
                }
        };
        if (accumulate) fold_stores(std::true_type{}); else fold_stores(std::false_type{});
        return;
    }
    float s1[VW], s2[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    auto plain_stores = [&](auto acc_c, auto act_c) {          // compile-time inside the store loop, as above
        constexpr bool ACC = decltype(acc_c)::value, ACT = decltype(act_c)::value;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            T o[R];
            t_out<R, T>(s[i], o);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int oy = R * ty + i, ox = R * tx + j;
                if (oy < OH && ox < OW) {
                    T v = o[j] + bv;
                    if (ACT) {
#pragma unroll
                        for (int k = 0; k < VW; ++k) v[k] = ss_apply_act(v[k], act, alpha);
                    }
                    TO* dst = y + ((long)(n * OH + oy) * OW + ox) * y_cs + c;
                    if (ACC) v += ld_t<T, TO>(dst);
                    st_t<T, TO>(dst, v);
#pragma unroll
                    for (int k = 0; k < VW; ++k) { s1[k] += v[k]; s2[k] = fmaf(v[k], v[k], s2[k]); }
                }
            }
        }
    };
    if (accumulate) { if (act != SS_ACT_NONE) plain_stores(std::true_type{}, std::true_type{}); else plain_stores(std::true_type{}, std::false_type{}); }
    else { if (act != SS_ACT_NONE) plain_stores(std::false_type{}, std::true_type{}); else plain_stores(std::false_type{}, std::false_type{}); }
    if (stats) {
        // statistics of what was just written, for the norm that follows: a block holds 256 / CV whole tiles of ONE sample (launcher);
        // their sums are combined in fixed order and written as one chunk [n][chunk][C][2] (the layout of norm.hip's partials)
        __shared__ float red[256][2 * VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) { red[threadIdx.x][2 * k] = s1[k]; red[threadIdx.x][2 * k + 1] = s2[k]; }
        __syncthreads();
        if ((int)threadIdx.x < CV) {
            float a[2 * VW];
#pragma unroll
            for (int k = 0; k < 2 * VW; ++k) a[k] = red[threadIdx.x][k];
            for (int t = 1; t < 256 / CV; ++t)
#pragma unroll
                for (int k = 0; k < 2 * VW; ++k) a[k] += red[t * CV + threadIdx.x][k];
            const int bps = TH * TW * CV / 256;            // blocks (= chunks) per sample
            float* o = stats + ((long)blockIdx.x * C + c) * 2;            // blockIdx.x = n * bps + chunk
            (void)bps;
#pragma unroll
            for (int k = 0; k < 2 * VW; ++k) o[k] = a[k];
        }
    }
}

// dw[kh][kw][ci][co] (+)= (G^T S G) with S_xi[ci][co] = sum_splits part[xi][split][ci][co]
// SP > 0: the split count as a compile-time constant (all loads of a column in flight at once; the runtime loop serialised them:
// 73 us for 151 MB), same summation order
template <int R, int SP = 0>
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ part, int splits, int Cin, int Cout, float* __restrict__ dw,
                                                      int accumulate) {
    constexpr int P = R + 2;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long cc = (long)Cin * Cout;
    if (e >= cc) return;
    float t[3][P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        float col[P], o[3];
        if (SP > 0) {
            float v[P][SP > 0 ? SP : 1];
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) v[i][sp] = part[((long)(i * P + j) * SP + sp) * cc + e];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int sp = 0; sp < SP; ++sp) acc += v[i][sp];
                col[i] = acc;
            }
        } else {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            float acc = 0.f;
            for (int sp = 0; sp < splits; ++sp) acc += part[((long)(i * P + j) * splits + sp) * cc + e];
            col[i] = acc;
        }
        }
        t_dw<R>(col, o);
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a][j] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float o[3];
        t_dw<R>(t[a], o);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float* d = dw + (long)(a * 3 + b) * cc + e;
            *d = accumulate ? (*d + o[b]) : o[b];
        }
    }
}

inline unsigned g256(long total) { return (unsigned)((total + 255) / 256); }

template <int R>
void launch_wino_dw(const float* part, int splits, int cin, int cout, float* dw, int accumulate, hipStream_t s) {
    const dim3 grid(g256((long)cin * cout)), block(256);
    switch (splits) {
        case 1: hipLaunchKernelGGL((wino_dw_kernel<R, 1>), grid, block, 0, s, part, splits, cin, cout, dw, accumulate); break;
        case 2: hipLaunchKernelGGL((wino_dw_kernel<R, 2>), grid, block, 0, s, part, splits, cin, cout, dw, accumulate); break;
        case 3: hipLaunchKernelGGL((wino_dw_kernel<R, 3>), grid, block, 0, s, part, splits, cin, cout, dw, accumulate); break;
        case 4: hipLaunchKernelGGL((wino_dw_kernel<R, 4>), grid, block, 0, s, part, splits, cin, cout, dw, accumulate); break;
        default: hipLaunchKernelGGL((wino_dw_kernel<R, 0>), grid, block, 0, s, part, splits, cin, cout, dw, accumulate); break;
    }
}

// output tile size: F(4x4,3x3) by default (4x fewer multiplies), F(2x2,3x3) with SS_WINO_R=2
inline int wino_r() { return ss_tuning().wino_r; }
inline long n_tiles(const WinoProb& q, int R) { return (long)q.n * ((q.oh + R - 1) / R) * ((q.ow + R - 1) / R); }

// 16-bit activation storage (q.dtype != F32): x / y are TS arrays; the x3h plane path runs with ONE fp16 plane per operand and one
// product (gemm_x6p_kernel<1>): the input transform reads the stored type, the output transform writes it -- no fp32 staging copies
template <int R, typename TS>
int fwd16_impl(const WinoProb& q, const TS* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, TS* y,
               int act, float alpha, int accumulate, void* ws, hipStream_t s) {
    constexpr int XI = (R + 2) * (R + 2), VW = WT<R>::VW;
    if (R != 4 || !ss_wino_fwd_x3h(q) || (((uintptr_t)w) & 15)) return SS_ERR_UNSUPPORTED;
    if (q.in_norm.groups > 0 && (q.cin % 32 || (q.in_norm.act != SS_ACT_NONE && q.in_norm.act != SS_ACT_RELU && q.in_norm.act != SS_ACT_LRELU))) return SS_ERR_UNSUPPORTED;
    const int TH = (q.oh + R - 1) / R, TW = (q.ow + R - 1) / R;
    const long tiles = (long)q.n * TH * TW;
    const int cvi = q.cin / VW;
    float* V = (float*)((char*)ws + ss_align_up((size_t)XI * q.cin * q.cout * 4, 256));
    const long Mpad = (tiles + SS_X6P_BM - 1) / SS_X6P_BM * SS_X6P_BM;
    const int Npad = ss_x6_npad(q.cout);
    float* Mx = (float*)((char*)V + ss_align_up((size_t)3 * XI * Mpad * q.cin * 2, 256));
    unsigned short* planes = (unsigned short*)((char*)Mx + ss_align_up((size_t)XI * tiles * q.cout * 4, 256));
    char* extra = (char*)planes + ss_align_up((size_t)3 * XI * Npad * q.cin * 2, 256);
    float* tile_inv = (float*)(extra + 256);
    const uint64_t wdet = (uint64_t)(flip ? 1 : 0) | ((uint64_t)R << 1);          // the SAME weight planes as the fp32-storage x3h path (plane 0 = h)
    bool fill, fill2;
    float* w_inv = (float*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X3H_INV, wdet), 256, extra, &fill);
    planes = (unsigned short*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X3H_PLANES, wdet), (size_t)3 * XI * Npad * q.cin * 2, planes, &fill2);
    if ((fill || fill2) && ss_wrec_on() && R == 4) {
        wino_record_weight_jobs(w, w_cin, w_cout, flip, Npad, q.cin, planes, w_inv, 0);
    } else if (fill || fill2) {
        (void)hipMemsetAsync(w_inv + 1, 0, 4, s);
        hipLaunchKernelGGL(amax_bits_kernel, dim3(256), dim3(256), 0, s, w, (long)9 * w_cin * w_cout, (unsigned int*)(w_inv + 1));
        SS_LAUNCH_CHECK();
        hipLaunchKernelGGL((wino_weight_x6_kernel<R, true>), dim3(q.cin / 32, Npad / 16, 2), dim3(256), 0, s, w, w_cin, w_cout, flip, Npad, planes,
                           (const unsigned int*)(w_inv + 1), w_inv, 0);
        SS_LAUNCH_CHECK();
    }
    if (q.wc && q.wc->fill_only) return SS_OK;
    const bool one = ss_tuning().wino16_products != 3;          // 1: one plane / one product; 3: the x3h arithmetic of the fp32-storage path
    {
        // algorithmic bytes (HBM roofline, bench.py): one read of x, one write of the fp16 operand plane(s)
        SsProfScope prof(q.in_norm.groups > 0 ? "wino_input_kernel<normalising>" : "wino_input_kernel", 0.0, (double)q.n * q.h * q.w * q.cin * sizeof(TS) + (double)XI * tiles * q.cin * 2 * (one ? 1 : 2), s);
        if (q.in_norm.groups > 0 && one)          // x is the PRE-norm tensor: normalised (and rounded to the stored type) in the load
            hipLaunchKernelGGL((wino_input_kernel<R, 5, true, TS>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                               TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv, (unsigned int*)nullptr, (const unsigned int*)nullptr, 0, 0, 0, q.in_norm);
        else if (q.in_norm.groups > 0)
            hipLaunchKernelGGL((wino_input_kernel<R, 3, true, TS>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                               TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv, (unsigned int*)nullptr, (const unsigned int*)nullptr, 0, 0, 0, q.in_norm);
        else if (one)
            hipLaunchKernelGGL((wino_input_kernel<R, 5, false, TS>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                               TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv);
        else
            hipLaunchKernelGGL((wino_input_kernel<R, 3, false, TS>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                               TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv);
        SS_LAUNCH_CHECK();
    }
    X6PParams g{};
    g.fp16x2 = one ? 2 : 1;
    g.a = (const unsigned short*)V; g.b = planes; g.c = Mx;
    g.M = (int)tiles; g.N = q.cout; g.K = q.cin; g.nbatch = XI; g.splits = 1; g.k_per_split = q.cin;
    g.lda = q.cin; g.ldb = q.cin; g.ldc = q.cout;
    g.a_plane = (long)XI * Mpad * q.cin; g.a_bs = Mpad * q.cin;
    g.b_plane = (long)XI * Npad * q.cin; g.b_bs = (long)Npad * q.cin;
    g.c_bs = tiles * q.cout; g.c_ss = 0;
    // one plane per operand: the product may be stored as fp16 as well (wino16_m16) -- |sum| <= 2^14 * 2^14 * K, scaled into fp16's range
    // by a fixed power of two (exact; typical sums sit ~2^10 below the bound, fp16 keeps 11 bits down to 2^-30 of it), undone in the
    // output transform: half the bytes of the product's round trip, which is what paces the one-plane GEMM
    const bool m16 = one && ss_tuning().wino16_m16;
    int kexp = 0;
    while ((1 << kexp) < q.cin) ++kexp;
    if (m16) { g.c16 = ss_tuning().wino16_m16 == 2 ? 2 : 1; g.c_scale = ldexpf(1.f, -(28 + kexp - 15)); }          // (2, measurement: 8-byte product stores)
    const int rcx = ss_launch_gemm_x6p(g, s);
    if (rcx != SS_OK) return rcx;
    SsProfScope prof("wino_output_kernel", 0.0, (double)XI * tiles * q.cout * (m16 ? 2 : 4) + (double)q.n * q.oh * q.ow * q.cout * sizeof(TS) * (accumulate ? 2 : 1), s);
    if (m16)
        hipLaunchKernelGGL((wino_output_kernel<R, TS, _Float16>), dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, (const _Float16*)Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                           bias, act, alpha, y, q.out_cs, accumulate, q.fold_h, q.fold_w, tile_inv, w_inv, (float*)nullptr, 1.f / g.c_scale);
    else
        hipLaunchKernelGGL((wino_output_kernel<R, TS>), dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                           bias, act, alpha, y, q.out_cs, accumulate, q.fold_h, q.fold_w, tile_inv, w_inv, (float*)nullptr);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <int R>
int fwd_impl(const WinoProb& q, const float* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, float* y,
             int act, float alpha, int accumulate, void* ws, hipStream_t s) {
    constexpr int XI = (R + 2) * (R + 2), VW = WT<R>::VW;
    const int TH = (q.oh + R - 1) / R, TW = (q.ow + R - 1) / R;
    const long tiles = (long)q.n * TH * TW;
    float* U = (float*)ws;
    float* V = (float*)((char*)ws + ss_align_up((size_t)XI * q.cin * q.cout * 4, 256));
    float* Mx = (float*)((char*)V + ss_align_up((size_t)XI * tiles * q.cin * 4, 256));
    // output statistics for a following norm: only where the output transform's blocks hold whole tiles of one sample
    float* stats = (q.y_stats && act == SS_ACT_NONE && !accumulate && ss_wino_stats_chunks(q) > 0) ? q.y_stats : nullptr;
    bool fill;      // transformed weights: in the layer's cache when there is one (computed on first use)
    const uint64_t wdet0 = (uint64_t)(flip ? 1 : 0) | ((uint64_t)R << 1);
    const uint64_t wdet = wdet0;
    const bool fill_only = q.wc && q.wc->fill_only;       // refresh of the cached operands: no activation work
    if (q.bf16x3) {
        U = (float*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_UBF, wdet), (size_t)XI * q.cin * q.cout * 4, U, &fill);
        if (fill) {
            if (ss_wrec_on()) ss_wrec_unbatched();
            hipLaunchKernelGGL((wino_weight_kernel<R, true>), dim3(g256((long)q.cin * q.cout)), dim3(256), 0, s, w, w_cin, w_cout, flip, U);
            SS_LAUNCH_CHECK();
        }
        if (fill_only) return SS_OK;
        hipLaunchKernelGGL((wino_input_kernel<R, true>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                           TH, TW, q.pt, q.pl, q.reflect, V);
        SS_LAUNCH_CHECK();
        BGemmParams b{};
        b.ah = (const unsigned short*)V; b.al = b.ah + (long)XI * tiles * q.cin;
        b.bh = (const unsigned short*)U; b.bl = b.bh + (long)XI * q.cin * q.cout;
        b.c = Mx; b.M = (int)tiles; b.N = q.cout; b.K = q.cin; b.nbatch = XI;
        b.a_bs = tiles * q.cin; b.b_bs = (long)q.cin * q.cout; b.c_bs = tiles * q.cout;
        b.lda = q.cin; b.ldb = q.cin; b.ldc = q.cout;
        int rcb = ss_launch_bgemm_bf16x3(b, s);
        if (rcb != SS_OK) return rcb;
        hipLaunchKernelGGL(wino_output_kernel<R>, dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                           bias, act, alpha, y, q.out_cs, accumulate, q.fold_h, q.fold_w);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    // x3h: fp16 two-piece operands (three products instead of six) where a tile's channels are whole waves of the input transform
    const int cvi = q.cin / VW;
    const bool x3h = R == 4 && ss_wino_fwd_x3h(q) && (((uintptr_t)w) & 15) == 0;
    if (q.in_norm.groups > 0 && !(x3h && q.cin % 32 == 0)) return SS_ERR_UNSUPPORTED;      // only the x3h plane path normalises in its load
    if (q.saved && ss_wino_saved_bytes(q) > 0 && !x3h) {          // promised (ss_conv2d_saved_operand_bytes) but this launch cannot write it: fail loudly
        ss_set_error("conv2d_fwd: saved_operand was promised but this launch does not take the x3h plane path (weight pointer alignment)");
        return SS_ERR_UNSUPPORTED;
    }
    if (q.in_norm.groups > 0 && q.in_norm.act != SS_ACT_NONE && q.in_norm.act != SS_ACT_RELU && q.in_norm.act != SS_ACT_LRELU) return SS_ERR_UNSUPPORTED;
    if (q.x6 && q.cin % 32 == 0 && (x3h || ss_x6p_wanted(tiles, q.cout, XI))) {
        // both GEMM operands as pre-split bf16 planes: V planes in the V region (1.5x the fp32 size, see ss_wino_fwd_ws)
        const long Mpad = (tiles + SS_X6P_BM - 1) / SS_X6P_BM * SS_X6P_BM;
        const int Npad = ss_x6_npad(q.cout);
        Mx = (float*)((char*)V + ss_align_up((size_t)3 * XI * Mpad * q.cin * 2, 256));
        unsigned short* planes = (unsigned short*)((char*)Mx + ss_align_up((size_t)XI * tiles * q.cout * 4, 256));
        float* tile_inv = nullptr;
        float* w_inv = nullptr;
        // 256-multiple output channels: 256 x 256 GEMM tiles on planes with the plain low piece (gemm_x6p.hip WIDE)
        const int wide = x3h && ss_x6p_wide_ok(tiles, q.cout, q.cin, XI) ? 1 : 0;
        const uint64_t wdet = wdet0 | ((uint64_t)wide << 8);
        if (x3h) {
            char* extra = (char*)planes + ss_align_up((size_t)3 * XI * Npad * q.cin * 2, 256);
            tile_inv = (float*)(extra + 256);
            w_inv = (float*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X3H_INV, wdet), 256, extra, &fill);                              // [0]: 1/s_w, [1]: max|w| bits
            bool fill2;
            planes = (unsigned short*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X3H_PLANES, wdet), (size_t)3 * XI * Npad * q.cin * 2, planes, &fill2);
            if ((fill || fill2) && ss_wrec_on() && R == 4) {
            wino_record_weight_jobs(w, w_cin, w_cout, flip, Npad, q.cin, planes, w_inv, wide);
            } else if (fill || fill2) {
            (void)hipMemsetAsync(w_inv + 1, 0, 4, s);
            hipLaunchKernelGGL(amax_bits_kernel, dim3(256), dim3(256), 0, s, w, (long)9 * w_cin * w_cout, (unsigned int*)(w_inv + 1));
            SS_LAUNCH_CHECK();
            hipLaunchKernelGGL((wino_weight_x6_kernel<R, true>), dim3(q.cin / 32, Npad / 16, 2), dim3(256), 0, s, w, w_cin, w_cout, flip, Npad, planes,
                               (const unsigned int*)(w_inv + 1), w_inv, wide);
            SS_LAUNCH_CHECK();
            }
            if (fill_only) return SS_OK;
            if (q.saved && ss_wino_saved_bytes(q) > 0) {
                // the caller keeps this pass's input planes for the weight gradient (ss_conv_desc::saved_operand): [2 planes][XI][Mpad][cin]
                // fp16 + the per-tile scales, instead of the workspace copies
                V = (float*)q.saved;
                tile_inv = (float*)((char*)q.saved + ss_align_up((size_t)2 * XI * Mpad * q.cin * 2, 256));
            }
            {
                SsProfScope prof(q.in_norm.groups > 0 ? "wino_input_kernel<normalising>" : "wino_input_kernel", 0.0,
                                 (double)q.n * q.h * q.w * q.cin * 4 + (double)XI * tiles * q.cin * 4, s);          // read x, write the (h, l) planes
                if (q.in_norm.groups > 0)
                    hipLaunchKernelGGL((wino_input_kernel<R, 3, true>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                                       TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv, (unsigned int*)nullptr, (const unsigned int*)nullptr, 0, 0, wide,
                                       q.in_norm);
                else
                    hipLaunchKernelGGL((wino_input_kernel<R, 3>), dim3(g256(tiles * cvi)), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                                       TH, TW, q.pt, q.pl, q.reflect, V, Mpad, tile_inv, (unsigned int*)nullptr, (const unsigned int*)nullptr, 0, 0, wide);
                SS_LAUNCH_CHECK();
            }
        } else {
        planes = (unsigned short*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X6_PLANES, wdet), (size_t)3 * XI * Npad * q.cin * 2, planes, &fill);
        if (fill) {
            if (ss_wrec_on()) ss_wrec_unbatched();          // (only the x3h operands are replayed from a recorded plan)
            hipLaunchKernelGGL(wino_weight_x6_kernel<R>, dim3(q.cin / 32, Npad / 16, 2), dim3(256), 0, s, w, w_cin, w_cout, flip, Npad, planes);
            SS_LAUNCH_CHECK();
        }
        if (fill_only) return SS_OK;
        hipLaunchKernelGGL((wino_input_kernel<R, 2>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin,
                           TH, TW, q.pt, q.pl, q.reflect, V, Mpad);
        SS_LAUNCH_CHECK();
        }
        X6PParams g{};
        g.fp16x2 = x3h ? 1 : 0;
        g.plain_l = wide;
        g.a = (const unsigned short*)V; g.b = planes; g.c = Mx;
        g.M = (int)tiles; g.N = q.cout; g.K = q.cin; g.nbatch = XI; g.splits = 1; g.k_per_split = q.cin;
        g.lda = q.cin; g.ldb = q.cin; g.ldc = q.cout;
        g.a_plane = (long)XI * Mpad * q.cin; g.a_bs = Mpad * q.cin;
        g.b_plane = (long)XI * Npad * q.cin; g.b_bs = (long)Npad * q.cin;
        g.c_bs = tiles * q.cout; g.c_ss = 0;
        const int rcx = ss_launch_gemm_x6p(g, s);
        if (rcx != SS_OK) return rcx;
        SsProfScope prof("wino_output_kernel", 0.0, (double)XI * tiles * q.cout * 4 + (double)q.n * q.oh * q.ow * q.cout * 4 * (accumulate ? 2 : 1), s);
        hipLaunchKernelGGL(wino_output_kernel<R>, dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                           bias, act, alpha, y, q.out_cs, accumulate, q.fold_h, q.fold_w, tile_inv, w_inv, stats);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    if (!fill_only) {
        hipLaunchKernelGGL((wino_input_kernel<R, false>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                           q.pt, q.pl, q.reflect, V);
        SS_LAUNCH_CHECK();
    }
    GConvParams g{};
    g.in = V; g.w = U; g.bias = nullptr; g.out = Mx;
    g.N = 1; g.IH = 1; g.IW = (int)tiles; g.Cin = q.cin; g.in_cs = q.cin;
    g.OHc = 1; g.OWc = (int)tiles; g.in_s = 1; g.in_oy = 0; g.in_ox = 0;
    g.OH = 1; g.OW = (int)tiles; g.Cout = q.cout; g.out_cs = q.cout; g.out_s = 1; g.out_oy = 0; g.out_ox = 0;
    g.ldb = q.cout; g.reflect = 0; g.act = SS_ACT_NONE; g.alpha = 0.f; g.accumulate = 0;
    g.nbatch = XI; g.in_bs = tiles * q.cin; g.w_bs = (long)q.cin * q.cout; g.out_bs = tiles * q.cout;
    g.ntaps = 1; g.taps[0].dy = 0; g.taps[0].dx = 0; g.taps[0].woff = 0;
    int rc;
    if (q.x6 && ss_gconv_x6_ok(g)) {     // fp32-exact GEMMs on the bf16 matrix cores: the weight transform emits the B planes
        unsigned short* planes = (unsigned short*)((char*)Mx + ss_align_up((size_t)XI * tiles * q.cout * 4, 256));
        const int Npad = ss_x6_npad(q.cout);
        planes = (unsigned short*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_X6_PLANES, wdet), (size_t)3 * XI * Npad * q.cin * 2, planes, &fill);
        if (fill) {
            if (ss_wrec_on()) ss_wrec_unbatched();          // (only the x3h operands are replayed from a recorded plan)
            hipLaunchKernelGGL(wino_weight_x6_kernel<R>, dim3(q.cin / 32, Npad / 16, 2), dim3(256), 0, s, w, w_cin, w_cout, flip, Npad, planes);
            SS_LAUNCH_CHECK();
        }
        if (fill_only) return SS_OK;
        rc = ss_launch_gconv_x6(g, planes, s);
    } else {
        U = (float*)ss_wc_region(q.wc, ss_wc_tag(SS_WC_WINO_U32, wdet), (size_t)XI * q.cin * q.cout * 4, U, &fill);
        g.w = U;
        if (fill) {
            if (ss_wrec_on()) ss_wrec_unbatched();
            hipLaunchKernelGGL((wino_weight_kernel<R, false>), dim3(g256((long)q.cin * q.cout)), dim3(256), 0, s, w, w_cin, w_cout, flip, U);
            SS_LAUNCH_CHECK();
        }
        if (fill_only) return SS_OK;
        rc = ss_launch_gconv_mfma(g, s);
    }
    if (rc != SS_OK) return rc;
    hipLaunchKernelGGL(wino_output_kernel<R>, dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                       bias, act, alpha, y, q.out_cs, accumulate, q.fold_h, q.fold_w, (const float*)nullptr, (const float*)nullptr, stats);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// TS = storage type of x / dy (float, or _Float16 / __bf16: 16-bit activation storage, pre-split-plane path only -- the transforms read
// the stored type, the planes and the GEMM are those of the fp32-storage path: the weight gradient stays fp32-grade)
template <int R, typename TS = float>
int wgrad_impl(const WinoProb& q, const TS* x, const TS* dy, float* dw, int accumulate, void* ws, hipStream_t s) {
    constexpr int XI = (R + 2) * (R + 2), VW = WT<R>::VW;
    const int TH = (q.oh + R - 1) / R, TW = (q.ow + R - 1) / R;
    const long tiles = (long)q.n * TH * TW;
    float* V = (float*)ws;
    float* E = (float*)((char*)ws + ss_align_up((size_t)XI * tiles * q.cin * 4, 256));
    float* part = (float*)((char*)E + ss_align_up((size_t)XI * tiles * q.cout * 4, 256));
    if (q.in_norm.groups > 0 && !(R == 4 && ss_wino_wgrad_tn(q) && q.x_amax && q.dy_amax)) return SS_ERR_UNSUPPORTED;
    if (sizeof(TS) != 4 && !(R == 4 && ss_wino_wgrad_tn(q) && q.x_amax && q.dy_amax)) return SS_ERR_UNSUPPORTED;
    if (R == 4 && ss_wino_wgrad_tn(q) && q.x_amax && q.dy_amax) {
        // both operands as K-major fp16 (h, l) planes, one power-of-two scale per tensor from max|x| / max|dy| and the gain bounds of
        // the transforms (|B^T d B| <= 100 max|d| < 2^7, |A e A^T| <= 225 max|e| < 2^8), GEMM by LDS-DMA + transposing LDS reads
        constexpr int BOUND_X = 7, BOUND_DY = 8;
        if constexpr (sizeof(TS) == 4) {
            if (q.saved && ss_wino_saved_bytes(q) > 0) {
                // The forward pass of this call left its V planes (per-TILE scales, h + 2^-11 l') and the scales' inverses in the caller's
                // buffer: no second input transform.  K = tile is the reduction index here, so the per-row factor tile_inv[k] moves to the
                // other operand: E'[k][:] = E[k][:] * tile_inv[k] (wino_dy_kernel), under one scale that bounds max|E'|.
                const long Mpad = (tiles + SS_X6P_BM - 1) / SS_X6P_BM * SS_X6P_BM;
                const float* tinv = (const float*)((const char*)q.saved + ss_align_up((size_t)2 * XI * Mpad * q.cin * 2, 256));
                {
                    SsProfScope prof("wino_dy_kernel", 0.0, (double)q.n * q.oh * q.ow * q.cout * sizeof(TS) + (double)XI * tiles * q.cout * 4, s);
                    hipLaunchKernelGGL((wino_dy_kernel<R, 1, TS>), dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, dy, q.out_cs, q.n, q.oh, q.ow, q.cout, TH, TW, (float*)E,
                                       (unsigned int*)nullptr, q.dy_amax, q.dy_stripes, BOUND_DY, tinv, q.x_amax, q.x_stripes, BOUND_X);
                    SS_LAUNCH_CHECK();
                }
                TNParams g{};
                g.a = (const unsigned short*)q.saved; g.b = (const unsigned short*)E; g.c = part;
                g.M = q.cin; g.N = q.cout; g.K = (int)tiles; g.nbatch = XI;
                g.lda = q.cin; g.ldb = q.cout;
                g.a_plane = (long)XI * Mpad * q.cin; g.b_plane = (long)XI * tiles * q.cout;
                g.a_bs = Mpad * q.cin; g.b_bs = tiles * q.cout;
                int kps;
                g.splits = ss_gemm_tn_splits(q.cin, q.cout, tiles, XI, &kps);
                g.k_per_split = kps;
                g.a_prescaled = 1;
                g.amax_a = q.x_amax; g.stripes_a = q.x_stripes; g.bound_a = 0;
                g.amax_b = q.dy_amax; g.stripes_b = q.dy_stripes; g.bound_b = BOUND_DY + BOUND_X - 14;
                g.amax_b2 = q.x_amax; g.stripes_b2 = q.x_stripes;
                const int rc = ss_launch_gemm_tn_x3h(g, s);
                if (rc != SS_OK) return rc;
                launch_wino_dw<R>(part, g.splits, q.cin, q.cout, dw, accumulate, s);
                SS_LAUNCH_CHECK();
                return SS_OK;
            }
            SsProfScope prof(q.in_norm.groups > 0 ? "wino_input_kernel<normalising>" : "wino_input_kernel", 0.0,
                             (double)q.n * q.h * q.w * q.cin * 4 + (double)XI * tiles * q.cin * 4, s);
            if (q.in_norm.groups > 0)
                hipLaunchKernelGGL((wino_input_kernel<R, 4, true>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                                   q.pt, q.pl, q.reflect, V, tiles, nullptr, nullptr, q.x_amax, q.x_stripes, BOUND_X, 0, q.in_norm);
            else
                hipLaunchKernelGGL((wino_input_kernel<R, 4>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                                   q.pt, q.pl, q.reflect, V, tiles, nullptr, nullptr, q.x_amax, q.x_stripes, BOUND_X);
        } else {
            SsProfScope prof(q.in_norm.groups > 0 ? "wino_input_kernel<normalising>" : "wino_input_kernel", 0.0, (double)q.n * q.h * q.w * q.cin * sizeof(TS) + (double)XI * tiles * q.cin * 4, s);
            if (q.in_norm.groups > 0)
                hipLaunchKernelGGL((wino_input_kernel<R, 4, true, TS>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                                   q.pt, q.pl, q.reflect, V, tiles, nullptr, nullptr, q.x_amax, q.x_stripes, BOUND_X, 0, q.in_norm);
            else
                hipLaunchKernelGGL((wino_input_kernel<R, 4, false, TS>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                                   q.pt, q.pl, q.reflect, V, tiles, nullptr, nullptr, q.x_amax, q.x_stripes, BOUND_X);
        }
        SS_LAUNCH_CHECK();
        {
            SsProfScope prof("wino_dy_kernel", 0.0, (double)q.n * q.oh * q.ow * q.cout * sizeof(TS) + (double)XI * tiles * q.cout * 4, s);
            hipLaunchKernelGGL((wino_dy_kernel<R, 1, TS>), dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, dy, q.out_cs, q.n, q.oh, q.ow, q.cout, TH, TW, (float*)E,
                               (unsigned int*)nullptr, q.dy_amax, q.dy_stripes, BOUND_DY);
            SS_LAUNCH_CHECK();
        }
        TNParams g{};
        g.a = (const unsigned short*)V; g.b = (const unsigned short*)E; g.c = part;
        g.M = q.cin; g.N = q.cout; g.K = (int)tiles; g.nbatch = XI;
        g.lda = q.cin; g.ldb = q.cout;
        g.a_plane = (long)XI * tiles * q.cin; g.b_plane = (long)XI * tiles * q.cout;
        g.a_bs = tiles * q.cin; g.b_bs = tiles * q.cout;
        int kps;
        g.splits = ss_gemm_tn_splits(q.cin, q.cout, tiles, XI, &kps);
        g.k_per_split = kps;
        g.amax_a = q.x_amax; g.stripes_a = q.x_stripes; g.bound_a = BOUND_X;
        g.amax_b = q.dy_amax; g.stripes_b = q.dy_stripes; g.bound_b = BOUND_DY;
        // 16-bit storage, wino16_products = 1 (the default): the leading planes alone, one product -- the operand rounding the forward
        // pass and the data gradient of these layers already run with (gemm_x6p_kernel<1>); 3: the planes' full 22 bits
        g.one_plane = sizeof(TS) != 4 && ss_tuning().wino16_products != 3 ? 1 : 0;
        const int rc = ss_launch_gemm_tn_x3h(g, s);
        if (rc != SS_OK) return rc;
        launch_wino_dw<R>(part, g.splits, q.cin, q.cout, dw, accumulate, s);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    if constexpr (sizeof(TS) != 4) {
        return SS_ERR_UNSUPPORTED;
    } else {
    // x3h weight gradient: the GEMM splits both operands in-kernel with one scale per operand, from the maxima the transforms report
    unsigned int* am = nullptr;
    if (q.x6 && ss_x3h_enabled() && (tiles * (q.cin / VW)) % 256 == 0 && (tiles * (q.cout / VW)) % 256 == 0) {
        am = (unsigned int*)((char*)ws + ss_wino_wgrad_ws(q) - 256);
        (void)hipMemsetAsync(am, 0, 8, s);
    }
    hipLaunchKernelGGL((wino_input_kernel<R, false>), dim3(g256(tiles * (q.cin / VW))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                       q.pt, q.pl, q.reflect, V, 0L, nullptr, am);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(wino_dy_kernel<R>, dim3(g256(tiles * (q.cout / VW))), dim3(256), 0, s, dy, q.out_cs, q.n, q.oh, q.ow, q.cout, TH, TW, E,
                       am ? am + 1 : nullptr);
    SS_LAUNCH_CHECK();
    WGradParams p{};
    p.x6 = q.x6;
    p.h_amax = am;
    p.h_amax2 = am ? am + 1 : nullptr;
    p.a = V; p.b = E; p.part = part;
    p.N = 1; p.AH = 1; p.AW = (int)tiles; p.Ca = q.cin; p.a_cs = q.cin;
    p.GH = 1; p.GW = (int)tiles; p.Cb = q.cout; p.b_cs = q.cout;
    p.a_s = 1; p.a_oy = 0; p.a_ox = 0; p.reflect = 0;
    p.ntaps = 1; p.taps[0].dy = 0; p.taps[0].dx = 0; p.taps[0].woff = 0;
    int pps;
    p.splits = ss_wgrad_mfma_splits(tiles, q.cin, q.cout, &pps, XI);
    p.pix_per_split = pps;
    p.nbatch = XI; p.a_bs = tiles * q.cin; p.b_bs = tiles * q.cout;
    int rc = ss_launch_wgrad_mfma_partials(p, s);
    if (rc != SS_OK) return rc;
    launch_wino_dw<R>(part, p.splits, q.cin, q.cout, dw, accumulate, s);
    SS_LAUNCH_CHECK();
    return SS_OK;
    }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------
bool ss_wino_ok(const WinoProb& q) {
    if (!ss_tuning().winograd) return false;
    const int kr = q.cin, no = q.cout;
    return kr % 32 == 0 && no % 4 == 0 && kr >= 64 && no >= 64 && q.in_cs % 4 == 0 && q.out_cs % 4 == 0 &&
           n_tiles(q, 2) >= 1024;
}

// forward pass on fp16 two-piece planes with per-tile scales (fwd_impl's x3h branch): F(4x4,3x3), a tile's channels whole waves
bool ss_wino_fwd_x3h(const WinoProb& q) {
    if (wino_r() != 4 || !q.x6 || q.bf16x3 || !ss_x3h_enabled() || q.cin % 32) return false;
    const int cvi = q.cin / 2;
    return (cvi == 64 || cvi == 128 || cvi == 256) && (n_tiles(q, 4) * cvi) % 256 == 0;
}

// What the forward / weight-gradient pair of this problem passes through ss_conv_desc::saved_operand: the x3h forward's input planes
// (two fp16 planes, rows padded to the GEMM tile) + the per-tile inverse scales.  Needs both halves: the x3h plane forward with
// the scaled low piece (not the opt-in wide tile, whose planes carry the plain low piece) and the pre-split-plane weight gradient.
size_t ss_wino_saved_bytes(const WinoProb& q) {
    if (!ss_tuning().wino_save || wino_r() != 4 || !ss_wino_fwd_x3h(q) || !ss_wino_wgrad_tn(q) || q.fold_h > 0) return 0;
    const long tiles = n_tiles(q, 4);
    if (ss_x6p_wide_ok(tiles, q.cout, q.cin, 36)) return 0;
    const long Mpad = (tiles + SS_X6P_BM - 1) / SS_X6P_BM * SS_X6P_BM;
    return ss_align_up((size_t)2 * 36 * Mpad * q.cin * 2, 256) + ss_align_up((size_t)tiles * 4, 256);
}

size_t ss_wino_fwd_ws(const WinoProb& q) {
    const int R = wino_r(), XI = (R + 2) * (R + 2);
    const long tiles = n_tiles(q, R);
    const size_t planes = ss_align_up((size_t)3 * XI * ss_x6_npad(q.cout) * q.cin * 2, 256);
    const long Mpad = (tiles + SS_X6P_BM - 1) / SS_X6P_BM * SS_X6P_BM;
    const size_t vbytes = ss_align_up((size_t)XI * Mpad * q.cin * 6, 256);      // fp32 V or three bf16 planes with padded rows
    return ss_align_up((size_t)XI * q.cin * q.cout * 4, 256) + vbytes + ss_align_up((size_t)XI * tiles * q.cout * 4, 256) + planes +
           256 + ss_align_up((size_t)tiles * 4, 256);          // x3h: weight scale + per-tile scales
}

// y (+)= act(bias + conv3x3_stride1(x)) with out[o] = sum_a in[map(o + a - pt)] * g[a];  flip = 1: g = rotated + transposed w
int ss_wino_conv_fwd(const WinoProb& q, const float* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, float* y,
                     int act, float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_fwd_ws(q)) return SS_ERR_WORKSPACE;
    if (wino_r() == 2) return fwd_impl<2>(q, x, w, w_cin, w_cout, flip, bias, y, act, alpha, accumulate, ws, s);
    return fwd_impl<4>(q, x, w, w_cin, w_cout, flip, bias, y, act, alpha, accumulate, ws, s);
}

// the same on 16-bit stored activations (dtype = SS_DTYPE_F16 / SS_DTYPE_BF16): x3h-plane shapes only (ss_wino_fwd_x3h), F(4x4,3x3)
int ss_wino_conv_fwd16(const WinoProb& q, int dtype, const void* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, void* y,
                       int act, float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_fwd_ws(q)) return SS_ERR_WORKSPACE;
    if (wino_r() != 4) return SS_ERR_UNSUPPORTED;
    if (dtype == SS_DTYPE_F16) return fwd16_impl<4, _Float16>(q, (const _Float16*)x, w, w_cin, w_cout, flip, bias, (_Float16*)y, act, alpha, accumulate, ws, s);
    if (dtype == SS_DTYPE_BF16) return fwd16_impl<4, __bf16>(q, (const __bf16*)x, w, w_cin, w_cout, flip, bias, (__bf16*)y, act, alpha, accumulate, ws, s);
    return SS_ERR_INVALID;
}

// Chunks per sample of the output statistics the forward output transform can emit: F(4x4,3x3) / F(2x2,3x3) on a plain (not folded)
// output grid whose 256-thread blocks hold whole tiles of one sample
int ss_wino_stats_chunks(const WinoProb& q) {
    if (q.fold_h > 0 || q.bf16x3) return 0;
    const int R = wino_r(), VW = R == 4 ? 2 : 4, CV = q.cout / VW;          // WT<R>::VW channels per thread of the output transform
    if (q.cout % VW || CV > 256 || 256 % CV) return 0;
    const int tps = ((q.oh + R - 1) / R) * ((q.ow + R - 1) / R), tpb = 256 / CV;
    if (tps % tpb) return 0;
    return tps / tpb;
}

// Weight gradient on pre-split K-major planes (gemm_tn_x3h.hip): F(4x4,3x3), x3h arithmetic, shapes the 256 x 128 x 32 tiles divide
bool ss_wino_wgrad_tn(const WinoProb& q) {
    return ss_tuning().wgrad_tn && wino_r() == 4 && q.x6 && !q.bf16x3 && ss_x3h_enabled() && q.cin % 4 == 0 && q.cout % 4 == 0 &&
           ss_gemm_tn_x3h_ok(q.cin, q.cout, n_tiles(q, 4));
}

size_t ss_wino_wgrad_ws(const WinoProb& q) {
    const int R = wino_r(), XI = (R + 2) * (R + 2);
    const long tiles = n_tiles(q, R);
    int pps;
    int splits = ss_wgrad_mfma_splits(tiles, q.cin, q.cout, &pps, XI);
    if (ss_wino_wgrad_tn(q)) { const int s2 = ss_gemm_tn_splits_max(q.cin, q.cout, tiles, XI); if (s2 > splits) splits = s2; }      // (the count itself follows the CU count of the launch: sized for its largest value)
    return ss_align_up((size_t)XI * tiles * q.cin * 4, 256) + ss_align_up((size_t)XI * tiles * q.cout * 4, 256) +
           ss_align_up((size_t)XI * (splits + 1) * q.cin * q.cout * 4, 256) + 256;      // + the x3h amax slot (last 256 bytes)
}

// dw (3,3,cin,cout) (+)= sum_pixels xpad[o + a] * dy[o]
int ss_wino_conv_wgrad(const WinoProb& q, const float* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_wgrad_ws(q)) return SS_ERR_WORKSPACE;
    if (wino_r() == 2) return wgrad_impl<2>(q, x, dy, dw, accumulate, ws, s);
    return wgrad_impl<4>(q, x, dy, dw, accumulate, ws, s);
}

// the same on 16-bit stored x / dy: pre-split-plane path only (ss_wino_wgrad_tn), q.x_amax / q.dy_amax must be set
int ss_wino_conv_wgrad16(const WinoProb& q, int dtype, const void* x, const void* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                         hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_wgrad_ws(q)) return SS_ERR_WORKSPACE;
    if (wino_r() != 4) return SS_ERR_UNSUPPORTED;
    if (dtype == SS_DTYPE_F16) return wgrad_impl<4, _Float16>(q, (const _Float16*)x, (const _Float16*)dy, dw, accumulate, ws, s);
    if (dtype == SS_DTYPE_BF16) return wgrad_impl<4, __bf16>(q, (const __bf16*)x, (const __bf16*)dy, dw, accumulate, ws, s);
    return SS_ERR_INVALID;
}

int ss_wbatch_launch_wino(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(wino_weight_x6_batch_kernel, dim3(nblocks), dim3(256), 0, s, jobs, map);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
