// C-ABI entry points for convolution: maps Conv2D / Conv2DTranspose forward, backward-data and
// backward-weight onto the two gather-GEMM problem forms of common.h.
//
// Conv2DTranspose is handled as the adjoint of a plain strided convolution "C" that goes from the
// transposed conv's OUTPUT space to its INPUT space and shares its kernel memory
// (Keras (kh,kw,cout_T,cin_T) == C's (kh,kw,cin_C,cout_C)):
//     convT.fwd = C.bwd_data,  convT.bwd_data = C.fwd,  convT.bwd_weight = C.bwd_weight with x/dy swapped.
#include "common.h"

namespace {

struct ConvProb {
    int n, ih, iw, cin, in_cs;
    int oh, ow, cout, out_cs;
    int kh, kw, s, pt, pl, reflect;
    // caller-owned x3h slots (ss_conv_desc::x_amax / dy_amax): bit pattern of max|x| / max|dy|; valid = already computed
    unsigned int* x_amax = nullptr;
    unsigned int* dy_amax = nullptr;
    int x_valid = 0, dy_valid = 0;
    int dtype = SS_DTYPE_F32;      // storage type of x / y / dy / dx: 16-bit only ever reaches the tile kernels (see ss_conv2d_fwd)
    WCache* wc = nullptr;          // caller-owned cache of the weight-derived operands of this pass (ss_conv_desc::w_cache)
    float* y_stats = nullptr;      // forward: output statistics for a following norm (ss_conv_desc::y_stats)
    InNorm in_norm;                // forward / weight gradient: x is pre-normalisation, normalised in the operand load (ss_conv_desc::in_norm_*)
    void* saved = nullptr;         // forward -> weight gradient: the transformed input operand (ss_conv_desc::saved_operand)
    int c1_dtype = SS_DTYPE_F32;   // one-channel layers on 16-bit storage: type of the MULTI-channel tensor, read / written by the matrix-core kernels of
                                   // conv_c1.hip themselves (its pointer is reinterpreted); the one-channel tensor is an fp32 staging copy
};

// ---- small helper kernels -----------------------------------------------------------------------
// WT[t][b][a] = W[t][a][b]
__device__ __forceinline__ void transpose_last2_body(const float* __restrict__ w, float* __restrict__ wt, int A, int B, int bx, int by, int bz) {
    __shared__ float tile[32][33];
    const int t = bz;
    const float* src = w + (long)t * A * B;
    float* dst = wt + (long)t * A * B;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int a0 = by * 32, b0 = bx * 32;
    for (int i = ty; i < 32; i += 8) {
        const int a = a0 + i, b = b0 + tx;
        tile[i][tx] = (a < A && b < B) ? src[(long)a * B + b] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int b = b0 + i, a = a0 + tx;
        if (a < A && b < B) dst[(long)b * A + a] = tile[tx][i];
    }
}
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ w, float* __restrict__ wt, int A, int B) {
    transpose_last2_body(w, wt, A, B, blockIdx.x, blockIdx.y, blockIdx.z);
}
// the transposes of a recorded plan in ONE launch (wprep_batch.hip): workgroup -> job through the block map
__global__ __launch_bounds__(256) void transpose_last2_batch_kernel(const SsWJob* __restrict__ jobs, const int* __restrict__ map) {
    const SsWJob& j = jobs[map[blockIdx.x]];
    const int l = blockIdx.x - j.blk0;
    transpose_last2_body(j.src, (float*)j.dst, j.a, j.b, l % j.gx, (l / j.gx) % j.gy, l / (j.gx * j.gy));
}
// max|v| of `n` contiguous, 16-byte aligned floats (a layer's kernel tensor), one job per tensor, `gx` workgroups each: the word was
// zeroed by the plan's first launch
__global__ __launch_bounds__(256) void amax_batch_kernel(const SsWJob* __restrict__ jobs, const int* __restrict__ map) {
    const SsWJob& j = jobs[map[blockIdx.x]];
    const int l = blockIdx.x - j.blk0;
    const float* v = j.src;
    const long n = j.n, n4 = n >> 2;
    unsigned int m = 0;
    for (long i = (long)l * blockDim.x + threadIdx.x; i < n4; i += (long)j.gx * blockDim.x) {
        const f32x4 t = ((const f32x4*)v)[i];
        m = max(max(m, __float_as_uint(fabsf(t[0]))), max(__float_as_uint(fabsf(t[1])), max(__float_as_uint(fabsf(t[2])), __float_as_uint(fabsf(t[3])))));
    }
    if (l == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(fabsf(v[(n4 << 2) + threadIdx.x])));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));
    __shared__ unsigned int wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax((unsigned int*)j.dst, max(max(wm[0], wm[1]), max(wm[2], wm[3])));      // order-independent: deterministic
}

// dx[n,iy,ix,c] (+)= sum over the padded positions that reflect onto (iy,ix) of dpad[n,py,px,c]
// One thread = V (4 or 1) consecutive channels of one dx pixel; 32-bit index arithmetic (the launcher checks the element count).
template <int V>
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx,
                                                           int N, int IH, int IW, int C, int dx_cs,
                                                           int pt, int pl, int PH, int PW, int accumulate) {
    const unsigned CV = (unsigned)C / V;
    const unsigned total = (unsigned)N * IH * IW * CV;
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    unsigned r = e / CV;
    const int c = (int)(e - r * CV) * V;
    const unsigned r2 = r / (unsigned)IW;
    const int ix = (int)(r - r2 * IW);
    const int n = (int)(r2 / (unsigned)IH);
    const int iy = (int)(r2 - (unsigned)n * IH);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = iy + pt;
    if (iy >= 1 && pt - iy >= 0) ys[ny++] = pt - iy;
    { const int py = pt + 2 * (IH - 1) - iy; if (iy <= IH - 2 && py < PH) ys[ny++] = py; }
    xs[nx++] = ix + pl;
    if (ix >= 1 && pl - ix >= 0) xs[nx++] = pl - ix;
    { const int px = pl + 2 * (IW - 1) - ix; if (ix <= IW - 2 && px < PW) xs[nx++] = px; }
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            const float* q = dpad + ((long)(n * PH + ys[a]) * PW + xs[b]) * C + c;
            if (V == 4) {
                const f32x4 t = *(const f32x4*)q;
                acc[0] += t[0]; acc[1 % V] += t[1]; acc[2 % V] += t[2]; acc[3 % V] += t[3];
            } else {
                acc[0] += *q;
            }
        }
    float* o = dx + ((long)(n * IH + iy) * IW + ix) * dx_cs + c;
    if (V == 4) {
        f32x4 t = {acc[0], acc[1 % V], acc[2 % V], acc[3 % V]};
        if (accumulate) t += *(const f32x4*)o;
        *(f32x4*)o = t;
    } else {
        *o = accumulate ? (*o + acc[0]) : acc[0];
    }
}

inline int launch_reflect_fold(const float* dpad, float* dx, int N, int IH, int IW, int C, int dx_cs, int pt, int pl, int PH, int PW,
                               int accumulate, hipStream_t s) {
    const bool v4 = C % 4 == 0 && dx_cs % 4 == 0 && (((uintptr_t)dpad | (uintptr_t)dx) & 15) == 0;
    const long total = (long)N * IH * IW * (C / (v4 ? 4 : 1));
    if (total >= (1L << 32)) return SS_ERR_UNSUPPORTED;      // 32-bit thread index
    if (v4) hipLaunchKernelGGL(reflect_fold_kernel<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpad, dx, N, IH, IW, C, dx_cs, pt, pl, PH, PW, accumulate);
    else hipLaunchKernelGGL(reflect_fold_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpad, dx, N, IH, IW, C, dx_cs, pt, pl, PH, PW, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// column sums of a [rows][C] view: stage 1 partials[chunk][c], stage 2 final
#define COLSUM_CHUNKS 1024
// CT channel lanes (power of two <= 64) x 256/CT row lanes per block: narrow tensors (the 1-channel head bias) still use all threads
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ v, long rows, int C, int cs, float* __restrict__ part, int CT) {
    const int PT = 256 / CT;
    const int cl = threadIdx.x % CT, pl = threadIdx.x / CT;
    const int c = blockIdx.y * CT + cl;
    __shared__ float red[256];
    float acc = 0.f;
    if (c < C) {
        const long per = (rows + gridDim.x - 1) / gridDim.x;
        const long r0 = (long)blockIdx.x * per;
        const long r1 = (r0 + per < rows) ? r0 + per : rows;
        for (long r = r0 + pl; r < r1; r += PT) acc += v[r * cs + c];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = PT / 2; off >= 1; off >>= 1) {          // PT is a power of two: fixed-order tree
        if (pl < off) red[threadIdx.x] += red[threadIdx.x + off * CT];
        __syncthreads();
    }
    if (pl == 0 && c < C) part[(long)blockIdx.x * C + c] = red[threadIdx.x];
}
// one block per 8 channels: 32 chunk lanes per channel walk the partials (fixed assignment k = lane, lane + 32, ...), then a fixed-order
// tree over the lanes -- deterministic, in double.  (One thread per channel walked all 1024 chunks alone: 64 dependent L2 round trips,
// 73 us per launch for the one-channel bias gradient of the generators' head.)
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int chunks, int C, float* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    const int cl = threadIdx.x & 7, kl = threadIdx.x >> 3;          // 8 channels x 32 chunk lanes
    const int c = blockIdx.x * 8 + cl;
    double acc = 0.0;
    if (c < C) {
        for (int k = kl; k < chunks; k += 32 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = k + 32 * u < chunks ? part[(long)(k + 32 * u) * C + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 16; off >= 1; off >>= 1) {
        if (kl < off) red[threadIdx.x] += red[threadIdx.x + off * 8];
        __syncthreads();
    }
    if (kl == 0 && c < C) out[c] = accumulate ? out[c] + (float)red[threadIdx.x] : (float)red[threadIdx.x];
}

int colsum(const float* v, long rows, int C, int cs, float* out, int accumulate, float* part, hipStream_t s) {
    int CT = 1;
    while (CT < C && CT < 64) CT <<= 1;
    int chunks = (int)((rows + 1023) / 1024);
    if (chunks > COLSUM_CHUNKS) chunks = COLSUM_CHUNKS;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(colsum_stage1, dim3(chunks, (C + CT - 1) / CT), dim3(256), 0, s, v, rows, C, cs, part, CT);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 7) / 8), dim3(256), 0, s, part, chunks, C, out, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// ---- single-output-channel restructuring ---------------------------------------------------------
// A conv with ONE output channel (7x7 64->1 generator head, 4x4 512->1 PatchGAN head and the backward-data of
// the Cin=1 stems) is a GEMV per pixel: not matrix-core shaped.  It is re-associated into
//   T[q][t]  = sum_ci in[q][ci] * w[t][ci]            (1x1 conv, N = #taps: an MFMA GEMM, same FLOPs)
//   out[p]   = act(bias + sum_t T[map(p*s + off + tap_t)][t])      (streaming "tap sum")
// and its weight gradient into the adjoint
//   U[q][t]  = sum_{g : map(g*s + off + tap_t) == q} dy[g]          ("tap scatter", one channel)
//   dw[t][ci] = sum_q U[q][t] * in[q][ci]                           (MFMA GEMM over pixels)

// wt2[ci][t] = w[woff_t + ci*ldb]  (t < ntaps), zero padded to tcs columns
__global__ __launch_bounds__(256) void gather_w_kernel(GConvParams p, float* __restrict__ wt2, int tcs) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.Cin * tcs) return;
    const int t = e % tcs, ci = e / tcs;
    wt2[e] = (t < p.ntaps) ? p.w[p.taps[t].woff + (long)ci * p.ldb] : 0.f;
}

// p.in = T with channel stride p.in_cs; one thread per output pixel
__global__ __launch_bounds__(256) void tapsum_kernel(GConvParams p) {
    const long total = (long)p.N * p.OHc * p.OWc;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const int xc = (int)(pix % p.OWc);
    const long r = pix / p.OWc;
    const int yc = (int)(r % p.OHc);
    const int n = (int)(r / p.OHc);
    const int oy = yc * p.out_s + p.out_oy, ox = xc * p.out_s + p.out_ox;
    if (oy < 0 || oy >= p.OH || ox < 0 || ox >= p.OW) return;
    const int by = yc * p.in_s + p.in_oy, bx = xc * p.in_s + p.in_ox;
    float acc = 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
        const int iy = ss_map_index(by + p.taps[t].dy, p.IH, p.reflect);
        const int ix = ss_map_index(bx + p.taps[t].dx, p.IW, p.reflect);
        if (iy >= 0 && ix >= 0) acc += p.in[((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs + t];
    }
    float* op = p.out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs;
    float v = ss_apply_act(acc + (p.bias ? p.bias[0] : 0.f), p.act, p.alpha);
    if (p.accumulate) v += *op;
    *op = v;
}

// U[n,qy,qx,t] = sum over grid positions g whose tap t reads input position q (through the pad map) of b[g]
// One thread = 4 consecutive taps of one pixel (one 16-byte store); 32-bit index arithmetic (the launcher checks the range):
// the per-element 64-bit div/mod chain of the first version made this streaming kernel VALU-bound (667 us for 436 MB).
__global__ __launch_bounds__(256) void tapscatter_kernel(WGradParams p, float* __restrict__ U, int tcs) {
    const unsigned T4 = (unsigned)tcs / 4;
    const unsigned total = (unsigned)p.N * p.AH * p.AW * T4;
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    unsigned r = e / T4;
    const int t0 = (int)(e - r * T4) * 4;
    const unsigned r2 = r / (unsigned)p.AW;
    const int qx = (int)(r - r2 * p.AW);
    const int n = (int)(r2 / (unsigned)p.AH);
    const int qy = (int)(r2 - (unsigned)n * p.AH);
    int cy[3], cx[3], ny = 0, nx = 0;
    cy[ny++] = qy;
    cx[nx++] = qx;
    if (p.reflect) {
        if (qy >= 1) cy[ny++] = -qy;
        if (qy <= p.AH - 2) cy[ny++] = 2 * (p.AH - 1) - qy;
        if (qx >= 1) cx[nx++] = -qx;
        if (qx <= p.AW - 2) cx[nx++] = 2 * (p.AW - 1) - qx;
    }
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + k;
        if (t >= p.ntaps) continue;
        float acc = 0.f;
        for (int a = 0; a < ny; ++a) {
            const int ty = cy[a] - p.a_oy - p.taps[t].dy;
            if (ty < 0) continue;
            int gy = ty;
            if (p.a_s != 1) { if (ty % p.a_s) continue; gy = ty / p.a_s; }
            if (gy >= p.GH) continue;
            for (int b = 0; b < nx; ++b) {
                const int tx = cx[b] - p.a_ox - p.taps[t].dx;
                if (tx < 0) continue;
                int gx = tx;
                if (p.a_s != 1) { if (tx % p.a_s) continue; gx = tx / p.a_s; }
                if (gx >= p.GW) continue;
                acc += p.b[((long)(n * p.GH + gy) * p.GW + gx) * p.b_cs];
            }
        }
        out[k] = acc;
    }
    *(f32x4*)(U + (long)e * 4) = out;
}

inline int round4(int v) { return (v + 3) / 4 * 4; }

// ---- problem builders ---------------------------------------------------------------------------
bool use_mfma(int algo, const GConvParams& p) {
    if (algo == SS_ALGO_DIRECT) return false;
    if (algo == SS_ALGO_MFMA) return true;
    return ss_gconv_mfma_ok(p);
}

bool gconv_two_stage(int algo, const GConvParams& p) {
    return algo != SS_ALGO_DIRECT && p.Cout == 1 && p.ntaps >= 4 && p.Cin >= 16;
}

// upper bound of the two-stage scratch for a gather conv with `cred` reduction channels over an n x ih x iw input
size_t two_stage_ws(int n, int ih, int iw, int cred, int cout, int ntaps) {
    if (cout != 1 || ntaps < 4 || cred < 16) return 256;
    const int tcs = round4(ntaps);
    return ss_align_up((size_t)cred * tcs * sizeof(float), 256) + ss_align_up((size_t)n * ih * iw * tcs * sizeof(float), 256);
}

// x6 (bf16 matrix cores, fp32-exact operands): AUTO (unless SS_X6=0) and SS_ALGO_X6, shapes with Cin % 32 == 0
bool x6_wanted(int algo) { return algo == SS_ALGO_X6 || (algo == SS_ALGO_AUTO && ss_tuning().x6); }
bool use_x6(int algo, const GConvParams& p) {
    return x6_wanted(algo) && ss_gconv_x6_ok(p) && (long)p.N * p.OHc * p.OWc >= 1024;
}
// upper bound of the weight-plane scratch of a gather conv with `cred` reduction channels, `cout` outputs, `ntaps` taps
size_t x6_planes_ub(int cred, int cout, int ntaps) {
    if (cred < 16 || cout < 32) return 0;
    return ss_align_up((size_t)3 * ss_x6_npad(cout) * ntaps * ((cred + 31) / 32 * 32) * sizeof(unsigned short), 256) + 256;   // + the x3h amax slot
}

// ---- x3h for the direct (non-Winograd) x6 convolutions: one power-of-two scale per operand TENSOR, from max|.| ----------------------
// out[0] = max over the [rows][C] view (row stride cs) of |v|, as a bit pattern (non-negative floats order like unsigned ints; an
// atomic max does not depend on the order: deterministic).  out must be zero on entry.
__global__ __launch_bounds__(256) void amax_view_kernel(const float* __restrict__ v, long rows, int C, int cs, unsigned int* __restrict__ out, int stripes) {
    unsigned int m = 0;
    const bool al = (((uintptr_t)v) & 15) == 0;
    if (cs == C && al) {                     // dense: one flat array, no index arithmetic
        const long n = rows * C, n4 = n >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const f32x4 t = ((const f32x4*)v)[i];
            m = max(max(m, __float_as_uint(fabsf(t[0]))), max(__float_as_uint(fabsf(t[1])), max(__float_as_uint(fabsf(t[2])), __float_as_uint(fabsf(t[3])))));
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(fabsf(v[(n4 << 2) + threadIdx.x])));
    } else if (C % 4 == 0 && cs % 4 == 0 && al) {
        const int C4 = C / 4;
        const long total = rows * C4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / C4;
            const int c = (int)(i - r * C4) * 4;
            const f32x4 t = *(const f32x4*)(v + r * cs + c);
            m = max(max(m, __float_as_uint(fabsf(t[0]))), max(__float_as_uint(fabsf(t[1])), max(__float_as_uint(fabsf(t[2])), __float_as_uint(fabsf(t[3])))));
        }
    } else {
        const long total = rows * C;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / C;
            m = max(m, __float_as_uint(fabsf(v[r * cs + (i - r * C)])));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));
    __shared__ unsigned int wm[4];           // ONE atomic per block: thousands of atomics on one address serialise in L2 (10 ns each)
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + (stripes > 1 ? (blockIdx.x % stripes) * SS_AMAX_STRIDE : 0), max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}

// the same over a 16-bit stored view (generic strided walk; the maximum of the stored values, as fp32 bits)
template <typename T>
__global__ __launch_bounds__(256) void amax_view16_kernel(const T* __restrict__ v, long rows, int C, int cs, unsigned int* __restrict__ out, int stripes) {
    unsigned int m = 0;
    const int C2 = C / 2;
    if (C % 2 == 0 && cs % 2 == 0 && ((uintptr_t)v & 3) == 0) {
        typedef T T2 __attribute__((ext_vector_type(2)));
        const long total = rows * C2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / C2;
            const T2 t = *(const T2*)(v + r * cs + (i - r * C2) * 2);
            m = max(m, max(__float_as_uint(fabsf((float)t[0])), __float_as_uint(fabsf((float)t[1]))));
        }
    } else {
        const long total = rows * C;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            const long r = i / C;
            m = max(m, __float_as_uint(fabsf((float)v[r * cs + (i - r * C)])));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));
    __shared__ unsigned int wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + (stripes > 1 ? (blockIdx.x % stripes) * SS_AMAX_STRIDE : 0), max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}

inline void launch_amax_view(const float* v, long rows, int C, int cs, unsigned int* out, int stripes, hipStream_t s) {
    const long work = rows * C / 4;
    const unsigned nb = (unsigned)(work / 4096 < 32 ? 32 : (work / 4096 > 1024 ? 1024 : work / 4096));
    hipLaunchKernelGGL(amax_view_kernel, dim3(nb), dim3(256), 0, s, v, rows, C, cs, out, stripes);
}
}  // namespace
void ss_launch_amax_view(const float* v, long rows, int C, int cs, unsigned int* out, int stripes, hipStream_t s) { launch_amax_view(v, rows, C, cs, out, stripes, s); }
int ss_wbatch_launch_amax(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(amax_batch_kernel, dim3(nblocks), dim3(256), 0, s, jobs, map);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
int ss_wbatch_launch_transpose(const SsWJob* jobs, const int* map, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(transpose_last2_batch_kernel, dim3(nblocks), dim3(256), 0, s, jobs, map);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
namespace {
struct AmaxRef { const unsigned int* p; int stripes; };
// maximum of an activation / gradient view for its x3h scale: into the caller's slot when there is one (computed only if the
// caller does not vouch for its contents), else into `scratch`; returns the slot the kernels read
// A caller's slot is striped (SS_AMAX_STRIPES words, one per 256-byte line), the scratch word is not.
AmaxRef act_amax(const float* v, long rows, int C, int cs, unsigned int* ext, int ext_valid, unsigned int* scratch, hipStream_t s) {
    unsigned int* slot = ext ? ext : scratch;
    const int stripes = ext ? SS_AMAX_STRIPES : 1;
    if (!(ext && ext_valid == 1)) {
        // (valid == 2: the caller vouches that its slot is ZERO -- fresh from a zeroed pool: one memset dispatch less per scan)
        if (!(ext && ext_valid == 2)) (void)hipMemsetAsync(slot, 0, ext ? (size_t)SS_AMAX_STRIPES * SS_AMAX_STRIDE * 4 : 4, s);
        launch_amax_view(v, rows, C, cs, slot, stripes, s);
    }
    return AmaxRef{slot, stripes};
}
// ... of a 16-bit stored view
AmaxRef act_amax16(const void* v, int dtype, long rows, int C, int cs, unsigned int* ext, int ext_valid, unsigned int* scratch, hipStream_t s) {
    unsigned int* slot = ext ? ext : scratch;
    const int stripes = ext ? SS_AMAX_STRIPES : 1;
    if (!(ext && ext_valid == 1)) {
        if (!(ext && ext_valid == 2)) (void)hipMemsetAsync(slot, 0, ext ? (size_t)SS_AMAX_STRIPES * SS_AMAX_STRIDE * 4 : 4, s);
        const long work = rows * C / 4;
        const unsigned nb = (unsigned)(work / 4096 < 32 ? 32 : (work / 4096 > 1024 ? 1024 : work / 4096));
        if (dtype == SS_DTYPE_F16) hipLaunchKernelGGL(amax_view16_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16*)v, rows, C, cs, slot, stripes);
        else hipLaunchKernelGGL(amax_view16_kernel<__bf16>, dim3(nb), dim3(256), 0, s, (const __bf16*)v, rows, C, cs, slot, stripes);
    }
    return AmaxRef{slot, stripes};
}
const unsigned int* weight_amax(const float* w, long wn, unsigned int* scratch, hipStream_t s, WCache* wc = nullptr) {
    bool fill;
    scratch = (unsigned int*)ss_wc_region(wc, ss_wc_tag(SS_WC_WAMAX, 0), 256, scratch, &fill);
    if (!fill) return scratch;
    const unsigned nb = wn / 8192 < 16 ? 16 : (wn / 8192 > 256 ? 256 : (unsigned)(wn / 8192));
    if (ss_wrec_on() && (((uintptr_t)w) & 15) == 0) {          // recorded (ss_wprep_*): the plan zeroes the word and scans the tensor
        SsWJob j{};
        j.type = SS_WJ_AMAX; j.gx = (int)(wn / 16384 < 4 ? 4 : (wn / 16384 > 64 ? 64 : wn / 16384)); j.gy = 1; j.gz = 1;
        j.src = w; j.n = wn; j.dst = scratch;
        ss_wrec_push(j);
        return scratch;
    }
    if (ss_wrec_on()) ss_wrec_unbatched();
    (void)hipMemsetAsync(scratch, 0, 4, s);
    hipLaunchKernelGGL(amax_view_kernel, dim3(nb), dim3(256), 0, s, w, 1L, (int)wn, (int)wn, scratch, 1);
    return scratch;
}
bool x3h_direct_wanted(int algo, int cred, int cout) {
    return ss_tuning().x3h_direct && ss_x3h_enabled() && x6_wanted(algo) && cred >= 16 && cout >= 32;
}
// Which tensor maxima a pass computes / reads.  The passes FILL the slots exactly when these say so (whatever kernel ends up
// running), so ss_conv2d_uses_amax can promise the caller that a slot is valid afterwards.
bool wino_fwd_prob(const ConvProb& c, int algo, WinoProb* q);
bool wino_dgrad_prob(const ConvProb& c, int algo, WinoProb* q);
bool need_x_amax_fwd(const ConvProb& c, int algo);
bool need_dy_amax_dgrad(const ConvProb& c, int algo);
bool need_amax_wgrad(const ConvProb& c, int algo);

// LDS-staged tile kernel (conv_tile.hip): small-channel stride-1 problems on large maps, after the one-channel kernels
bool tconv_takes(int algo, const GConvParams& p) {
    return (algo == SS_ALGO_AUTO || algo == SS_ALGO_X6) && !ss_conv_out1_ok(p) && !ss_conv_in1_ok(p) && ss_tconv_ok(p);
}
// upper bound of its weight-plane scratch for `cred` reduction channels, `cout` outputs, `ntaps` taps
size_t tconv_ws_ub(int cred, int cout, int ntaps) {
    const int nq = ntaps * (((cred + 7) / 8 + 1) / 2) * 2, nb = (cout + 31) / 32;
    return 256 + (size_t)2 * nb * 32 * nq * 16;
}

size_t gconv_ws_bytes(int algo, const GConvParams& p) {
    if (!gconv_two_stage(algo, p)) {
        const size_t a = 256 + x6_planes_ub(p.Cin, p.Cout, p.ntaps), b = tconv_ws_ub(p.Cin, p.Cout, p.ntaps);
        return a > b ? a : b;
    }
    return two_stage_ws(p.N, p.IH, p.IW, p.Cin, p.Cout, p.ntaps);
}

// identity of the weight planes wprep_x6 derives for `p` (within one layer: which taps, in which order, from which source array)
uint64_t x6_planes_detail(const GConvParams& p, int wsrc) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix((uint64_t)wsrc); mix((uint64_t)p.ntaps); mix((uint64_t)p.Cin); mix((uint64_t)p.Cout); mix((uint64_t)p.ldb); mix(p.h_amax ? 1 : 0);
    mix((uint64_t)(p.nbatch > 1 ? p.nbatch : 1)); mix((uint64_t)p.w_bs);
    for (int i = 0; i < p.ntaps; ++i) mix((uint64_t)(uint32_t)p.taps[i].woff);
    return h >> 8;
}

// wsrc: 0 = p.w is the layer's weight array, 1 = its tap-wise transpose (conv_bwd_data) -- part of the cache identity of the planes
int run_gconv(int algo, const GConvParams& p, void* ws, size_t ws_bytes, hipStream_t s, WCache* wc = nullptr, int wsrc = 0) {
    if (wc && wc->fill_only) {            // refresh of cached operands: only the split-plane path keeps any
        const bool c1 = algo != SS_ALGO_DIRECT && algo != SS_ALGO_MFMA && (ss_conv_out1_ok(p) || ss_conv_in1_ok(p));
        if (c1 || (tconv_takes(algo, p) && ws && ws_bytes >= ss_tconv_ws(p)) || gconv_two_stage(algo, p)) return SS_OK;
        if (!(use_x6(algo, p) && ws && ws_bytes >= ss_gconv_x6_planes_bytes(p))) return SS_OK;
    }
    if (p.dtype != SS_DTYPE_F32) {          // 16-bit storage: only the x3h gather kernels load / store the stored type here (callers check with gconv16_takes)
        const bool v2 = (algo == SS_ALGO_AUTO || algo == SS_ALGO_X6) && !ss_conv_out1_ok(p) && !ss_conv_in1_ok(p) && !tconv_takes(algo, p) &&
                        !gconv_two_stage(algo, p) && use_x6(algo, p) && (ss_gconv_x6v2_ok(p) || ss_gconv_x6_typed_ok(p)) && ws &&
                        ws_bytes >= ss_gconv_x6_planes_bytes(p);
        if (!v2) { ss_set_error("run_gconv: no kernel with a 16-bit loader for this problem"); return SS_ERR_UNSUPPORTED; }
    }
    if (algo != SS_ALGO_DIRECT && algo != SS_ALGO_MFMA) {        // full-resolution 7x7 stem / head shapes: LDS-tiled VALU kernels
        if (ss_conv_out1_ok(p)) return ss_launch_conv_out1(p, s);
        if (ss_conv_in1_ok(p)) return ss_launch_conv_in1(p, s);
    }
    if (p.c1_dtype != SS_DTYPE_F32) { ss_set_error("run_gconv: c1_dtype set on a problem the one-channel kernels do not take"); return SS_ERR_UNSUPPORTED; }
    if (p.stats && !(use_x6(algo, p) && ss_gconv_x6v2_ok(p) && ws && ws_bytes >= ss_gconv_x6_planes_bytes(p))) {
        ss_set_error("conv2d_fwd: y_stats was promised (ss_conv2d_stats_chunks) but this launch cannot take the kernel that writes it (workspace / operand alignment)");
        return SS_ERR_UNSUPPORTED;
    }
    if (tconv_takes(algo, p) && ws && ws_bytes >= ss_tconv_ws(p)) return ss_launch_tconv(p, ws, ws_bytes, s);
    if (gconv_two_stage(algo, p)) {
        if (!ws || ws_bytes < gconv_ws_bytes(algo, p)) return SS_ERR_WORKSPACE;
        const int tcs = round4(p.ntaps);
        float* wt2 = (float*)ws;
        float* T = (float*)((char*)ws + ss_align_up((size_t)p.Cin * tcs * sizeof(float), 256));
        hipLaunchKernelGGL(gather_w_kernel, dim3((p.Cin * tcs + 255) / 256), dim3(256), 0, s, p, wt2, tcs);
        SS_LAUNCH_CHECK();
        GConvParams q{};
        q.in = p.in; q.w = wt2; q.bias = nullptr; q.out = T;
        q.N = p.N; q.IH = p.IH; q.IW = p.IW; q.Cin = p.Cin; q.in_cs = p.in_cs;
        q.OHc = p.IH; q.OWc = p.IW; q.in_s = 1; q.in_oy = 0; q.in_ox = 0;
        q.OH = p.IH; q.OW = p.IW; q.Cout = tcs; q.out_cs = tcs; q.out_s = 1; q.out_oy = 0; q.out_ox = 0;
        q.ldb = tcs; q.reflect = 0; q.act = SS_ACT_NONE; q.alpha = 0.f; q.accumulate = 0;
        q.ntaps = 1; q.taps[0].dy = 0; q.taps[0].dx = 0; q.taps[0].woff = 0;
        int rc = ss_launch_gconv_mfma(q, s);
        if (rc != SS_OK) return rc;
        GConvParams r = p;
        r.in = T; r.in_cs = tcs;
        const long total = (long)p.N * p.OHc * p.OWc;
        hipLaunchKernelGGL(tapsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    if (use_x6(algo, p) && ws && ws_bytes >= ss_gconv_x6_planes_bytes(p)) {
        bool fill;
        unsigned short* planes = (unsigned short*)ss_wc_region(wc, ss_wc_tag(SS_WC_X6_PLANES, x6_planes_detail(p, wsrc)), ss_gconv_x6_planes_bytes(p), ws, &fill);
        if (fill) {
            int rc = ss_launch_wprep_x6(p, planes, s);
            if (rc != SS_OK) return rc;
        }
        if (wc && wc->fill_only) return SS_OK;
        return ss_launch_gconv_x6(p, planes, s);
    }
    return use_mfma(algo, p) ? ss_launch_gconv_mfma(p, s) : ss_launch_gconv_direct(p, s);
}

// the sub-pixel phases of one data gradient (conv_bwd_data): one launch when all of them take the x3h gather kernels and line up
int run_gconv_phases(int algo, const GConvParams* ps, int count, void* ws, size_t ws_bytes, hipStream_t s, WCache* wc) {
    bool multi = count >= 2 && algo != SS_ALGO_DIRECT && algo != SS_ALGO_MFMA && ws;
    const bool fused = multi && ss_gconv_phases_fused_ok(ps, count);          // (takes class grids that differ by one between the phases)
    size_t need = 0;
    for (int i = 0; i < count && multi; ++i) {
        const GConvParams& q = ps[i];
        if (ss_conv_out1_ok(q) || ss_conv_in1_ok(q) || tconv_takes(algo, q) || gconv_two_stage(algo, q) || !use_x6(algo, q) || q.stats ||
            q.ntaps > SS_MAX_PHASE_TAPS || !q.h_amax || !q.h_amax2 || (!fused && (q.OHc != ps[0].OHc || q.OWc != ps[0].OWc)))
            multi = false;
        if (q.dtype != SS_DTYPE_F32 && !(ss_gconv_x6v2_ok(q) || ss_gconv_x6_typed_ok(q))) multi = false;
        need += ss_gconv_x6_planes_bytes(q);
    }
    if (fused) {          // one workgroup per input tile for all phases: ONE set of weight planes over the phases' taps
        GConvParams wq;
        if (ss_gconv_phases_fused_wprob(ps, count, &wq) && ss_gconv_x6_planes_bytes(wq) <= ws_bytes) {
            bool fill;
            unsigned short* pl = (unsigned short*)ss_wc_region(wc, ss_wc_tag(SS_WC_X6_PLANES, x6_planes_detail(wq, 2)), ss_gconv_x6_planes_bytes(wq), ws, &fill);
            if (fill) {
                const int rc = ss_launch_wprep_x6(wq, pl, s);
                if (rc != SS_OK) return rc;
            }
            if (wc && wc->fill_only) return SS_OK;
            return ss_launch_gconv_phases_fused(ps, pl, count, s);
        }
    }
    if (multi && need <= ws_bytes && ss_tuning().gconv_phases) {          // (the joint launch of the per-phase kernels is opt-in)
        const unsigned short* planes[SS_MAX_PHASES];
        char* wp = (char*)ws;
        for (int i = 0; i < count; ++i) {
            bool fill;
            unsigned short* pl = (unsigned short*)ss_wc_region(wc, ss_wc_tag(SS_WC_X6_PLANES, x6_planes_detail(ps[i], 1)), ss_gconv_x6_planes_bytes(ps[i]), wp, &fill);
            if (fill) {
                const int rc = ss_launch_wprep_x6(ps[i], pl, s);
                if (rc != SS_OK) return rc;
            }
            planes[i] = pl;
            wp += ss_gconv_x6_planes_bytes(ps[i]);
        }
        if (wc && wc->fill_only) return SS_OK;
        const int rc = ss_launch_gconv_x6_multi(ps, planes, count, s);
        if (rc != SS_ERR_UNSUPPORTED) return rc;          // launched (or failed for real); UNSUPPORTED: nothing launched, planes stay valid in the cache / workspace
        if (!wc) {          // the workspace copies are laid out for the joint launch: per-phase launches refill their own
            for (int i = 0; i < count; ++i) {
                const int rc1 = run_gconv(algo, ps[i], ws, ws_bytes, s, wc, 1);
                if (rc1 != SS_OK) return rc1;
            }
            return SS_OK;
        }
    }
    for (int i = 0; i < count; ++i) {
        const int rc = run_gconv(algo, ps[i], ws, ws_bytes, s, wc, 1);
        if (rc != SS_OK) return rc;
    }
    return SS_OK;
}

GConvParams fwd_params(const ConvProb& c, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
                       int accumulate) {
    GConvParams p{};
    p.in = x; p.w = w; p.bias = bias; p.out = y;
    p.N = c.n; p.IH = c.ih; p.IW = c.iw; p.Cin = c.cin; p.in_cs = c.in_cs;
    p.OHc = c.oh; p.OWc = c.ow;
    p.in_s = c.s; p.in_oy = -c.pt; p.in_ox = -c.pl;
    p.OH = c.oh; p.OW = c.ow; p.Cout = c.cout; p.out_cs = c.out_cs;
    p.out_s = 1; p.out_oy = 0; p.out_ox = 0;
    p.ldb = c.cout; p.reflect = c.reflect; p.act = act; p.alpha = alpha; p.accumulate = accumulate;
    p.dtype = c.dtype;
    p.c1_dtype = c.c1_dtype;
    p.ntaps = 0;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = p.taps[p.ntaps++];
            t.dy = (int16_t)a; t.dx = (int16_t)b; t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    return p;
}

// Winograd F(2x2,3x3) eligibility of the plain conv `c` (forward / weight-gradient view)
bool wino_fwd_prob(const ConvProb& c, int algo, WinoProb* q) {
    if (algo == SS_ALGO_DIRECT || c.kh != 3 || c.kw != 3 || c.s != 1) return false;
    *q = WinoProb{c.n, c.ih, c.iw, c.cin, c.in_cs, c.oh, c.ow, c.cout, c.out_cs, c.pt, c.pl, c.reflect, algo == SS_ALGO_BF16X3, x6_wanted(algo), 0, 0};
    return ss_wino_ok(*q);
}
// reflect-pad(1) + 3x3 valid conv whose input dims are multiples of the F(4x4,3x3) tile: fold inside the output transform
bool wino_fold_ok(const ConvProb& c) {
    return c.reflect && c.pt == 1 && c.pl == 1 && c.oh == c.ih && c.ow == c.iw && c.ih % 4 == 0 && c.iw % 4 == 0 && c.ih >= 4 &&
           c.iw >= 4 && ss_tuning().wino_r == 4;
}
// backward-data view: gathers dy (zero extension), produces dx (zero padding) or the padded gradient (reflect)
bool wino_dgrad_prob(const ConvProb& c, int algo, WinoProb* q) {
    if (algo == SS_ALGO_DIRECT || c.kh != 3 || c.kw != 3 || c.s != 1) return false;
    const int bf = algo == SS_ALGO_BF16X3;
    const int x6 = x6_wanted(algo);
    if (c.reflect && wino_fold_ok(c))       // padded gradient shifted by one, folded by the output transform straight into dx
        *q = WinoProb{c.n, c.oh, c.ow, c.cout, c.out_cs, c.oh + 4, c.ow + 4, c.cin, c.in_cs, 3, 3, 0, bf, x6, c.ih, c.iw};
    else if (c.reflect) *q = WinoProb{c.n, c.oh, c.ow, c.cout, c.out_cs, c.oh + 2, c.ow + 2, c.cin, c.cin, 2, 2, 0, bf, x6, 0, 0};
    else *q = WinoProb{c.n, c.oh, c.ow, c.cout, c.out_cs, c.ih, c.iw, c.cin, c.in_cs, 2 - c.pt, 2 - c.pl, 0, bf, x6, 0, 0};
    return ss_wino_ok(*q);
}

// Chunks per sample of output statistics the NON-Winograd forward of `c` emits (ss_conv_desc::y_stats): the kernel run_gconv will
// pick must be one that writes them -- the 1 -> C matrix-core kernel (conv_c1.hip) or gconv_x6v2 on the x3h path.  Shape-only: the
// same answer with and without pointers (operand alignment is the caller's promise, checked again at launch).
int gconv_stats_chunks(const ConvProb& c, int algo) {
    if (c.kh * c.kw > SS_MAX_TAPS || (algo != SS_ALGO_AUTO && algo != SS_ALGO_X6) || c.in_norm.groups > 0) return 0;
    GConvParams p = fwd_params(c, nullptr, nullptr, nullptr, nullptr, SS_ACT_NONE, 0.f, 0);
    if (ss_conv_out1_ok(p)) return 0;
    if (ss_conv_in1_ok(p)) return ss_conv_in1_stats_chunks(p);
    if (tconv_takes(algo, p) || gconv_two_stage(algo, p) || !need_x_amax_fwd(c, algo)) return 0;
    static const unsigned int dummy = 0;
    p.h_amax = &dummy; p.h_amax2 = &dummy; p.amax_stripes = 1;
    if (!use_x6(algo, p) || !ss_gconv_x6v2_ok(p)) return 0;
    return ss_gconv_x6v2_stats_chunks(p);
}

size_t fwd_ws(const ConvProb& c, int algo) {
    if (c.kh * c.kw > SS_MAX_TAPS) return 0;
    WinoProb q;
    if (wino_fwd_prob(c, algo, &q)) return ss_wino_fwd_ws(q);
    return gconv_ws_bytes(algo, fwd_params(c, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0));
}

int conv_fwd(const ConvProb& c, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
             int accumulate, int algo, void* ws, size_t ws_bytes, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    WinoProb q;
    if (wino_fwd_prob(c, algo, &q)) {
        q.wc = c.wc;
        q.y_stats = c.y_stats;
        q.in_norm = c.in_norm;
        q.saved = c.saved;
        if (c.in_norm.groups > 0 && c.x_amax && c.x_valid != 1 && !(c.wc && c.wc->fill_only)) {      // the transform reports max|normalised x|
            if (c.x_valid != 2) (void)hipMemsetAsync(c.x_amax, 0, (size_t)SS_AMAX_STRIPES * SS_AMAX_STRIDE * 4, s);
            q.in_norm.amax_out = c.x_amax;
        }
        return ss_wino_conv_fwd(q, x, w, c.cin, c.cout, 0, bias, y, act, alpha, accumulate, ws, ws_bytes, s);
    }
    if (c.in_norm.groups > 0) { ss_set_error("in_norm: this forward pass does not normalise in its operand load (ss_conv2d_fuses_in_norm)"); return SS_ERR_UNSUPPORTED; }
    GConvParams p = fwd_params(c, x, w, bias, y, act, alpha, accumulate);
    if (c.y_stats && act == SS_ACT_NONE && !accumulate) {
        p.stats_chunks = gconv_stats_chunks(c, algo);
        if (p.stats_chunks > 0) p.stats = c.y_stats;
    }
    if (c.c1_dtype == SS_DTYPE_F32 && need_x_amax_fwd(c, algo) && ws && ws_bytes >= ss_gconv_x6_planes_bytes(p) + 256) {
        unsigned int* sl = (unsigned int*)((char*)ws + ss_gconv_x6_planes_bytes(p));
        const bool fill_only = c.wc && c.wc->fill_only;
        const AmaxRef ax = fill_only ? AmaxRef{sl, 1} : act_amax(x, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
        if (use_x6(algo, p)) {
            p.h_amax = ax.p; p.amax_stripes = ax.stripes;
            p.h_amax2 = weight_amax(w, (long)c.kh * c.kw * c.cin * c.cout, sl + 1, s, c.wc);
        }
    }
    return run_gconv(algo, p, ws, ws_bytes, s, c.wc);
}

size_t bwd_data_wt_bytes(const ConvProb& c) { return ss_align_up((size_t)c.kh * c.kw * c.cin * c.cout * sizeof(float), 256); }
size_t bwd_data_dpad_bytes(const ConvProb& c) {
    if (!c.reflect) return 0;
    const int PH = c.oh + c.kh - 1, PW = c.ow + c.kw - 1;
    return ss_align_up((size_t)c.n * PH * PW * c.cin * sizeof(float), 256);
}
size_t bwd_data_ws(const ConvProb& c) {
    // [transposed weights][padded gradient (reflect)][two-stage scratch of the gather conv (Cin == 1 stems) | Winograd scratch]
    size_t extra = two_stage_ws(c.n, c.oh, c.ow, c.cout, c.cin, c.kh * c.kw);
    { const size_t xq = x6_planes_ub(c.cout, c.cin, c.kh * c.kw); if (xq > extra) extra = xq; }
    { const size_t tq = tconv_ws_ub(c.cout, c.cin, c.kh * c.kw); if (tq > extra) extra = tq; }
    WinoProb q;
    if (wino_dgrad_prob(c, SS_ALGO_AUTO, &q)) { const size_t wq = ss_wino_fwd_ws(q); if (wq > extra) extra = wq; }
    return bwd_data_wt_bytes(c) + bwd_data_dpad_bytes(c) + extra;
}

// dx = dC/dx for the plain conv `c`; bias/act only used when this implements a transposed-conv forward
int conv_bwd_data(const ConvProb& c, const float* dy, const float* w, float* dx, const float* bias, int act, float alpha,
                  int accumulate, int algo, void* ws, size_t ws_bytes, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    if (c.reflect && c.s != 1) return SS_ERR_UNSUPPORTED;
    if (ws_bytes < bwd_data_ws(c) || !ws) return SS_ERR_WORKSPACE;
    float* wt = (float*)ws;
    const int T = c.kh * c.kw;
    void* gws = (char*)ws + bwd_data_wt_bytes(c) + bwd_data_dpad_bytes(c);
    const size_t gws_bytes = ws_bytes - bwd_data_wt_bytes(c) - bwd_data_dpad_bytes(c);
    {
        WinoProb q;
        if (wino_dgrad_prob(c, algo, &q)) {      // rotated + transposed weights are formed inside the weight transform
            q.wc = c.wc;
            if (!c.reflect || q.fold_h > 0)
                return ss_wino_conv_fwd(q, dy, w, c.cin, c.cout, 1, bias, dx, act, alpha, accumulate, gws, gws_bytes, s);
            float* dpad = (float*)((char*)ws + bwd_data_wt_bytes(c));
            int rc = ss_wino_conv_fwd(q, dy, w, c.cin, c.cout, 1, nullptr, dpad, SS_ACT_NONE, 0.f, 0, gws, gws_bytes, s);
            if (rc != SS_OK || (c.wc && c.wc->fill_only)) return rc;
            return launch_reflect_fold(dpad, dx, c.n, c.ih, c.iw, c.cin, c.in_cs, c.pt, c.pl, c.oh + 2, c.ow + 2, accumulate, s);
        }
    }
    {
        bool fill;
        wt = (float*)ss_wc_region(c.wc, ss_wc_tag(SS_WC_WT, 0), bwd_data_wt_bytes(c), wt, &fill);
        if (fill && ss_wrec_on()) {          // recorded (ss_wprep_*)
            SsWJob j{};
            j.type = SS_WJ_TRANSPOSE; j.gx = (c.cout + 31) / 32; j.gy = (c.cin + 31) / 32; j.gz = T;
            j.src = w; j.dst = wt; j.a = c.cin; j.b = c.cout;
            ss_wrec_push(j);
        } else if (fill) {
            hipLaunchKernelGGL(transpose_last2_kernel, dim3((c.cout + 31) / 32, (c.cin + 31) / 32, T), dim3(256), 0, s, w, wt, c.cin, c.cout);
            SS_LAUNCH_CHECK();
        }
    }

    GConvParams p{};
    p.in = dy; p.w = wt; p.bias = bias;
    p.N = c.n; p.IH = c.oh; p.IW = c.ow; p.Cin = c.cout; p.in_cs = c.out_cs;
    p.in_s = 1; p.Cout = c.cin; p.ldb = c.cin; p.reflect = 0; p.act = act; p.alpha = alpha;
    p.dtype = c.dtype;
    p.c1_dtype = c.c1_dtype;
    if (c.c1_dtype == SS_DTYPE_F32 && need_dy_amax_dgrad(c, algo) && gws_bytes >= x6_planes_ub(c.cout, c.cin, T)) {
        unsigned int* sl = (unsigned int*)((char*)gws + x6_planes_ub(c.cout, c.cin, T) - 256);
        const AmaxRef ay = (c.wc && c.wc->fill_only) ? AmaxRef{sl, 1}
                           : (c.dtype == SS_DTYPE_F32 ? act_amax(dy, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl, s)
                                                      : act_amax16(dy, c.dtype, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl, s));
        p.h_amax = ay.p; p.amax_stripes = ay.stripes;
        p.h_amax2 = weight_amax(w, (long)T * c.cin * c.cout, sl + 1, s, c.wc);
    }

    if (c.reflect) {
        const int PH = c.oh + c.kh - 1, PW = c.ow + c.kw - 1;
        float* dpad = (float*)((char*)ws + bwd_data_wt_bytes(c));
        p.out = dpad; p.OH = PH; p.OW = PW; p.out_cs = c.cin; p.OHc = PH; p.OWc = PW;
        p.out_s = 1; p.out_oy = 0; p.out_ox = 0; p.in_oy = 0; p.in_ox = 0; p.accumulate = 0;
        p.ntaps = 0;
        for (int a = 0; a < c.kh; ++a)
            for (int b = 0; b < c.kw; ++b) {
                GTap& t = p.taps[p.ntaps++];
                t.dy = (int16_t)(-a); t.dx = (int16_t)(-b); t.woff = (a * c.kw + b) * c.cin * c.cout;
            }
        if (algo != SS_ALGO_DIRECT && algo != SS_ALGO_MFMA && c.cout == 1) {
            // one-channel gradient: the reflection's transpose is applied to the im2col of dy inside the kernel, dx is written once
            GConvParams q = p;
            q.out = dx; q.out_cs = c.in_cs; q.accumulate = accumulate;
            if (ss_conv_in1_fold_ok(q, c.pt, c.pl, c.ih, c.iw)) return (c.wc && c.wc->fill_only) ? SS_OK : ss_launch_conv_in1_fold(q, c.pt, c.pl, c.ih, c.iw, s);
        }
        if (c.c1_dtype != SS_DTYPE_F32) { ss_set_error("conv_bwd_data: c1_dtype set on a problem the folded one-channel kernel does not take"); return SS_ERR_UNSUPPORTED; }
        int rc = run_gconv(algo, p, gws, gws_bytes, s, c.wc, 1);
        if (rc != SS_OK || (c.wc && c.wc->fill_only)) return rc;
        return launch_reflect_fold(dpad, dx, c.n, c.ih, c.iw, c.cin, c.in_cs, c.pt, c.pl, PH, PW, accumulate, s);
    }

    p.out = dx; p.OH = c.ih; p.OW = c.iw; p.out_cs = c.in_cs; p.accumulate = accumulate;
    p.out_s = c.s; p.in_oy = 0; p.in_ox = 0;
    // the sub-pixel phases (output pixels of one residue class mod s each: their own taps, the same dy)
    GConvParams phs[SS_MAX_PHASES];
    int nph = 0;
    bool collect = c.s * c.s <= SS_MAX_PHASES && (ss_tuning().gconv_phases || (ss_tuning().phases_fused && c.s == 2));
    int rc_all = SS_OK;
    for (int ry = 0; ry < c.s; ++ry)
        for (int rx = 0; rx < c.s; ++rx) {
            p.OHc = (c.ih - ry + c.s - 1) / c.s;
            p.OWc = (c.iw - rx + c.s - 1) / c.s;
            if (p.OHc <= 0 || p.OWc <= 0) continue;
            p.out_oy = ry; p.out_ox = rx;
            p.ntaps = 0;
            for (int a = 0; a < c.kh; ++a) {
                if (((ry + c.pt - a) % c.s) != 0) continue;
                for (int b = 0; b < c.kw; ++b) {
                    if (((rx + c.pl - b) % c.s) != 0) continue;
                    GTap& t = p.taps[p.ntaps++];
                    t.dy = (int16_t)((ry + c.pt - a) / c.s);
                    t.dx = (int16_t)((rx + c.pl - b) / c.s);
                    t.woff = (a * c.kw + b) * c.cin * c.cout;
                }
            }
            if (collect && nph < SS_MAX_PHASES) { phs[nph++] = p; continue; }
            int rc = run_gconv(algo, p, gws, gws_bytes, s, c.wc, 1);
            if (rc != SS_OK) return rc;
        }
    if (nph > 0) {
        // ONE launch for all phases where they line up (same class grid, the x3h gather kernels, <= 4 taps each): the phases of a tile
        // run side by side on one XCD and share its rows of dy in that L2 (ss_launch_gconv_x6_multi); otherwise one launch per phase
        rc_all = run_gconv_phases(algo, phs, nph, gws, gws_bytes, s, c.wc);
    }
    return rc_all;
}

bool wgrad_two_stage(const ConvProb& c, int algo) {
    return algo != SS_ALGO_DIRECT && c.cout == 1 && c.kh * c.kw >= 4 && c.cin >= 16;
}

// LDS-tiled weight-gradient kernel for stride-1 convs with one channel on one side (conv_c1.hip): 0: Cout == 1, 1: Cin == 1, -1: no
int wgrad_c1_mode(const ConvProb& c, int algo) {
    if (algo == SS_ALGO_DIRECT || c.s != 1 || c.ih < 2 * c.pt + 4 || c.iw < 2 * c.pl + 4) return -1;
    if (c.cout == 1 && c.cin > 1 && ss_wgrad_c1_ok(c.n, c.ih, c.iw, c.cin, c.kh, c.kw)) return 0;
    if (c.cin == 1 && c.cout > 1 && ss_wgrad_c1_ok(c.n, c.oh, c.ow, c.cout, c.kh, c.kw)) return 1;
    return -1;
}

// the stride-1, zero-padded data-gradient problem of `c` as conv_bwd_data builds it (pointers left null): eligibility checks only
GConvParams dgrad_params_s1(const ConvProb& c) {
    GConvParams p{};
    p.N = c.n; p.IH = c.oh; p.IW = c.ow; p.Cin = c.cout; p.in_cs = c.out_cs;
    p.in_s = 1; p.Cout = c.cin; p.ldb = c.cin; p.reflect = 0;
    p.OH = c.ih; p.OW = c.iw; p.out_cs = c.in_cs; p.out_s = 1; p.OHc = c.ih; p.OWc = c.iw;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = p.taps[p.ntaps++];
            t.dy = (int16_t)(c.pt - a); t.dx = (int16_t)(c.pl - b); t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    return p;
}
bool tconv_takes_fwd(const ConvProb& c, int algo) {
    return c.kh * c.kw <= SS_MAX_TAPS && tconv_takes(algo, fwd_params(c, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0));
}
bool tconv_takes_dgrad(const ConvProb& c, int algo) {
    return c.kh * c.kw <= SS_MAX_TAPS && c.s == 1 && !c.reflect && tconv_takes(algo, dgrad_params_s1(c));
}
WGradParams wgrad_params(const ConvProb& c, const float* x, const float* dy, float* part) {
    WGradParams p{};
    p.a = x; p.b = dy; p.part = part;
    p.N = c.n; p.AH = c.ih; p.AW = c.iw; p.Ca = c.cin; p.a_cs = c.in_cs;
    p.GH = c.oh; p.GW = c.ow; p.Cb = c.cout; p.b_cs = c.out_cs;
    p.a_s = c.s; p.a_oy = -c.pt; p.a_ox = -c.pl; p.reflect = c.reflect;
    p.dtype = c.dtype;
    p.ntaps = 0;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = p.taps[p.ntaps++];
            t.dy = (int16_t)a; t.dx = (int16_t)b; t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    return p;
}
// LDS-staged tile weight gradient (conv_tile.hip, fp32 MFMA): the shapes the 16-bit split kernel (wgrad_x6) does not take
bool twgrad_takes(const ConvProb& c, int algo) {
    if (!(algo == SS_ALGO_AUTO || algo == SS_ALGO_X6) || c.kh * c.kw > SS_MAX_TAPS) return false;
    const bool x6_shape = c.dtype == SS_DTYPE_F32 && c.cin % 32 == 0 && c.cout >= 32 && c.cout % 4 == 0;
    return !x6_shape && ss_twgrad_ok(wgrad_params(c, nullptr, nullptr, nullptr));
}

bool need_x_amax_fwd(const ConvProb& c, int algo) {
    WinoProb q;
    if (tconv_takes_fwd(c, algo)) return false;           // per-tile scales, taken while the tile is staged
    return x3h_direct_wanted(algo, c.cin, c.cout) && (long)c.n * c.oh * c.ow >= 1024 &&
           c.kh * c.kw <= SS_MAX_TAPS && !wino_fwd_prob(c, algo, &q);
}
bool need_dy_amax_dgrad(const ConvProb& c, int algo) {
    WinoProb q;
    if (tconv_takes_dgrad(c, algo) && !wino_dgrad_prob(c, algo, &q)) return false;
    return x3h_direct_wanted(algo, c.cout, c.cin) && (long)c.n * c.oh * c.ow >= 1024 &&
           c.kh * c.kw <= SS_MAX_TAPS && !(c.reflect && c.s != 1) && !wino_dgrad_prob(c, algo, &q);
}
bool need_amax_wgrad(const ConvProb& c, int algo) {
    WinoProb q;
    if (c.kh * c.kw <= SS_MAX_TAPS && algo != SS_ALGO_DIRECT && wino_fwd_prob(c, algo, &q)) return ss_wino_wgrad_tn(q);
    if (c.kh * c.kw > SS_MAX_TAPS || algo == SS_ALGO_DIRECT || wgrad_c1_mode(c, algo) >= 0 ||
        wgrad_two_stage(c, algo) || twgrad_takes(c, algo))
        return false;
    return x6_wanted(algo) && ss_x3h_enabled() && x3h_direct_wanted(algo, 32, 32) && c.cin % 32 == 0 && c.in_cs % 4 == 0 &&
           c.cout >= 32 && c.cout % 4 == 0 && c.out_cs % 4 == 0 && c.ow >= 4;
}

size_t bwd_weight_ws(const ConvProb& c) {
    int pps;
    const long P = (long)c.n * c.oh * c.ow;
    const int M = c.kh * c.kw * c.cin;
    int splits = ss_wgrad_mfma_splits(P, M, c.cout, &pps);
    {
        const int ssp = ss_wgrad_stage_splits(wgrad_params(c, (const float*)16, (const float*)16, nullptr));      // (aligned dummy pointers: shape-only answer)
        if (ssp > splits) splits = ssp;
    }
    size_t b = ss_align_up((size_t)splits * M * c.cout * sizeof(float), 256) + 256;      // + the x3h amax slot
    if (wgrad_two_stage(c, SS_ALGO_AUTO)) {
        const int tcs = round4(c.kh * c.kw);
        const long Q = (long)c.n * c.ih * c.iw;
        const int sp2 = ss_wgrad_mfma_splits(Q, tcs, c.cin, &pps);
        const size_t b2 = ss_align_up((size_t)Q * tcs * sizeof(float), 256) + ss_align_up((size_t)sp2 * tcs * c.cin * sizeof(float), 256);
        if (b2 > b) b = b2;
    }
    {
        WinoProb q;
        if (wino_fwd_prob(c, SS_ALGO_AUTO, &q)) { const size_t wq = ss_wino_wgrad_ws(q); if (wq > b) b = wq; }
    }
    if (twgrad_takes(c, SS_ALGO_AUTO)) {
        const size_t tq = ss_align_up((size_t)ss_twgrad_splits(wgrad_params(c, nullptr, nullptr, nullptr)) * M * c.cout * sizeof(float), 256);
        if (tq > b) b = tq;
    }
    if (wgrad_c1_mode(c, SS_ALGO_AUTO) >= 0) {
        const bool m0 = wgrad_c1_mode(c, SS_ALGO_AUTO) == 0;
        const size_t wq = m0 ? ss_wgrad_c1_ws(c.n, c.ih, c.iw, c.cin, c.kh, c.kw) : ss_wgrad_c1_ws(c.n, c.oh, c.ow, c.cout, c.kh, c.kw);
        if (wq > b) b = wq;
    }
    b += ss_align_up((size_t)COLSUM_CHUNKS * (c.cout > c.cin ? c.cout : c.cin) * sizeof(float), 256);
    return b;
}

int conv_bwd_weight(const ConvProb& c, const float* x, const float* dy, float* dw, int accumulate, int algo,
                    void* ws, size_t ws_bytes, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    if (ws_bytes < bwd_weight_ws(c) || !ws) return SS_ERR_WORKSPACE;
    {
        WinoProb q;
        if (wino_fwd_prob(c, algo, &q)) {
            q.in_norm = c.in_norm;
            q.saved = c.saved;
            if (c.in_norm.groups > 0 && !(ss_wino_wgrad_tn(q) && c.x_amax && c.x_valid == 1)) {
                ss_set_error("in_norm: the weight gradient needs the pre-split-plane path and the forward pass's max|normalised x| (x_amax, x_amax_valid)");
                return SS_ERR_UNSUPPORTED;
            }
            if (ss_wino_wgrad_tn(q)) {        // pre-split planes need max|x| and max|dy| before the transforms run
                unsigned int* sl = (unsigned int*)((char*)ws + ss_wino_wgrad_ws(q) - 256);
                const AmaxRef ax = act_amax(x, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
                const AmaxRef ay = act_amax(dy, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl + 1, s);
                q.x_amax = ax.p; q.x_stripes = ax.stripes; q.dy_amax = ay.p; q.dy_stripes = ay.stripes;
            }
            return ss_wino_conv_wgrad(q, x, dy, dw, accumulate, ws, ws_bytes, s);
        }
    }
    if (c.in_norm.groups > 0) { ss_set_error("in_norm: this weight-gradient pass does not normalise in its operand load"); return SS_ERR_UNSUPPORTED; }
    {
        const int m = wgrad_c1_mode(c, algo);
        if (m == 0)
            return ss_launch_wgrad_c1(0, x, c.in_cs, c.cin, c.n, c.ih, c.iw, dy, c.out_cs, c.oh, c.ow, c.kh, c.kw, c.pt, c.pl, c.reflect,
                                      dw, accumulate, ws, s, c.c1_dtype);
        if (m == 1)
            return ss_launch_wgrad_c1(1, dy, c.out_cs, c.cout, c.n, c.oh, c.ow, x, c.in_cs, c.ih, c.iw, c.kh, c.kw, c.pt, c.pl, c.reflect,
                                      dw, accumulate, ws, s, c.c1_dtype);
    }
    WGradParams p = wgrad_params(c, x, dy, (float*)ws);
    p.x6 = x6_wanted(algo);
    if (twgrad_takes(c, algo)) {
        p.splits = ss_twgrad_splits(p);
        p.pix_per_split = 0;
        int rc = ss_launch_twgrad_partials(p, s);
        if (rc != SS_OK) return rc;
        return ss_launch_wgrad_reduce(p, dw, c.cout, accumulate, p.ntaps * p.Ca, s);
    }
    if (algo == SS_ALGO_DIRECT) {
        p.splits = 1; p.pix_per_split = 0;
        return ss_launch_wgrad_direct(p, dw, c.cout, accumulate, s);
    }
    int pps;
    if (wgrad_two_stage(c, algo)) {
        // U[q][t] (tap scatter of the one-channel dy), then dw[t][ci] = sum_q U[q][t] * x[q][ci]
        const int tcs = round4(p.ntaps);
        const long Q = (long)c.n * c.ih * c.iw;
        float* U = (float*)ws;
        if (Q * (tcs / 4) >= (1L << 32)) return SS_ERR_UNSUPPORTED;      // 32-bit thread index
        hipLaunchKernelGGL(tapscatter_kernel, dim3((unsigned)((Q * (tcs / 4) + 255) / 256)), dim3(256), 0, s, p, U, tcs);
        SS_LAUNCH_CHECK();
        WGradParams q{};
        q.a = U; q.b = x; q.part = (float*)((char*)ws + ss_align_up((size_t)Q * tcs * sizeof(float), 256));
        q.N = c.n; q.AH = c.ih; q.AW = c.iw; q.Ca = tcs; q.a_cs = tcs;
        q.GH = c.ih; q.GW = c.iw; q.Cb = c.cin; q.b_cs = c.in_cs;
        q.a_s = 1; q.a_oy = 0; q.a_ox = 0; q.reflect = 0;
        q.ntaps = 1; q.taps[0].dy = 0; q.taps[0].dx = 0; q.taps[0].woff = 0;
        q.splits = ss_wgrad_mfma_splits(Q, tcs, c.cin, &pps);
        q.pix_per_split = pps;
        // rows t >= ntaps of U are zero and land beyond the kh*kw*cin weights: mask them by shrinking Ca in the reduce
        return ss_launch_wgrad_mfma_rows(q, dw, c.cin, accumulate, p.ntaps, s);
    }
    if (need_amax_wgrad(c, algo) && ss_wgrad_stage_ok(p)) {
        // stride-2 layers: operands staged once per spatial tile, every tap served from LDS (conv_wgrad_stage.hip)
        p.splits = ss_wgrad_stage_splits(p);
        p.pix_per_split = 0;
        unsigned int* sl = (unsigned int*)((char*)ws + ss_align_up((size_t)p.splits * p.ntaps * p.Ca * p.Cb * sizeof(float), 256));
        const AmaxRef ax = act_amax(x, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
        const AmaxRef ay = act_amax(dy, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl + 1, s);
        p.h_amax = ax.p; p.h_amax2 = ay.p; p.amax_stripes = ax.stripes; p.amax2_stripes = ay.stripes;
        const int rc = ss_launch_wgrad_stage_partials(p, s);
        if (rc != SS_OK) return rc;
        return ss_launch_wgrad_reduce(p, dw, c.cout, accumulate, p.ntaps * p.Ca, s);
    }
    p.splits = ss_wgrad_mfma_splits((long)c.n * c.oh * c.ow, p.ntaps * p.Ca, p.Cb, &pps);
    p.pix_per_split = pps;
    if (need_amax_wgrad(c, algo)) {
        unsigned int* sl = (unsigned int*)((char*)ws + ss_align_up((size_t)p.splits * p.ntaps * p.Ca * p.Cb * sizeof(float), 256));
        const AmaxRef ax = act_amax(x, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
        const AmaxRef ay = act_amax(dy, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl + 1, s);
        if (p.x6 && ss_wgrad_x6_ok(p)) { p.h_amax = ax.p; p.h_amax2 = ay.p; p.amax_stripes = ax.stripes; p.amax2_stripes = ay.stripes; }
    }
    return ss_launch_wgrad_mfma(p, dw, c.cout, accumulate, s);
}

bool valid_desc(const ss_conv_desc* d) {
    if (!d) { ss_set_error("ss_conv_desc is NULL"); return false; }
    if (d->struct_size != sizeof(ss_conv_desc)) {
        ss_set_error("ss_conv_desc.struct_size = %u, this library expects %zu (caller built against another header?)", d->struct_size, sizeof(ss_conv_desc));
        return false;
    }
    if (d->dtype != SS_DTYPE_F32) { ss_set_error("ss_conv_desc.dtype = %d: this entry point takes SS_DTYPE_F32 activations", d->dtype); return false; }
    if (d->n <= 0 || d->ih <= 0 || d->iw <= 0 || d->cin <= 0 || d->oh <= 0 || d->ow <= 0 || d->cout <= 0) return false;
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0) return false;
    if (d->in_cstride < d->cin || d->out_cstride < d->cout) return false;
    if (d->pad_mode == SS_PAD_REFLECT && (d->stride != 1 || d->transposed)) return false;
    if (d->pad_mode == SS_PAD_REFLECT && (d->pad_top >= d->ih || d->pad_left >= d->iw)) return false;
    if (d->in_norm_groups != 0) {
        if (d->transposed || (d->in_norm_groups != 1 && d->in_norm_groups != d->n) || !d->in_norm_mean || !d->in_norm_rstd || !d->in_norm_beta) {
            ss_set_error("ss_conv_desc.in_norm_*: groups must be 1 or n, mean / rstd / beta non-NULL, not a transposed convolution");
            return false;
        }
    }
    return true;
}

ConvProb plain(const ss_conv_desc* d) {
    ConvProb c{d->n, d->ih, d->iw, d->cin, d->in_cstride, d->oh, d->ow, d->cout, d->out_cstride,
               d->kh, d->kw, d->stride, d->pad_top, d->pad_left, d->pad_mode == SS_PAD_REFLECT};
    c.x_amax = (unsigned int*)d->x_amax; c.x_valid = d->x_amax_valid;
    c.dy_amax = (unsigned int*)d->dy_amax; c.dy_valid = d->dy_amax_valid;
    c.saved = d->saved_operand;
    if (d->in_norm_groups > 0) {
        c.in_norm.mean = d->in_norm_mean; c.in_norm.rstd = d->in_norm_rstd; c.in_norm.gamma = d->in_norm_gamma; c.in_norm.beta = d->in_norm_beta;
        c.in_norm.groups = d->in_norm_groups; c.in_norm.act = d->in_norm_act; c.in_norm.alpha = d->in_norm_alpha;
    }
    return c;
}
// adjoint conv of a transposed conv: output space -> input space (its "x" is the transposed conv's dy and vice versa)
ConvProb adjoint(const ss_conv_desc* d) {
    ConvProb c{d->n, d->oh, d->ow, d->cout, d->out_cstride, d->ih, d->iw, d->cin, d->in_cstride,
               d->kh, d->kw, d->stride, d->pad_top, d->pad_left, 0};
    c.x_amax = (unsigned int*)d->dy_amax; c.x_valid = d->dy_amax_valid;
    c.dy_amax = (unsigned int*)d->x_amax; c.dy_valid = d->x_amax_valid;
    return c;
}

}  // namespace

extern "C" {

int ss_version(void) { return 100; }

void ss_wcache_invalidate(ss_wcache* wc) {
    if (!wc) return;
    wc->count = 0;
    wc->used = 0;
}

const char* ss_status_string(int status) {
    switch (status) {
        case SS_OK: return "ok";
        case SS_ERR_INVALID: return "invalid descriptor or pointer";
        case SS_ERR_WORKSPACE: return "workspace too small";
        case SS_ERR_LAUNCH: return "HIP launch failed";
        case SS_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

}  // extern "C"

namespace {

size_t conv2d_workspace_bytes32(const ss_conv_desc* d, int pass) {
    if (!valid_desc(d)) return 0;
    const size_t colsum_b = ss_align_up((size_t)COLSUM_CHUNKS * (d->cout > d->cin ? d->cout : d->cin) * sizeof(float), 256);
    if (!d->transposed) {
        if (pass == SS_PASS_FWD) return fwd_ws(plain(d), d->algo);
        if (pass == SS_PASS_BWD_DATA) return bwd_data_ws(plain(d));
        return bwd_weight_ws(plain(d)) + colsum_b;
    }
    if (pass == SS_PASS_FWD) return bwd_data_ws(adjoint(d));
    if (pass == SS_PASS_BWD_DATA) return fwd_ws(adjoint(d), d->algo);
    return bwd_weight_ws(adjoint(d)) + colsum_b;
}

// which tensor maxima a pass reads (bit 0: of x, bit 1: of dy, in the caller's naming) -- mirrors the dispatch of conv_fwd /
// conv_bwd_data / conv_bwd_weight: only the direct x3h kernels use per-tensor scales (the Winograd passes scale per tile)
int conv2d_uses_amax32(const ss_conv_desc* d, int pass) {
    if (!valid_desc(d)) return 0;
    const ConvProb c = d->transposed ? adjoint(d) : plain(d);
    // pass and roles inside the (adjoint) problem: bit 0 = its x, bit 1 = its dy
    const int p2 = !d->transposed ? pass : (pass == SS_PASS_FWD ? SS_PASS_BWD_DATA : (pass == SS_PASS_BWD_DATA ? SS_PASS_FWD : pass));
    int roles;
    if (p2 == SS_PASS_FWD) roles = need_x_amax_fwd(c, d->algo) ? 1 : 0;
    else if (p2 == SS_PASS_BWD_DATA) roles = need_dy_amax_dgrad(c, d->algo) ? 2 : 0;
    else roles = need_amax_wgrad(c, d->algo) ? 3 : 0;
    if (!d->transposed) return roles;
    return ((roles & 1) ? 2 : 0) | ((roles & 2) ? 1 : 0);
}

// upper bound of what conv_fwd keeps of the weights of `c`
size_t fwd_wcache(const ConvProb& c, int algo) {
    if (c.kh * c.kw > SS_MAX_TAPS) return 0;
    WinoProb q;
    if (wino_fwd_prob(c, algo, &q)) {
        const int R = ss_tuning().wino_r, XI = (R + 2) * (R + 2);
        const size_t planes = ss_align_up((size_t)3 * XI * ss_x6_npad(q.cout) * q.cin * 2, 256), u = ss_align_up((size_t)XI * q.cin * q.cout * 4, 256);
        return 256 + (planes > u ? planes : u);
    }
    const size_t xq = x6_planes_ub(c.cin, c.cout, c.kh * c.kw);
    return xq ? xq + 256 : 0;
}
// ... conv_bwd_data: transposed weights, then (Winograd) transformed planes or one plane set per output phase of a strided conv
size_t bwd_data_wcache(const ConvProb& c, int algo) {
    if (c.kh * c.kw > SS_MAX_TAPS || (c.reflect && c.s != 1)) return 0;
    WinoProb q;
    if (wino_dgrad_prob(c, algo, &q)) {
        const int R = ss_tuning().wino_r, XI = (R + 2) * (R + 2);
        const size_t planes = ss_align_up((size_t)3 * XI * ss_x6_npad(q.cout) * q.cin * 2, 256), u = ss_align_up((size_t)XI * q.cin * q.cout * 4, 256);
        return 256 + (planes > u ? planes : u);
    }
    const size_t xq = x6_planes_ub(c.cout, c.cin, c.kh * c.kw);
    return bwd_data_wt_bytes(c) + 256 + xq + (size_t)256 * c.s * c.s;
}
size_t conv2d_wcache_bytes32(const ss_conv_desc* d, int pass) {
    if (!valid_desc(d) || pass == SS_PASS_BWD_WEIGHT) return 0;
    const bool fwd_like = (pass == SS_PASS_FWD) != (d->transposed != 0);
    const ConvProb c = d->transposed ? adjoint(d) : plain(d);
    return fwd_like ? fwd_wcache(c, d->algo) : bwd_data_wcache(c, d->algo);
}
// chunks per sample of the output statistics the forward pass can emit (ss_conv_desc::y_stats): Winograd forward, the 1 -> C matrix-core
// kernel, gconv_x6v2
int conv2d_stats_chunks32(const ss_conv_desc* d) {
    if (!valid_desc(d) || d->transposed || d->act != SS_ACT_NONE) return 0;
    const ConvProb c = plain(d);
    WinoProb q;
    if (c.kh * c.kw > SS_MAX_TAPS) return 0;
    if (!wino_fwd_prob(c, d->algo, &q)) return gconv_stats_chunks(c, d->algo);
    return ss_wino_stats_chunks(q);
}

// the layer's weight cache when the descriptor carries a usable one
WCache* desc_wcache(const ss_conv_desc* d) {
    ss_wcache* wc = d->w_cache;
    return wc && wc->struct_size == sizeof(ss_wcache) && wc->base && wc->count >= 0 ? wc : nullptr;
}

int conv2d_fwd32(const ss_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                  void* ws, size_t ws_bytes, void* stream, int c1_dtype = SS_DTYPE_F32) {
    const bool fill_only = desc_wcache(d) && desc_wcache(d)->fill_only;
    if (!valid_desc(d) || !w || (!fill_only && (!x || !y))) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    ConvProb c = d->transposed ? adjoint(d) : plain(d);
    c.wc = desc_wcache(d);
    c.c1_dtype = c1_dtype;
    if (!d->transposed && d->act == SS_ACT_NONE && conv2d_stats_chunks32(d) > 0) c.y_stats = (float*)d->y_stats;
    if (!d->transposed) return conv_fwd(c, x, w, bias, y, d->act, d->act_alpha, 0, d->algo, ws, ws_bytes, s);
    return conv_bwd_data(c, x, w, y, bias, d->act, d->act_alpha, 0, d->algo, ws, ws_bytes, s);
}

int conv2d_bwd_data32(const ss_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                       void* ws, size_t ws_bytes, void* stream, int c1_dtype = SS_DTYPE_F32) {
    const bool fill_only = desc_wcache(d) && desc_wcache(d)->fill_only;
    if (!valid_desc(d) || !w || (!fill_only && (!dy || !dx))) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    ConvProb c = d->transposed ? adjoint(d) : plain(d);
    c.wc = desc_wcache(d);
    c.c1_dtype = c1_dtype;
    if (!d->transposed)
        return conv_bwd_data(c, dy, w, dx, nullptr, SS_ACT_NONE, 0.f, accumulate, d->algo, ws, ws_bytes, s);
    return conv_fwd(c, dy, w, nullptr, dx, SS_ACT_NONE, 0.f, accumulate, d->algo, ws, ws_bytes, s);
}

int conv2d_bwd_weight32(const ss_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                         int accumulate, void* ws, size_t ws_bytes, void* stream, int c1_dtype = SS_DTYPE_F32) {
    if (!valid_desc(d) || !x || !dy || !dw) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (ws_bytes < conv2d_workspace_bytes32(d, SS_PASS_BWD_WEIGHT) || !ws) return SS_ERR_WORKSPACE;
    int rc;
    ConvProb c = d->transposed ? adjoint(d) : plain(d);
    c.c1_dtype = c1_dtype;
    const size_t main_b = bwd_weight_ws(c);
    if (!d->transposed) rc = conv_bwd_weight(c, x, dy, dw, accumulate, d->algo, ws, main_b, s);
    else rc = conv_bwd_weight(c, dy, x, dw, accumulate, d->algo, ws, main_b, s);
    if (rc != SS_OK) return rc;
    if (dbias) {
        float* part = (float*)((char*)ws + main_b);
        rc = colsum(dy, (long)d->n * d->oh * d->ow, d->cout, d->out_cstride, dbias, accumulate, part, s);
    }
    return rc;
}


// ---- 16-bit activation storage (ss_conv_desc::dtype = SS_DTYPE_BF16 / SS_DTYPE_F16) ------------------------------------------------
// Every convolution path computes on fp32-grade operands; a path without a native 16-bit loader (all but the tile kernels, for now)
// runs on dense fp32 staging copies of its activation arguments taken from the caller's workspace: convert in, run the fp32 path,
// convert out.  Weights, weight gradients and bias gradients are fp32 throughout.
struct ConvShim {
    ss_conv_desc d32;
    float* a;          // [n*ih*iw*cin]   x / dx
    float* b;          // [n*oh*ow*cout]  y / dy
    void* ws;
    size_t ws_bytes, a_bytes, b_bytes;
};
ConvShim make_shim(const ss_conv_desc* d, void* ws, size_t ws_bytes) {
    ConvShim sh;
    sh.d32 = *d;
    sh.d32.dtype = SS_DTYPE_F32;
    sh.d32.in_cstride = d->cin;
    sh.d32.out_cstride = d->cout;
    // (the caller's amax slots stay: the maximum of the fp32 staging copy IS the maximum of the stored tensor, so a slot a norm
    // kernel filled saves the scan, and a scan done here serves the caller's later passes)
    sh.d32.y_stats = nullptr;
    sh.d32.in_norm_groups = 0;                 // fp32 storage only (valid_desc_any rejects it for the 16-bit types)
    sh.a_bytes = ss_align_up((size_t)d->n * d->ih * d->iw * d->cin * sizeof(float), 256);
    sh.b_bytes = ss_align_up((size_t)d->n * d->oh * d->ow * d->cout * sizeof(float), 256);
    sh.a = (float*)ws;
    sh.b = (float*)((char*)ws + sh.a_bytes);
    sh.ws = (char*)ws + sh.a_bytes + sh.b_bytes;
    sh.ws_bytes = ws_bytes > sh.a_bytes + sh.b_bytes ? ws_bytes - sh.a_bytes - sh.b_bytes : 0;
    return sh;
}
// ---- native 16-bit paths: the LDS-staged tile kernels (conv_tile.hip) load / store bf16 / fp16 themselves --------------------------
// Winograd x3h-plane shapes on 16-bit stored activations: one fp16 plane per operand, one product, transforms in the storage type
bool wino16_fwd_takes(const ConvProb& c, int algo, WinoProb* q) {
    return (algo == SS_ALGO_AUTO || algo == SS_ALGO_X6) && c.kh * c.kw <= SS_MAX_TAPS && wino_fwd_prob(c, algo, q) && ss_tuning().wino_r == 4 &&
           ss_wino_fwd_x3h(*q);
}
bool wino16_dgrad_takes(const ConvProb& c, int algo, WinoProb* q) {
    return (algo == SS_ALGO_AUTO || algo == SS_ALGO_X6) && c.kh * c.kw <= SS_MAX_TAPS && !(c.reflect && c.s != 1) && wino_dgrad_prob(c, algo, q) &&
           (!c.reflect || q->fold_h > 0) && ss_tuning().wino_r == 4 && ss_wino_fwd_x3h(*q);
}

// Gather convolutions on 16-bit stored activations: gconv_x6v2 / gconv_x6 read and write the stored type (one fp16 operand plane).
// `p` is the problem run_gconv would see; true when run_gconv's choice for it is that kernel.
bool gconv16_takes(int algo, const GConvParams& p0) {
    if (algo != SS_ALGO_AUTO && algo != SS_ALGO_X6) return false;
    static const unsigned int dummy = 0;
    GConvParams p = p0;
    p.h_amax = &dummy; p.h_amax2 = &dummy; p.amax_stripes = 1;
    if (ss_conv_out1_ok(p) || ss_conv_in1_ok(p) || tconv_takes(algo, p) || gconv_two_stage(algo, p)) return false;
    return use_x6(algo, p) && (ss_gconv_x6v2_ok(p) || ss_gconv_x6_typed_ok(p));
}
// ... every sub-pixel phase of the data gradient of `c` (the loop of conv_bwd_data)
bool bwd_data_gather16_ok(const ConvProb& c, int algo) {
    WinoProb q;
    if (c.reflect || c.kh * c.kw > SS_MAX_TAPS || wino_dgrad_prob(c, algo, &q) || tconv_takes_dgrad(c, algo) || !need_dy_amax_dgrad(c, algo)) return false;
    GConvParams p{};
    p.N = c.n; p.IH = c.oh; p.IW = c.ow; p.Cin = c.cout; p.in_cs = c.out_cs;
    p.in_s = 1; p.Cout = c.cin; p.ldb = c.cin; p.dtype = c.dtype;
    p.OH = c.ih; p.OW = c.iw; p.out_cs = c.in_cs; p.out_s = c.s;
    for (int ry = 0; ry < c.s; ++ry)
        for (int rx = 0; rx < c.s; ++rx) {
            p.OHc = (c.ih - ry + c.s - 1) / c.s;
            p.OWc = (c.iw - rx + c.s - 1) / c.s;
            if (p.OHc <= 0 || p.OWc <= 0) continue;
            p.out_oy = ry; p.out_ox = rx;
            p.ntaps = 0;
            for (int a = 0; a < c.kh; ++a) {
                if (((ry + c.pt - a) % c.s) != 0) continue;
                for (int b = 0; b < c.kw; ++b)
                    if (((rx + c.pl - b) % c.s) == 0) ++p.ntaps;
            }
            if (!gconv16_takes(algo, p)) return false;
        }
    return true;
}

int native16_fwd(const ss_conv_desc* d, const void* x, const float* w, const float* bias, void* y, void* ws, size_t ws_bytes, hipStream_t s,
                 bool* taken) {
    *taken = false;
    if (d->transposed) {          // a transposed convolution's forward is the data gradient of its adjoint
        ConvProb c = adjoint(d);
        c.dtype = d->dtype;
        c.wc = desc_wcache(d);
        if (!bwd_data_gather16_ok(c, d->algo) || !ws || ws_bytes < bwd_data_ws(c)) return SS_OK;
        *taken = true;
        return conv_bwd_data(c, (const float*)x, w, (float*)y, bias, d->act, d->act_alpha, 0, d->algo, ws, ws_bytes, s);
    }
    ConvProb c = plain(d);
    c.dtype = d->dtype;
    WinoProb q;
    if (wino16_fwd_takes(c, d->algo, &q) && ws && ws_bytes >= ss_wino_fwd_ws(q)) {
        *taken = true;
        q.wc = desc_wcache(d);
        q.in_norm = c.in_norm;          // x is a pre-normalisation tensor: normalised in the input transform (ss_conv_desc::in_norm_*)
        if (c.in_norm.groups > 0 && c.x_amax && c.x_valid != 1 && !(q.wc && q.wc->fill_only)) {      // the transform reports max|normalised x|
            if (c.x_valid != 2) (void)hipMemsetAsync(c.x_amax, 0, (size_t)SS_AMAX_STRIPES * SS_AMAX_STRIDE * 4, s);
            q.in_norm.amax_out = c.x_amax;
        }
        return ss_wino_conv_fwd16(q, d->dtype, x, w, c.cin, c.cout, 0, bias, y, d->act, d->act_alpha, 0, ws, ws_bytes, s);
    }
    if (c.in_norm.groups > 0) {          // (taken, so that the caller does not fall through to the staged fp32 path, which would ignore it)
        *taken = true;
        ss_set_error("in_norm: this forward pass does not normalise in its operand load (ss_conv2d_fuses_in_norm)");
        return SS_ERR_UNSUPPORTED;
    }
    if (tconv_takes_fwd(c, d->algo)) {
        *taken = true;
        const GConvParams p = fwd_params(c, (const float*)x, w, bias, (float*)y, d->act, d->act_alpha, 0);
        return ss_launch_tconv(p, nullptr, 0, s);
    }
    if (c.kh * c.kw > SS_MAX_TAPS || wino_fwd_prob(c, d->algo, &q) || !need_x_amax_fwd(c, d->algo)) return SS_OK;
    GConvParams p = fwd_params(c, (const float*)x, w, bias, (float*)y, d->act, d->act_alpha, 0);
    if (!gconv16_takes(d->algo, p) || !ws || ws_bytes < ss_gconv_x6_planes_bytes(p) + 256) return SS_OK;
    *taken = true;
    c.wc = desc_wcache(d);
    unsigned int* sl = (unsigned int*)((char*)ws + ss_gconv_x6_planes_bytes(p));
    const AmaxRef ax = act_amax16(x, d->dtype, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
    p.h_amax = ax.p; p.amax_stripes = ax.stripes;
    p.h_amax2 = weight_amax(w, (long)c.kh * c.kw * c.cin * c.cout, sl + 1, s, c.wc);
    return run_gconv(d->algo, p, ws, ws_bytes, s, c.wc);
}
int native16_bwd_data(const ss_conv_desc* d, const void* dy, const float* w, void* dx, int accumulate, void* ws, size_t ws_bytes,
                      hipStream_t s, bool* taken) {
    *taken = false;
    if (d->transposed) {          // the data gradient of a transposed convolution is the forward of its adjoint
        ConvProb c = adjoint(d);
        c.dtype = d->dtype;
        WinoProb q;
        if (c.kh * c.kw > SS_MAX_TAPS || wino_fwd_prob(c, d->algo, &q) || tconv_takes_fwd(c, d->algo) || !need_x_amax_fwd(c, d->algo)) return SS_OK;
        GConvParams p = fwd_params(c, (const float*)dy, w, nullptr, (float*)dx, SS_ACT_NONE, 0.f, accumulate);
        if (!gconv16_takes(d->algo, p) || !ws || ws_bytes < ss_gconv_x6_planes_bytes(p) + 256) return SS_OK;
        *taken = true;
        c.wc = desc_wcache(d);
        unsigned int* sl = (unsigned int*)((char*)ws + ss_gconv_x6_planes_bytes(p));
        const AmaxRef ax = act_amax16(dy, d->dtype, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
        p.h_amax = ax.p; p.amax_stripes = ax.stripes;
        p.h_amax2 = weight_amax(w, (long)c.kh * c.kw * c.cin * c.cout, sl + 1, s, c.wc);
        return run_gconv(d->algo, p, ws, ws_bytes, s, c.wc);
    }
    ConvProb c = plain(d);
    c.dtype = d->dtype;
    if (bwd_data_gather16_ok(c, d->algo) && ws && ws_bytes >= bwd_data_ws(c)) {
        *taken = true;
        c.wc = desc_wcache(d);
        return conv_bwd_data(c, (const float*)dy, w, (float*)dx, nullptr, SS_ACT_NONE, 0.f, accumulate, d->algo, ws, ws_bytes, s);
    }
    {
        WinoProb q;
        if (wino16_dgrad_takes(c, d->algo, &q) && ws && ws_bytes >= ss_wino_fwd_ws(q)) {
            *taken = true;
            q.wc = desc_wcache(d);
            return ss_wino_conv_fwd16(q, d->dtype, dy, w, c.cin, c.cout, 1, nullptr, dx, SS_ACT_NONE, 0.f, accumulate, ws, ws_bytes, s);
        }
    }
    if (!tconv_takes_dgrad(c, d->algo) || !ws || ws_bytes < bwd_data_wt_bytes(c)) return SS_OK;
    *taken = true;
    float* wt = (float*)ws;
    hipLaunchKernelGGL(transpose_last2_kernel, dim3((c.cout + 31) / 32, (c.cin + 31) / 32, c.kh * c.kw), dim3(256), 0, s, w, wt, c.cin, c.cout);
    SS_LAUNCH_CHECK();
    GConvParams p = dgrad_params_s1(c);
    p.in = (const float*)dy; p.w = wt; p.out = (float*)dx; p.accumulate = accumulate; p.act = SS_ACT_NONE; p.dtype = d->dtype;
    return ss_launch_tconv(p, nullptr, 0, s);
}
int native16_bwd_weight(const ss_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate, void* ws,
                        size_t ws_bytes, hipStream_t s, bool* taken) {
    *taken = false;
    // gather weight gradient (strided / 4x4 / transposed layers): wgrad_x6_kernel reads the stored 16-bit operands itself -- one fp16
    // plane each, one product = the exact product of the stored values.  `xa` / `ya`: input / output-side operand of the plain conv `cc`
    auto gather = [&](const ConvProb& cc, const void* xa, const void* ya) -> int {
        WinoProb q;
        if (dbias || (d->algo != SS_ALGO_AUTO && d->algo != SS_ALGO_X6) || cc.kh * cc.kw > SS_MAX_TAPS || wino_fwd_prob(cc, d->algo, &q) ||
            twgrad_takes(cc, d->algo) || wgrad_c1_mode(cc, d->algo) >= 0 || wgrad_two_stage(cc, d->algo) || !need_amax_wgrad(cc, d->algo) ||
            !ws || ws_bytes < bwd_weight_ws(cc))
            return SS_OK;
        WGradParams p = wgrad_params(cc, (const float*)xa, (const float*)ya, (float*)ws);
        p.x6 = x6_wanted(d->algo);
        if (ss_wgrad_stage_ok(p)) {          // stride-2 layers: operands staged once per spatial tile (conv_wgrad_stage.hip), one plane, one product
            *taken = true;
            p.splits = ss_wgrad_stage_splits(p);
            p.pix_per_split = 0;
            unsigned int* sl = (unsigned int*)((char*)ws + ss_align_up((size_t)p.splits * p.ntaps * p.Ca * p.Cb * sizeof(float), 256));
            const AmaxRef ax = act_amax16(xa, d->dtype, (long)cc.n * cc.ih * cc.iw, cc.cin, cc.in_cs, cc.x_amax, cc.x_valid, sl, s);
            const AmaxRef ay = act_amax16(ya, d->dtype, (long)cc.n * cc.oh * cc.ow, cc.cout, cc.out_cs, cc.dy_amax, cc.dy_valid, sl + 1, s);
            p.h_amax = ax.p; p.h_amax2 = ay.p; p.amax_stripes = ax.stripes; p.amax2_stripes = ay.stripes;
            const int rc = ss_launch_wgrad_stage_partials(p, s);
            if (rc != SS_OK) return rc;
            return ss_launch_wgrad_reduce(p, dw, cc.cout, accumulate, p.ntaps * p.Ca, s);
        }
        int pps;
        p.splits = ss_wgrad_mfma_splits((long)cc.n * cc.oh * cc.ow, p.ntaps * p.Ca, p.Cb, &pps);
        p.pix_per_split = pps;
        if (!(p.x6 && ss_wgrad_x6_ok(p))) return SS_OK;
        *taken = true;
        unsigned int* sl = (unsigned int*)((char*)ws + ss_align_up((size_t)p.splits * p.ntaps * p.Ca * p.Cb * sizeof(float), 256));
        const AmaxRef ax = act_amax16(xa, d->dtype, (long)cc.n * cc.ih * cc.iw, cc.cin, cc.in_cs, cc.x_amax, cc.x_valid, sl, s);
        const AmaxRef ay = act_amax16(ya, d->dtype, (long)cc.n * cc.oh * cc.ow, cc.cout, cc.out_cs, cc.dy_amax, cc.dy_valid, sl + 1, s);
        p.h_amax = ax.p; p.h_amax2 = ay.p; p.amax_stripes = ax.stripes; p.amax2_stripes = ay.stripes;
        return ss_launch_wgrad_mfma(p, dw, cc.cout, accumulate, s);
    };
    if (d->transposed) {          // the weight gradient of a transposed convolution is that of its adjoint with the operands swapped
        ConvProb ca = adjoint(d);
        ca.dtype = d->dtype;
        return gather(ca, dy, x);
    }
    ConvProb c = plain(d);
    c.dtype = d->dtype;
    {   // Winograd weight gradient on pre-split planes: the transforms read the stored type; planes + GEMM as for fp32 storage
        WinoProb q;
        if ((d->algo == SS_ALGO_AUTO || d->algo == SS_ALGO_X6) && c.kh * c.kw <= SS_MAX_TAPS && wino_fwd_prob(c, d->algo, &q) &&
            ss_tuning().wino_r == 4 && ss_wino_wgrad_tn(q) && !dbias && ws && ws_bytes >= ss_wino_wgrad_ws(q)) {
            *taken = true;
            q.in_norm = c.in_norm;
            if (c.in_norm.groups > 0 && !(c.x_amax && c.x_valid == 1)) {
                ss_set_error("in_norm: the weight gradient needs the forward pass's max|normalised x| (x_amax, x_amax_valid)");
                return SS_ERR_UNSUPPORTED;
            }
            unsigned int* sl = (unsigned int*)((char*)ws + ss_wino_wgrad_ws(q) - 256);
            const AmaxRef ax = act_amax16(x, d->dtype, (long)c.n * c.ih * c.iw, c.cin, c.in_cs, c.x_amax, c.x_valid, sl, s);
            const AmaxRef ay = act_amax16(dy, d->dtype, (long)c.n * c.oh * c.ow, c.cout, c.out_cs, c.dy_amax, c.dy_valid, sl + 1, s);
            q.x_amax = ax.p; q.x_stripes = ax.stripes; q.dy_amax = ay.p; q.dy_stripes = ay.stripes;
            return ss_wino_conv_wgrad16(q, d->dtype, x, dy, dw, accumulate, ws, ss_wino_wgrad_ws(q), s);
        }
    }
    if (c.in_norm.groups > 0) {
        *taken = true;
        ss_set_error("in_norm: this weight-gradient pass does not normalise in its operand load");
        return SS_ERR_UNSUPPORTED;
    }
    if (dbias) return SS_OK;
    if (!twgrad_takes(c, d->algo)) return gather(c, x, dy);
    WGradParams p = wgrad_params(c, (const float*)x, (const float*)dy, (float*)ws);
    p.x6 = x6_wanted(d->algo);
    p.splits = ss_twgrad_splits(p);
    if (!ws || ws_bytes < (size_t)p.splits * p.ntaps * p.Ca * p.Cb * sizeof(float)) return SS_OK;
    *taken = true;
    int rc = ss_launch_twgrad_partials(p, s);
    if (rc != SS_OK) return rc;
    return ss_launch_wgrad_reduce(p, dw, c.cout, accumulate, p.ntaps * p.Ca, s);
}

// ---- one-channel layers (the generators' 7x7 stem / head, the discriminators' 4x4 stem) on 16-bit storage ------------------------------
// The matrix-core kernels of conv_c1.hip read / write the MULTI-channel tensor in its stored type themselves (GConvParams::c1_dtype,
// C1WParams::x_dtype); only the ONE-channel tensor (1 / C of the bytes) goes through an fp32 staging copy in the shim's buffers.  The
// predicates below say whether the fp32 path would hand the problem to exactly those kernels; `dm` = the shim descriptor with the
// multi-channel side's real stride.  Returns 0: no; 1: the input side is the one-channel tensor; 2: the output side is.
bool c1_algo(int algo) { return algo == SS_ALGO_AUTO || algo == SS_ALGO_X6; }
int c1_typed_fwd(const ss_conv_desc* d, const ConvShim& sh, const void* x, const float* w, const float* bias, void* y, ss_conv_desc* dm) {
    if (d->transposed || !c1_algo(d->algo) || d->kh * d->kw > SS_MAX_TAPS) return 0;
    *dm = sh.d32;
    WinoProb q;
    if (d->cin == 1 && d->cout > 1) {
        dm->out_cstride = d->out_cstride;
        ConvProb c = plain(dm);
        c.c1_dtype = d->dtype;
        if (wino_fwd_prob(c, d->algo, &q)) return 0;
        const GConvParams p = fwd_params(c, sh.a, w, bias, (float*)y, d->act, d->act_alpha, 0);
        return ss_conv_in1_typed_ok(p) ? 1 : 0;
    }
    if (d->cout == 1 && d->cin > 1) {
        dm->in_cstride = d->in_cstride;
        ConvProb c = plain(dm);
        c.c1_dtype = d->dtype;
        if (wino_fwd_prob(c, d->algo, &q)) return 0;
        const GConvParams p = fwd_params(c, (const float*)x, w, bias, sh.b, d->act, d->act_alpha, 0);
        return ss_conv_out1_typed_ok(p) ? 2 : 0;
    }
    return 0;
}
// data gradient of a reflection-padded C -> 1 layer (the head): dy is the one-channel tensor, dx is written in its stored type by the
// folded 1 -> C kernel (the q of conv_bwd_data's reflect branch)
bool c1_typed_dgrad(const ss_conv_desc* d, const ConvShim& sh, void* dx, int accumulate, ss_conv_desc* dm) {
    if (d->transposed || !c1_algo(d->algo) || d->kh * d->kw > SS_MAX_TAPS || d->cout != 1 || d->cin <= 1) return false;
    *dm = sh.d32;
    dm->in_cstride = d->in_cstride;
    const ConvProb c = plain(dm);
    WinoProb wq;
    if (!c.reflect || c.s != 1 || wino_dgrad_prob(c, d->algo, &wq)) return false;
    GConvParams q{};
    q.in = sh.b; q.N = c.n; q.IH = c.oh; q.IW = c.ow; q.Cin = c.cout; q.in_cs = c.out_cs;
    q.in_s = 1; q.Cout = c.cin; q.ldb = c.cin; q.reflect = 0; q.act = SS_ACT_NONE;
    q.OH = c.oh + c.kh - 1; q.OW = c.ow + c.kw - 1; q.OHc = q.OH; q.OWc = q.OW;
    q.out_s = 1; q.ntaps = 0;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = q.taps[q.ntaps++];
            t.dy = (int16_t)(-a); t.dx = (int16_t)(-b); t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    q.out = (float*)dx; q.out_cs = c.in_cs; q.accumulate = accumulate;
    q.c1_dtype = d->dtype;
    return ss_conv_in1_fold_ok(q, c.pt, c.pl, c.ih, c.iw);
}
// weight gradient: 1 = the input side is the one-channel tensor (X = dy), 2 = the output side is (X = x)
int c1_typed_wgrad(const ss_conv_desc* d, const ConvShim& sh, const void* x, const void* dy, const float* dbias, ss_conv_desc* dm) {
    if (d->transposed || !c1_algo(d->algo) || d->kh * d->kw > SS_MAX_TAPS) return 0;
    *dm = sh.d32;
    WinoProb q;
    if (d->cin == 1 && d->cout > 1 && !dbias) {          // (the bias gradient sums dy: the multi-channel tensor here)
        dm->out_cstride = d->out_cstride;
        const ConvProb c = plain(dm);
        if (wino_fwd_prob(c, d->algo, &q) || wgrad_c1_mode(c, d->algo) != 1) return 0;
        return ss_wgrad_c1_typed_ok(dy, c.out_cs, c.cout, c.oh, c.ow, c.kh, c.kw, c.pt, c.pl) ? 1 : 0;
    }
    if (d->cout == 1 && d->cin > 1) {
        dm->in_cstride = d->in_cstride;
        const ConvProb c = plain(dm);
        if (wino_fwd_prob(c, d->algo, &q) || wgrad_c1_mode(c, d->algo) != 0) return 0;
        return ss_wgrad_c1_typed_ok(x, c.in_cs, c.cin, c.ih, c.iw, c.kh, c.kw, c.pt, c.pl) ? 2 : 0;
    }
    return 0;
}

bool valid_desc_any(const ss_conv_desc* d) {
    if (!d) { ss_set_error("ss_conv_desc is NULL"); return false; }
    if (d->struct_size != sizeof(ss_conv_desc)) return valid_desc(d);          // sets the message
    if (d->dtype == SS_DTYPE_F32) return valid_desc(d);
    if (d->dtype != SS_DTYPE_BF16 && d->dtype != SS_DTYPE_F16) { ss_set_error("ss_conv_desc.dtype = %d unknown", d->dtype); return false; }
    ss_conv_desc t = *d;          // (in_norm_*: checked by valid_desc; only the passes ss_conv2d_fuses_in_norm names take it, the others fail)
    t.dtype = SS_DTYPE_F32;
    return valid_desc(&t);
}

}  // namespace

extern "C" {

size_t ss_conv2d_workspace_bytes(const ss_conv_desc* d, int pass) {
    if (!valid_desc_any(d)) return 0;
    if (d->dtype == SS_DTYPE_F32) return conv2d_workspace_bytes32(d, pass);
    const ConvShim sh = make_shim(d, nullptr, 0);
    size_t need = sh.a_bytes + sh.b_bytes + conv2d_workspace_bytes32(&sh.d32, pass);
    if (pass == SS_PASS_BWD_WEIGHT && !d->transposed) {        // native tile weight gradient: one fp32 partial per workgroup
        ConvProb c = plain(d);
        c.dtype = d->dtype;
        if (twgrad_takes(c, d->algo)) {
            const size_t tq = (size_t)ss_twgrad_splits(wgrad_params(c, nullptr, nullptr, nullptr)) * c.kh * c.kw * c.cin * c.cout * sizeof(float);
            if (tq > need) need = tq;
        }
    }
    return need;
}

int ss_conv2d_stats_chunks(const ss_conv_desc* d) {
    if (!valid_desc_any(d) || d->dtype != SS_DTYPE_F32) return 0;
    return conv2d_stats_chunks32(d);
}

size_t ss_conv2d_wcache_bytes(const ss_conv_desc* d, int pass) {
    if (!valid_desc_any(d)) return 0;
    if (d->dtype == SS_DTYPE_F32) return conv2d_wcache_bytes32(d, pass);
    const ConvShim sh = make_shim(d, nullptr, 0);           // 16-bit storage: the staged fp32 problem has the same weights
    return conv2d_wcache_bytes32(&sh.d32, pass);
}

int ss_conv2d_uses_amax(const ss_conv_desc* d, int pass) {
    if (!valid_desc_any(d)) return 0;
    if (d->dtype == SS_DTYPE_F32) return conv2d_uses_amax32(d, pass);
    // 16-bit storage: the native paths scale per tile (tile kernels, Winograd forward / data gradient); the others run the fp32
    // problem on staging copies and take the maxima that problem takes
    if (!d->transposed) {
        ConvProb c = plain(d);
        c.dtype = d->dtype;
        WinoProb q;
        if (pass == SS_PASS_FWD && (wino16_fwd_takes(c, d->algo, &q) || tconv_takes_fwd(c, d->algo))) return 0;
        if (pass == SS_PASS_BWD_DATA && (wino16_dgrad_takes(c, d->algo, &q) || tconv_takes_dgrad(c, d->algo))) return 0;
        if (pass == SS_PASS_BWD_WEIGHT && twgrad_takes(c, d->algo)) return 0;
    }
    const ConvShim sh = make_shim(d, nullptr, 0);
    return conv2d_uses_amax32(&sh.d32, pass);
}

size_t ss_conv2d_saved_operand_bytes(const ss_conv_desc* d) {
    if (!d || d->struct_size != sizeof(ss_conv_desc) || d->dtype != SS_DTYPE_F32 || d->transposed) return 0;
    ss_conv_desc t = *d;
    t.in_norm_groups = 0;
    if (!valid_desc(&t)) return 0;
    const ConvProb c = plain(&t);
    WinoProb q;
    if (c.kh * c.kw > SS_MAX_TAPS || !wino_fwd_prob(c, d->algo, &q)) return 0;
    return ss_wino_saved_bytes(q);
}

int ss_conv2d_fuses_in_norm(const ss_conv_desc* d, int pass) {
    if (!d || d->struct_size != sizeof(ss_conv_desc) || d->transposed) return 0;
    ss_conv_desc t = *d;
    t.in_norm_groups = 0;
    if (!valid_desc_any(&t)) return 0;
    if (d->dtype != SS_DTYPE_F32) {          // 16-bit storage: the native Winograd passes (native16_fwd / native16_bwd_weight)
        ConvProb c = plain(&t);
        c.dtype = d->dtype;
        WinoProb q;
        if (pass == SS_PASS_FWD) return wino16_fwd_takes(c, d->algo, &q) && c.cin % 32 == 0 ? 1 : 0;
        if (pass == SS_PASS_BWD_WEIGHT)
            return (d->algo == SS_ALGO_AUTO || d->algo == SS_ALGO_X6) && c.kh * c.kw <= SS_MAX_TAPS && wino_fwd_prob(c, d->algo, &q) &&
                   ss_tuning().wino_r == 4 && ss_wino_wgrad_tn(q) ? 1 : 0;
        return 0;
    }
    const ConvProb c = plain(&t);
    WinoProb q;
    if (c.kh * c.kw > SS_MAX_TAPS || !wino_fwd_prob(c, d->algo, &q)) return 0;
    if (pass == SS_PASS_FWD) return ss_wino_fwd_x3h(q) ? 1 : 0;
    if (pass == SS_PASS_BWD_WEIGHT) return ss_wino_wgrad_tn(q) ? 1 : 0;
    return 0;
}

int ss_conv2d_fwd(const ss_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                  void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc_any(d)) return SS_ERR_INVALID;
    if (d->dtype == SS_DTYPE_F32) return conv2d_fwd32(d, (const float*)x, w, bias, (float*)y, ws, ws_bytes, stream);
    const bool fill_only = desc_wcache(d) && desc_wcache(d)->fill_only;
    if (!w || (!fill_only && (!x || !y))) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_conv2d_workspace_bytes(d, SS_PASS_FWD)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (fill_only) {           // refresh of the layer's cached weight operands: those of the staged fp32 problem (the tile kernels keep none)
        ConvProb c = plain(d);
        c.dtype = d->dtype;
        if (!d->transposed && tconv_takes_fwd(c, d->algo)) return SS_OK;
        const ConvShim sh = make_shim(d, ws, ws_bytes);
        return conv2d_fwd32(&sh.d32, nullptr, w, bias, nullptr, sh.ws, sh.ws_bytes, stream);
    }
    bool taken;
    int rc = native16_fwd(d, x, w, bias, y, ws, ws_bytes, s, &taken);
    if (taken) return rc;
    const ConvShim sh = make_shim(d, ws, ws_bytes);
    {   // one-channel layers: only the one-channel tensor is staged, the matrix-core kernel reads / writes the other one as stored
        ss_conv_desc dm;
        const int side = c1_typed_fwd(d, sh, x, w, bias, y, &dm);
        if (side == 1) {
            rc = ss_convert_launch(x, d->dtype, d->in_cstride, sh.a, SS_DTYPE_F32, 1, (long)d->n * d->ih * d->iw, 1, s);
            if (rc != SS_OK) return rc;
            return conv2d_fwd32(&dm, sh.a, w, bias, (float*)y, sh.ws, sh.ws_bytes, stream, d->dtype);
        }
        if (side == 2) {
            rc = conv2d_fwd32(&dm, (const float*)x, w, bias, sh.b, sh.ws, sh.ws_bytes, stream, d->dtype);
            if (rc != SS_OK) return rc;
            return ss_convert_launch(sh.b, SS_DTYPE_F32, 1, y, d->dtype, d->out_cstride, (long)d->n * d->oh * d->ow, 1, s);
        }
    }
    rc = ss_convert_launch(x, d->dtype, d->in_cstride, sh.a, SS_DTYPE_F32, d->cin, (long)d->n * d->ih * d->iw, d->cin, s);
    if (rc != SS_OK) return rc;
    rc = conv2d_fwd32(&sh.d32, sh.a, w, bias, sh.b, sh.ws, sh.ws_bytes, stream);
    if (rc != SS_OK) return rc;
    return ss_convert_launch(sh.b, SS_DTYPE_F32, d->cout, y, d->dtype, d->out_cstride, (long)d->n * d->oh * d->ow, d->cout, s);
}

int ss_conv2d_bwd_data(const ss_conv_desc* d, const void* dy, const float* w, void* dx, int accumulate,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc_any(d)) return SS_ERR_INVALID;
    if (d->dtype == SS_DTYPE_F32) return conv2d_bwd_data32(d, (const float*)dy, w, (float*)dx, accumulate, ws, ws_bytes, stream);
    const bool fill_only = desc_wcache(d) && desc_wcache(d)->fill_only;
    if (!w || (!fill_only && (!dy || !dx))) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_conv2d_workspace_bytes(d, SS_PASS_BWD_DATA)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (fill_only) {
        ConvProb c = plain(d);
        c.dtype = d->dtype;
        if (!d->transposed && tconv_takes_dgrad(c, d->algo)) return SS_OK;
        const ConvShim sh = make_shim(d, ws, ws_bytes);
        return conv2d_bwd_data32(&sh.d32, nullptr, w, nullptr, accumulate, sh.ws, sh.ws_bytes, stream);
    }
    bool taken;
    int rc = native16_bwd_data(d, dy, w, dx, accumulate, ws, ws_bytes, s, &taken);
    if (taken) return rc;
    const ConvShim sh = make_shim(d, ws, ws_bytes);
    const long xr = (long)d->n * d->ih * d->iw, yr = (long)d->n * d->oh * d->ow;
    {   // reflection-padded C -> 1 head: dy (one channel) staged, dx written as stored by the folded matrix-core kernel
        ss_conv_desc dm;
        if (c1_typed_dgrad(d, sh, dx, accumulate, &dm)) {
            rc = ss_convert_launch(dy, d->dtype, d->out_cstride, sh.b, SS_DTYPE_F32, 1, yr, 1, s);
            if (rc != SS_OK) return rc;
            return conv2d_bwd_data32(&dm, sh.b, w, (float*)dx, accumulate, sh.ws, sh.ws_bytes, stream, d->dtype);
        }
    }
    rc = ss_convert_launch(dy, d->dtype, d->out_cstride, sh.b, SS_DTYPE_F32, d->cout, yr, d->cout, s);
    if (rc != SS_OK) return rc;
    if (accumulate) {
        rc = ss_convert_launch(dx, d->dtype, d->in_cstride, sh.a, SS_DTYPE_F32, d->cin, xr, d->cin, s);
        if (rc != SS_OK) return rc;
    }
    rc = conv2d_bwd_data32(&sh.d32, sh.b, w, sh.a, accumulate, sh.ws, sh.ws_bytes, stream);
    if (rc != SS_OK) return rc;
    return ss_convert_launch(sh.a, SS_DTYPE_F32, d->cin, dx, d->dtype, d->in_cstride, xr, d->cin, s);
}

int ss_conv2d_bwd_weight(const ss_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                         int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc_any(d)) return SS_ERR_INVALID;
    if (d->dtype == SS_DTYPE_F32) return conv2d_bwd_weight32(d, (const float*)x, (const float*)dy, dw, dbias, accumulate, ws, ws_bytes, stream);
    if (!x || !dy || !dw) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_conv2d_workspace_bytes(d, SS_PASS_BWD_WEIGHT)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    bool taken;
    int rc = native16_bwd_weight(d, x, dy, dw, dbias, accumulate, ws, ws_bytes, s, &taken);
    if (taken) return rc;
    const ConvShim sh = make_shim(d, ws, ws_bytes);
    {   // one-channel layers: the multi-channel operand is read as stored
        ss_conv_desc dm;
        const int side = c1_typed_wgrad(d, sh, x, dy, dbias, &dm);
        if (side == 1) {
            rc = ss_convert_launch(x, d->dtype, d->in_cstride, sh.a, SS_DTYPE_F32, 1, (long)d->n * d->ih * d->iw, 1, s);
            if (rc != SS_OK) return rc;
            return conv2d_bwd_weight32(&dm, sh.a, (const float*)dy, dw, dbias, accumulate, sh.ws, sh.ws_bytes, stream, d->dtype);
        }
        if (side == 2) {
            rc = ss_convert_launch(dy, d->dtype, d->out_cstride, sh.b, SS_DTYPE_F32, 1, (long)d->n * d->oh * d->ow, 1, s);
            if (rc != SS_OK) return rc;
            return conv2d_bwd_weight32(&dm, (const float*)x, sh.b, dw, dbias, accumulate, sh.ws, sh.ws_bytes, stream, d->dtype);
        }
    }
    rc = ss_convert_launch(x, d->dtype, d->in_cstride, sh.a, SS_DTYPE_F32, d->cin, (long)d->n * d->ih * d->iw, d->cin, s);
    if (rc != SS_OK) return rc;
    rc = ss_convert_launch(dy, d->dtype, d->out_cstride, sh.b, SS_DTYPE_F32, d->cout, (long)d->n * d->oh * d->ow, d->cout, s);
    if (rc != SS_OK) return rc;
    return conv2d_bwd_weight32(&sh.d32, sh.a, sh.b, dw, dbias, accumulate, sh.ws, sh.ws_bytes, stream);
}

}  // extern "C"
