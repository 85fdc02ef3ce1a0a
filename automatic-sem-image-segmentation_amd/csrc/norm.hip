// InstanceNorm (GroupNormalization groups=-1) and BatchNorm, forward / inference / backward, NHWC views.
// A tensor is seen as [G groups][P pixels][C channels]: G = n for instance norm, G = 1 (P = n*h*w) for batch norm.
// HBM-bound: statistics pass (read x), apply pass (read x, write y); backward: statistics pass
// (read dy, x[, y]) and apply pass (read dy, x[, y], write dx[, dres]).
// Statistics: per-thread fp32 partial sums over a pixel chunk, wave/LDS reduction over the pixel lanes,
// per-chunk partials in the workspace, fp64 combine in the finalize kernel.
#include "common.h"
#include <stdlib.h>
#include <initializer_list>
#include <type_traits>

namespace {

// Pixel chunks per group: enough blocks to fill 256 CUs several times over even for batch norm (G = 1) with < 64 channels
// (one channel block): 128 chunks there meant 128 workgroups on 256 CUs.
constexpr int NORM_CHUNK_BLOCKS = 4096;
constexpr int FIN_CL = 8;          // finalize kernels: 8 channels x 32 chunk lanes per block ...
// ... fewer channels (more chunk lanes) per block while the grid would hold fewer than 32 blocks: a BatchNorm over 4..51 channels
// (one group, up to 2048 chunks) ran on 1-7 blocks, each lane walking 64 chunks in 8 dependent rounds of loads (17 us per launch)
inline int fin_cl(int C, int G) {
    int cl = FIN_CL;
    while (cl > 1 && (long)((C + cl - 1) / cl) * G < 32) cl >>= 1;
    return cl;
}
inline int norm_max_chunks(int groups) {
    int m = NORM_CHUNK_BLOCKS / (groups > 0 ? groups : 1);
    return m < 128 ? 128 : (m > 2048 ? 2048 : m);
}

struct NormGeom {
    int G, C, CT, PT, chunks, cblocks;   // CT channel lanes (pow2 <= 64, each V channels wide), PT = 256/CT pixel lanes
    long P, pix_per_chunk;
};

NormGeom geom(const ss_norm_desc* d, int V = 1) {
    NormGeom g;
    g.G = d->groups;
    g.C = d->c;
    g.P = (long)d->n * d->h * d->w / d->groups;
    const int cv = (d->c + V - 1) / V;
    int ct = 1;
    while (ct < cv && ct < 64) ct <<= 1;
    if (cv < 64) ct = cv;          // exact channel lanes (odd MultiResUNet widths): every lane of a pixel row is useful
    g.CT = ct;
    g.PT = 256 / ct;
    g.cblocks = (cv + ct - 1) / ct;
    long chunks = (g.P + 63) / 64;
    long want = NORM_CHUNK_BLOCKS / ((long)g.G * g.cblocks);   // aim at >= ~4096 blocks in total
    if (want < 1) want = 1;
    if (chunks > want) chunks = want;
    if (chunks > norm_max_chunks(g.G)) chunks = norm_max_chunks(g.G);
    if (chunks < 1) chunks = 1;
    g.chunks = (int)chunks;
    g.pix_per_chunk = (g.P + chunks - 1) / chunks;
    return g;
}

// V-wide (1 or 4 channels per thread) global access helpers
template <int V> __device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
    if (V == 4) { const f32x4 t = *(const f32x4*)p; o[0] = t[0]; o[1 % V] = t[1]; o[2 % V] = t[2]; o[3 % V] = t[3]; }
    else o[0] = *p;
}
template <int V> __device__ __forceinline__ void stv(float* p, const float (&o)[V]) {
    if (V == 4) { f32x4 t = {o[0], o[1 % V], o[2 % V], o[3 % V]}; *(f32x4*)p = t; }
    else *p = o[0];
}
// 16-bit activation storage (ss_dtype F16 / BF16): four channels = one 8-byte access, arithmetic in fp32
typedef _Float16 nf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
template <int V> __device__ __forceinline__ void ldv(const _Float16* p, float (&o)[V]) {
    if (V == 4) { const f32x4 t = __builtin_convertvector(*(const nf16x4*)p, f32x4); o[0] = t[0]; o[1 % V] = t[1]; o[2 % V] = t[2]; o[3 % V] = t[3]; }
    else o[0] = (float)*p;
}
template <int V> __device__ __forceinline__ void stv(_Float16* p, const float (&o)[V]) {
    if (V == 4) { f32x4 t = {o[0], o[1 % V], o[2 % V], o[3 % V]}; *(nf16x4*)p = __builtin_convertvector(t, nf16x4); }
    else *p = (_Float16)o[0];
}
template <int V> __device__ __forceinline__ void ldv(const __bf16* p, float (&o)[V]) {
    if (V == 4) { const f32x4 t = __builtin_convertvector(*(const nbf16x4*)p, f32x4); o[0] = t[0]; o[1 % V] = t[1]; o[2 % V] = t[2]; o[3 % V] = t[3]; }
    else o[0] = (float)*p;
}
template <int V> __device__ __forceinline__ void stv(__bf16* p, const float (&o)[V]) {
    if (V == 4) { f32x4 t = {o[0], o[1 % V], o[2 % V], o[3 % V]}; *(nbf16x4*)p = __builtin_convertvector(t, nbf16x4); }
    else *p = (__bf16)o[0];
}

// non-temporal forms (measurement, norm_order bits 3 / 4): streaming hints for tensors a pass touches once
template <int V, typename T> __device__ __forceinline__ void ldv_nt(const T* p, float (&o)[V], bool nt) {
    if (nt && std::is_same<T, float>::value) {
        if (V == 4) { const f32x4 t = __builtin_nontemporal_load((const f32x4*)p); o[0] = t[0]; o[1 % V] = t[1]; o[2 % V] = t[2]; o[3 % V] = t[3]; }
        else o[0] = __builtin_nontemporal_load((const float*)p);
    } else ldv<V>(p, o);
}
template <int V, typename T> __device__ __forceinline__ void stv_nt(T* p, const float (&o)[V], bool nt) {
    if (nt && std::is_same<T, float>::value) {
        if (V == 4) { f32x4 t = {o[0], o[1 % V], o[2 % V], o[3 % V]}; __builtin_nontemporal_store(t, (f32x4*)p); }
        else __builtin_nontemporal_store(o[0], (float*)p);
    } else stv<V>(p, o);
}

// partial sums: part[((g*chunks + chunk)*C + c)*2 + {0,1}]
// MODE 0: (sum x, sum x^2)        MODE 1: (sum g, sum g*xhat) with g = dy * act'(y)
// Thread = V consecutive channels x a strided set of pixels; CT = channel lanes (in units of V), PT = 256/CT.
// ACC64 (the fused-finalize geometry: few, long chunks): the block-level tree runs in fp64 and the partial is STORED as fp64 -- with
// <= 128 chunks per group the rounding of every fp32 partial (2^-24 relative, no longer averaged over hundreds of chunks) showed up
// as 3 - 4 x the error of E[x^2] - E[x]^2 on low-variance channels (tools/norm_fuse_diag.py); `part` is then double[...].
template <typename T, int MODE, int V, bool ACC64 = false>
__global__ __launch_bounds__(256) void norm_stats_kernel(const T* __restrict__ x, int x_cs,
                                                         const T* __restrict__ dy, int dy_cs,
                                                         const T* __restrict__ y, int y_cs,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         int act, float alpha,
                                                         int C, long P, long pix_per_chunk, int CT, int PT,
                                                         float* __restrict__ part,
                                                         const float* __restrict__ rgamma = nullptr, const float* __restrict__ rbeta = nullptr,
                                                         int order = 0, float* __restrict__ pivot = nullptr) {
    // MODE 1 with y == nullptr (relu / leaky relu without residual): the activation mask is recomputed from x with the forward's
    // expression (y > 0 <=> t > 0) -- one tensor read less in each of the two backward passes
    typedef typename std::conditional<ACC64, double, float>::type RT;
    __shared__ RT red[2 * V][256];
    const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;      // threads with pt >= PT (256 % CT leftovers) idle
    const int c = (blockIdx.y * CT + ct) * V;
    // order & 1: the workgroups walk the tensor from its END (last group, last chunk first): what the kernel before this one wrote or
    // read last is what the 256 MiB Infinity Cache still holds.  Same partials in the same slots: bit-identical results.
    const int g = (order & 1) ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int bx = (order & 1) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const long p0 = (long)bx * pix_per_chunk;
    const long p1 = (p0 + pix_per_chunk < P) ? p0 + pix_per_chunk : P;
    float s1[V], s2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { s1[v] = 0.f; s2[v] = 0.f; }
    if (c < C && pt < PT) {
        const long base = (long)g * P;
        float mu[V], rs[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { mu[v] = 0.f; rs[v] = 0.f; }
        float gm[V], bt[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { gm[v] = 1.f; bt[v] = 0.f; }
        if (MODE == 1) { ldv<V>(mean + (long)g * C + c, mu); ldv<V>(rstd + (long)g * C + c, rs); }
        // ACC64 forward statistics are taken of x - x0 with x0 = the group's FIRST pixel of the channel (pivot[g][c], published by chunk 0):
        // a channel that is constant over the group -- a constant input image stays constant through the whole network -- then has the
        // sums 0 and 0 EXACTLY, i.e. mean = x0 and variance 0 whatever the summation order (plain sums of identical values round at
        // 3x, 5x, ...; the ReLU behind beta = 0 turns that rounding's sign into a whole channel's mask), and E[d^2] - E[d]^2 loses
        // nothing to cancellation when |mean| >> the spread
        float x0[V];
#pragma unroll
        for (int v = 0; v < V; ++v) x0[v] = 0.f;
        if (ACC64 && MODE == 0 && pivot) {
            ldv<V>(x + base * x_cs + c, x0);
            if (bx == 0 && pt == 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) pivot[(long)g * C + c + v] = x0[v];
            }
        }
        const bool recompute = MODE == 1 && act != SS_ACT_NONE && y == nullptr;
        if (recompute) { if (rgamma) ldv<V>(rgamma + c, gm); ldv<V>(rbeta + c, bt); }
        const bool relu = act == SS_ACT_RELU;
        const float slope = act == SS_ACT_LRELU ? alpha : 1.f;
        float kk[V];
#pragma unroll
        for (int v = 0; v < V; ++v) kk[v] = rs[v] * gm[v];
        // MASK (MODE 1): 0 no activation, 1 mask recomputed from x (relu / leaky relu), 2 from y (piecewise linear), 3 from y (tanh /
        // sigmoid): the loop is instantiated per mask source, no per-element case analysis inside (same sums in the same order)
        auto body = [&](auto MASKc) __attribute__((always_inline)) {
            constexpr int MASK = decltype(MASKc)::value;
#pragma unroll 4
            for (long p = p0 + pt; p < p1; p += PT) {
                float xv[V];
                ldv<V>(x + (base + p) * x_cs + c, xv);
                if (MODE == 0) {
#pragma unroll
                    for (int v = 0; v < V; ++v) { const float dv = ACC64 ? xv[v] - x0[v] : xv[v]; s1[v] += dv; s2[v] = fmaf(dv, dv, s2[v]); }
                } else {
                    float gv[V];
                    ldv<V>(dy + (base + p) * dy_cs + c, gv);
                    if (MASK == 1) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            // the forward's expression, bit for bit (norm_apply_kernel / the fused operand loads: explicit fma)
                            const float t = __builtin_fmaf(xv[v] - mu[v], kk[v], bt[v]);
                            gv[v] *= ss_act_grad_pwl(t, relu, slope);          // relu / lrelu: depends on the sign only
                        }
                    } else if (MASK >= 2) {
                        float yv[V];
                        ldv<V>(y + (base + p) * y_cs + c, yv);
#pragma unroll
                        for (int v = 0; v < V; ++v) gv[v] *= MASK == 2 ? ss_act_grad_pwl(yv[v], relu, slope) : ss_act_grad_from_out(yv[v], act, alpha);
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) { s1[v] += gv[v]; s2[v] = fmaf(gv[v], (xv[v] - mu[v]) * rs[v], s2[v]); }
                }
            }
        };
        if (MODE == 0 || act == SS_ACT_NONE) body(std::integral_constant<int, 0>{});
        else if (recompute) body(std::integral_constant<int, 1>{});
        else if (ss_act_is_pwl(act)) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 3>{});
    }
#pragma unroll
    for (int v = 0; v < V; ++v) { red[2 * v][threadIdx.x] = s1[v]; red[2 * v + 1][threadIdx.x] = s2[v]; }
    __syncthreads();
    if ((PT & (PT - 1)) == 0) {        // power-of-two pixel lanes: tree
        for (int off = PT / 2; off >= 1; off >>= 1) {
            if (pt < off) {
#pragma unroll
                for (int k = 0; k < 2 * V; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off * CT];
            }
            __syncthreads();
        }
    } else if (pt == 0) {              // odd lane count: fixed-order serial sum (deterministic)
        for (int q = 1; q < PT; ++q)
#pragma unroll
            for (int k = 0; k < 2 * V; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + q * CT];
    }
    if (pt == 0 && c < C) {
        RT* o = (RT*)part + (((long)g * gridDim.x + bx) * C + c) * 2;
#pragma unroll
        for (int v = 0; v < V; ++v) { o[2 * v] = red[2 * v][threadIdx.x]; o[2 * v + 1] = red[2 * v + 1][threadIdx.x]; }
    }
}

// mean / rstd per (g,c); optional moving-average update (batch norm, G == 1).
// Block = 32 channels x 8 chunk lanes (fixed-order LDS combine -> deterministic); grid = (C/32, G).
__global__ __launch_bounds__(256) void norm_finalize_fwd(const float* __restrict__ part, int chunks, int G, int C, long P, float eps,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ mm, float* __restrict__ mv, float momentum, int CL) {
    __shared__ double red[2][256];
    const int KL = 256 / CL;                      // CL channels x KL chunk lanes per block (CL = 32, or 8 for narrow tensors)
    const int cl = threadIdx.x % CL, kl = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl, g = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        // 8 loads in flight per thread (the loop is latency-bound: 23 us for a 2 MB partials array when rolled); same k order
        for (int k = kl; k < chunks; k += 8 * KL) {
            float a0[8], a1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u * KL;
                const float2 t = kk < chunks ? *(const float2*)(part + (((long)g * chunks + kk) * C + c) * 2) : make_float2(0.f, 0.f);
                a0[u] = t.x;
                a1[u] = t.y;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += a0[u]; s2 += a1[u]; }
        }
    }
    red[0][kl * CL + cl] = s1;
    red[1][kl * CL + cl] = s2;
    __syncthreads();
    if (kl == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int k = 0; k < KL; ++k) { s1 += red[0][k * CL + cl]; s2 += red[1][k * CL + cl]; }
        const long i = (long)g * C + c;
        const double mu = s1 / (double)P;
        double var = s2 / (double)P - mu * mu;   // E[x^2] - E[x]^2 (keras.ops.moments, torch backend)
        if (var < 0.0) var = 0.0;
        mean[i] = (float)mu;
        rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
        if (mm) {
            mm[c] = mm[c] * momentum + (float)mu * (1.f - momentum);
            mv[c] = mv[c] * momentum + (float)var * (1.f - momentum);
        }
    }
}

// ---- consumer-side finalize (ss_config "norm_fuse_fin") ---------------------------------------------------------------------------
// On mid-size tensors (per-GPU batch 1 - 2, 256 x 256 tiles) a norm pass is three ~10 us launches and the middle one -- the
// finalize -- does almost nothing; a dependent dispatch costs ~10 us whoever issues it.  In fused mode the APPLY kernels reduce the
// statistics partials of THEIR OWN channel block in a prologue: fixed order, fp64, no atomics, no cross-workgroup ordering (the
// partials were written by the previous launch).  Every workgroup of a (group, channel block) repeats the same small sum (<= 128
// chunks x <= 32 channels x 8 bytes = 32 KB from L2), so the geometry is chosen for it: narrow channel blocks, few chunks.  The
// workgroup with blockIdx.x == 0 also publishes mean / rstd (the backward pass and deferred consumers read them), the BatchNorm
// moving averages and, in backward, the parameter gradients (summed over the groups in group order).
constexpr long NORM_FUSE_ELEMS = 8L << 20;
constexpr int NORM_FUSE_CH = 32, NORM_FUSE_MAX_CHUNKS = 128, NORM_FUSE_EXT_CHUNKS = 256;

struct NormFin {
    const float* part;          // [G][chunks][C][2] partial sums; nullptr: not fused (mean / rstd / sums come finalized)
    int part64;                 // the partials are fp64 (norm_stats_kernel<..., ACC64>); 0: fp32 (a convolution epilogue's)
    const float* pivot;         // forward, fp64 partials: they are sums of x - pivot[g][c] (nullptr: of x)
    int chunks, G;
    double P;                   // elements per (group, channel)
    float eps, momentum;
    float* mean; float* rstd;   // forward: published by the blockIdx.x == 0 workgroups
    float* mm; float* mv;       // forward, BatchNorm: moving averages (G == 1)
    float* dgamma; float* dbeta; int acc_params;          // backward: parameter gradients
};

// Totals (sum, sum2) of group `g` for the block's channels [c0, c0 + nch): thread t < nch returns them; red = LDS, 512 doubles.
__device__ __forceinline__ void fin_block_totals(const float* __restrict__ part, int part64, int chunks, int g, int C, int c0, int nch,
                                                 double* red /* LDS, 512 doubles */, double& T1, double& T2) {
    const int KL = 256 / nch;
    const int j = threadIdx.x % nch, kl = threadIdx.x / nch;
    double s1 = 0.0, s2 = 0.0;
    if (kl < KL && c0 + j < C) {
        const long b0 = (((long)g * chunks) * C + c0 + j) * 2;
        for (int k = kl; k < chunks; k += 4 * KL) {          // four loads in flight, same k order
            double a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k + u * KL;
                a0[u] = 0.0; a1[u] = 0.0;
                if (kk < chunks) {
                    if (part64) { const double2 t = *(const double2*)((const double*)part + b0 + (long)kk * C * 2); a0[u] = t.x; a1[u] = t.y; }
                    else { const float2 t = *(const float2*)(part + b0 + (long)kk * C * 2); a0[u] = t.x; a1[u] = t.y; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += a0[u]; s2 += a1[u]; }
        }
    }
    __syncthreads();          // (a previous use of red is over)
    if (kl < KL) { red[kl * nch + j] = s1; red[256 + kl * nch + j] = s2; }
    __syncthreads();
    T1 = 0.0; T2 = 0.0;
    if (threadIdx.x < nch)
        for (int k = 0; k < KL; ++k) { T1 += red[k * nch + threadIdx.x]; T2 += red[256 + k * nch + threadIdx.x]; }
}

// mean / rstd of group `g` for the block's channels [c0, c0 + nch) into fstat[0 / 1][channel - c0]; `publish`: also to global memory
// (+ the BatchNorm moving averages).  norm_finalize_fwd's arithmetic.  Ends with a barrier: fstat is readable by every thread.
__device__ __forceinline__ void fin_fwd_stats(const NormFin& fin, int g, int C, int c0, int nch, double* fred, float (*fstat)[NORM_FUSE_CH],
                                              bool publish) {
    double T1, T2;
    fin_block_totals(fin.part, fin.part64, fin.chunks, g, C, c0, nch, fred, T1, T2);
    if ((int)threadIdx.x < nch && c0 + (int)threadIdx.x < C) {
        const int cc = c0 + threadIdx.x;
        const double md = T1 / fin.P;                                                     // mean of x - pivot
        const double m = (fin.pivot ? (double)fin.pivot[(long)g * C + cc] : 0.0) + md;
        double var = T2 / fin.P - md * md;          // E[d^2] - E[d]^2 = E[x^2] - E[x]^2 (keras.ops.moments, torch backend)
        if (var < 0.0) var = 0.0;
        const float muf = (float)m, rsf = (float)(1.0 / sqrt(var + (double)fin.eps));
        fstat[0][threadIdx.x] = muf;
        fstat[1][threadIdx.x] = rsf;
        if (publish) {
            fin.mean[(long)g * C + cc] = muf;
            fin.rstd[(long)g * C + cc] = rsf;
            if (fin.mm) {
                fin.mm[cc] = fin.mm[cc] * fin.momentum + muf * (1.f - fin.momentum);
                fin.mv[cc] = fin.mv[cc] * fin.momentum + (float)var * (1.f - fin.momentum);
            }
        }
    }
    __syncthreads();
}
// statistics-only calls (the apply pass belongs to a consuming convolution, engine.DeferredNorm) on tensors the fused form takes: the
// SAME reduction as the apply kernels' prologue, so a deferred and a materialised norm see the same mean / rstd bits.  grid (channel blocks, G)
__global__ __launch_bounds__(256) void norm_finalize_fused_kernel(NormFin fin, int C, int nch) {
    __shared__ double fred[512];
    __shared__ float fstat[2][NORM_FUSE_CH];
    fin_fwd_stats(fin, blockIdx.y, C, blockIdx.x * nch, nch, fred, fstat, true);
}

// y = act((x-mean)*rstd*gamma + beta + residual)
// CHANNEL-STATIONARY threads: thread = V consecutive channels (lane ct of CT) x a strided set of the rows of ONE group; grid =
// (row chunks, channel blocks, groups).  The per-channel operands (mean, rstd * gamma, beta) are loaded once and stay in registers:
// the flat grid-strided form re-gathered them for every element (5 loads + 1 store per vector: the kernel was paced by the
// vector-memory instruction rate, 3.7 TB/s on the trunk tensors and 1.7 TB/s on the odd-width MultiResUNet tensors), four rows are
// in flight per thread.  Row chunks are handed out from the END (block 0 takes the last one): the statistics pass before this
// kernel walked the tensor front to back, so its tail is what the 256 MiB Infinity Cache still holds.
template <typename T, int V, bool FIN = false>
__global__ __launch_bounds__(256) void norm_apply_kernel(const T* __restrict__ x, int x_cs,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const T* __restrict__ res, int res_cs,
                                                         T* __restrict__ y, int y_cs,
                                                         int act, float alpha, int C, long P, int CT, int PT, long rows_per_chunk,
                                                         unsigned int* __restrict__ amax = nullptr, int order = 0,
                                                         NormFin fin = NormFin{}) {
    const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
    const int c = (blockIdx.y * CT + ct) * V;
    const int g = (order & 2) ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;          // (order & 2: the groups from the end as well)
    const long p0 = (long)(gridDim.x - 1 - blockIdx.x) * rows_per_chunk;
    const long p1 = p0 + rows_per_chunk < P ? p0 + rows_per_chunk : P;
    float am = 0.f;
    __shared__ double fred[FIN ? 512 : 1];
    __shared__ float fstat[2][NORM_FUSE_CH];
    if constexpr (FIN) {          // consumer-side finalize: mean / rstd of this block's channels from the statistics partials
        fin_fwd_stats(fin, g, C, blockIdx.y * CT * V, CT * V, fred, fstat, blockIdx.x == 0);
    }
    if (c < C && pt < PT) {
        const long gi = (long)g * C + c;
        float mu[V], kk[V], bt[V], gm[V];
        if (FIN) {
#pragma unroll
            for (int v = 0; v < V; ++v) { mu[v] = fstat[0][ct * V + v]; kk[v] = fstat[1][ct * V + v]; }
        } else {
            ldv<V>(mean + gi, mu);
            ldv<V>(rstd + gi, kk);
        }
        ldv<V>(beta + c, bt);
        if (gamma) {
            ldv<V>(gamma + c, gm);
#pragma unroll
            for (int v = 0; v < V; ++v) kk[v] *= gm[v];
        }
        const T* const xb = x + (long)g * P * x_cs + c;
        const T* const rb = res ? res + (long)g * P * res_cs + c : nullptr;
        T* const yb = y + (long)g * P * y_cs + c;
        const bool relu = act == SS_ACT_RELU;
        const float slope = act == SS_ACT_LRELU ? alpha : 1.f;
        // the loop, instantiated per (piecewise-linear activation?, residual?): no per-element case analysis inside
        auto body = [&](auto PWLc, auto RESc) __attribute__((always_inline)) {
            constexpr bool PWL = decltype(PWLc)::value, RES = decltype(RESc)::value;
            for (long p = p0 + pt; p < p1; p += 4L * PT) {
                float xv[4][V], rv[4][V];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = p + (long)u * PT;
                    if (q < p1) {
                        ldv_nt<V>(xb + q * x_cs, xv[u], (order & 8) != 0);
                        if (RES) ldv_nt<V>(rb + q * res_cs, rv[u], (order & 8) != 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = p + (long)u * PT;
                    if (q < p1) {
                        float o[V];
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            // explicit fma: a convolution that normalises in its operand load (conv_wino.hip, ss_conv_desc::in_norm_*) forms the
                            // same expression and must produce the same bits
                            float t = __builtin_fmaf(xv[u][v] - mu[v], kk[v], bt[v]);
                            if (RES) t += rv[u][v];
                            o[v] = PWL ? ss_act_pwl(t, relu, slope) : ss_apply_act(t, act, alpha);
                            am = fmaxf(am, fabsf((float)(T)o[v]));          // what is stored: rounded to the storage type
                        }
                        stv_nt<V>(yb + q * y_cs, o, (order & 16) != 0);
                    }
                }
            }
        };
        if (ss_act_is_pwl(act)) {
            if (res) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{});
        } else {
            if (res) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{});
        }
    }
    if (amax) ss_block_amax_to_slot(am, amax);
}

// inference: statistics from moving mean / variance
template <typename T>
__global__ __launch_bounds__(256) void norm_infer_kernel(const T* __restrict__ x, int x_cs,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mm, const float* __restrict__ mv, float eps,
                                                         const T* __restrict__ res, int res_cs,
                                                         T* __restrict__ y, int y_cs, int act, float alpha, int C, long rows) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const int c = (int)(e % C);
    const long row = e / C;
    const float sc = rsqrtf(mv[c] + eps) * (gamma ? gamma[c] : 1.f);
    float v = ((float)x[row * x_cs + c] - mm[c]) * sc + beta[c];
    if (res) v += (float)res[row * res_cs + c];
    y[row * y_cs + c] = (T)ss_apply_act(v, act, alpha);
}

// backward finalize: per (g,c) means of g and g*xhat -> sums array, and the raw per-group totals (double) -> rt.
// Block = CL channels x 256/CL chunk lanes of ONE group, grid (C/CL, G) (it walked the groups one after the other on C/CL blocks:
// a chain of G memory round trips, 23 us for the 8 instances of a trunk layer); every combine is a fixed-order LDS sum.  dgamma /
// dbeta = the totals summed over the groups in group order: norm_bwd_apply_kernel does that from rt (same values as before).
__global__ __launch_bounds__(256) void norm_finalize_bwd(const float* __restrict__ part, int chunks, int G, int C, long P,
                                                         float* __restrict__ sums /* [G*C*2] */, double* __restrict__ rt /* [G*C*2] */, int CL) {
    __shared__ double red[2][256];
    const int KL = 256 / CL;
    const int cl = threadIdx.x % CL, kl = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl, g = blockIdx.y;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int k = kl; k < chunks; k += 8 * KL) {          // 8 loads in flight, same k order (see norm_finalize_fwd)
            float a0[8], a1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + u * KL;
                const float2 t = kk < chunks ? *(const float2*)(part + (((long)g * chunks + kk) * C + c) * 2) : make_float2(0.f, 0.f);
                a0[u] = t.x;
                a1[u] = t.y;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += a0[u]; s2 += a1[u]; }
        }
    }
    red[0][kl * CL + cl] = s1;
    red[1][kl * CL + cl] = s2;
    __syncthreads();
    if (kl == 0 && c < C) {
        s1 = 0.0; s2 = 0.0;
        for (int k = 0; k < KL; ++k) { s1 += red[0][k * CL + cl]; s2 += red[1][k * CL + cl]; }
        const long i = ((long)g * C + c) * 2;
        sums[i + 0] = (float)(s1 / (double)P);
        sums[i + 1] = (float)(s2 / (double)P);
        rt[i + 0] = s1;
        rt[i + 1] = s2;
    }
}

// dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) ; dres = g.  Channel-stationary threads as norm_apply_kernel: the seven
// per-channel operands live in registers (the flat form issued 8 + 2 V loads per vector for them).
template <typename T, int V, bool FIN = false>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ dy, int dy_cs,
                                                             const T* __restrict__ x, int x_cs,
                                                             const T* __restrict__ y, int y_cs,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ sums,
                                                             T* __restrict__ dx, int dx_cs, int acc_dx,
                                                             T* __restrict__ dres, int dres_cs, int acc_dres,
                                                             int act, float alpha, int C, long P, int CT, int PT, long rows_per_chunk,
                                                             const float* __restrict__ rbeta = nullptr, unsigned int* __restrict__ amax = nullptr,
                                                             const double* __restrict__ rt = nullptr, int G = 0,
                                                             float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr,
                                                             int acc_params = 0, int order = 0, NormFin fin = NormFin{}) {
    float am = 0.f;
    if (rt) {          // parameter gradients: the per-group totals of norm_finalize_bwd summed in group order (one thread per channel)
        const long nblk = (long)gridDim.x * gridDim.y * gridDim.z;
        const long blk = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
        for (long c = blk * blockDim.x + threadIdx.x; c < C; c += nblk * blockDim.x) {
            double tg = 0.0, tgx = 0.0;
            for (int g = 0; g < G; ++g) { tg += rt[((long)g * C + c) * 2]; tgx += rt[((long)g * C + c) * 2 + 1]; }
            if (dbeta) dbeta[c] = acc_params ? dbeta[c] + (float)tg : (float)tg;
            if (dgamma) dgamma[c] = acc_params ? dgamma[c] + (float)tgx : (float)tgx;
        }
    }
    const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
    const int c = (blockIdx.y * CT + ct) * V;
    // order & 4: backward apply walks FORWARD (first group, first chunk first) -- the counterpart of a statistics pass that walked from the end
    const int g = (order & 4) ? (int)blockIdx.z : ((order & 2) ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z);
    const long p0 = (long)((order & 4) ? blockIdx.x : gridDim.x - 1 - blockIdx.x) * rows_per_chunk;
    const long p1 = p0 + rows_per_chunk < P ? p0 + rows_per_chunk : P;
    __shared__ double fred[FIN ? 512 : 1];
    __shared__ float fstat[2][NORM_FUSE_CH];
    if constexpr (FIN) {          // consumer-side finalize: the means of g and g * xhat of this block's channels from the statistics partials
        const int c0 = blockIdx.y * CT * V, nch = CT * V;
        const bool mine = (int)threadIdx.x < nch && c0 + (int)threadIdx.x < C;
        double T1, T2;
        fin_block_totals(fin.part, fin.part64, fin.chunks, g, C, c0, nch, fred, T1, T2);
        if (mine) { fstat[0][threadIdx.x] = (float)(T1 / fin.P); fstat[1][threadIdx.x] = (float)(T2 / fin.P); }
        if (blockIdx.x == 0 && g == 0 && (fin.dgamma || fin.dbeta)) {
            // parameter gradients: the groups' raw totals summed in group order (norm_finalize_bwd + the rt loop of the unfused form)
            double tg = T1, tgx = T2;
            for (int gg = 1; gg < fin.G; ++gg) {
                double A, B;
                fin_block_totals(fin.part, fin.part64, fin.chunks, gg, C, c0, nch, fred, A, B);
                tg += A;
                tgx += B;
            }
            if (mine) {
                const int cc = c0 + threadIdx.x;
                if (fin.dbeta) fin.dbeta[cc] = fin.acc_params ? fin.dbeta[cc] + (float)tg : (float)tg;
                if (fin.dgamma) fin.dgamma[cc] = fin.acc_params ? fin.dgamma[cc] + (float)tgx : (float)tgx;
            }
        }
        __syncthreads();
    }
    if (c < C && pt < PT) {
        const long gi = (long)g * C + c;
        float mu[V], rs[V], gm[V], bt[V], sm[2 * V];
        ldv<V>(mean + gi, mu);
        ldv<V>(rstd + gi, rs);
#pragma unroll
        for (int v = 0; v < V; ++v) { gm[v] = 1.f; bt[v] = 0.f; }
        if (gamma) ldv<V>(gamma + c, gm);
        if (FIN) {
#pragma unroll
            for (int v = 0; v < V; ++v) { sm[2 * v] = fstat[0][ct * V + v]; sm[2 * v + 1] = fstat[1][ct * V + v]; }
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) { sm[2 * v] = sums[(gi + v) * 2]; sm[2 * v + 1] = sums[(gi + v) * 2 + 1]; }
        }
        const bool recompute = act != SS_ACT_NONE && y == nullptr;          // mask recomputed from x (see norm_stats_kernel)
        const bool from_y = act != SS_ACT_NONE && y != nullptr;
        if (recompute) ldv<V>(rbeta + c, bt);
        const T* const gb = dy + (long)g * P * dy_cs + c;
        const T* const xb = x + (long)g * P * x_cs + c;
        const T* const yb = from_y ? y + (long)g * P * y_cs + c : nullptr;
        T* const ob = dx + (long)g * P * dx_cs + c;
        T* const rb = dres ? dres + (long)g * P * dres_cs + c : nullptr;
        const bool relu = act == SS_ACT_RELU;
        const float slope = act == SS_ACT_LRELU ? alpha : 1.f;
        float kk[V];          // rstd * gamma: the forward's factor
#pragma unroll
        for (int v = 0; v < V; ++v) kk[v] = rs[v] * (gamma ? gm[v] : 1.f);
        // MASK: 0 no activation, 1 mask recomputed from x (relu / leaky relu), 2 from y (piecewise linear), 3 from y (tanh / sigmoid).
        // The loop is instantiated per mask source: no per-element case analysis inside
        auto body = [&](auto MASKc) __attribute__((always_inline)) {
            constexpr int MASK = decltype(MASKc)::value;
            for (long p = p0 + pt; p < p1; p += 4L * PT) {
                float gv[4][V], xv[4][V], yv[4][V], o[4][V], r[4][V];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = p + (long)u * PT;
                    if (q < p1) {
                        ldv_nt<V>(gb + q * dy_cs, gv[u], (order & 8) != 0);
                        ldv_nt<V>(xb + q * x_cs, xv[u], (order & 8) != 0);
                        if (MASK >= 2) ldv<V>(yb + q * y_cs, yv[u]);
                        if (acc_dx) ldv<V>(ob + q * dx_cs, o[u]);
                        if (dres && acc_dres) ldv<V>(rb + q * dres_cs, r[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long q = p + (long)u * PT;
                    if (q < p1) {
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            if (MASK == 1) {
                                // the forward's expression, bit for bit (norm_apply_kernel / the fused operand loads: explicit fma)
                                const float t = __builtin_fmaf(xv[u][v] - mu[v], kk[v], bt[v]);
                                gv[u][v] *= ss_act_grad_pwl(t, relu, slope);
                            } else if (MASK == 2) {
                                gv[u][v] *= ss_act_grad_pwl(yv[u][v], relu, slope);
                            } else if (MASK == 3) {
                                gv[u][v] *= ss_act_grad_from_out(yv[u][v], act, alpha);
                            }
                        }
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const float xh = (xv[u][v] - mu[v]) * rs[v];
                            const float dv = kk[v] * (gv[u][v] - sm[2 * v] - xh * sm[2 * v + 1]);
                            o[u][v] = acc_dx ? o[u][v] + dv : dv;
                            am = fmaxf(am, fabsf((float)(T)o[u][v]));          // what is stored: rounded to the storage type
                        }
                        stv_nt<V>(ob + q * dx_cs, o[u], (order & 16) != 0);
                        if (dres) {
                            if (acc_dres) {
#pragma unroll
                                for (int v = 0; v < V; ++v) r[u][v] += gv[u][v];
                                stv<V>(rb + q * dres_cs, r[u]);
                            } else {
                                stv<V>(rb + q * dres_cs, gv[u]);
                            }
                        }
                    }
                }
            }
        };
        if (recompute) body(std::integral_constant<int, 1>{});
        else if (!from_y) body(std::integral_constant<int, 0>{});
        else if (ss_act_is_pwl(act)) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 3>{});
    }
    if (amax) ss_block_amax_to_slot(am, amax);
}

// ---- InstanceNorm backward in ONE pass over the tensors (fp32, 32-channel blocks, <= 16384 pixels per sample: the residual trunk) ----
// The two-pass form reads dy and x twice (statistics, then apply: five tensor passes where three are needed, 838 MB of traffic per
// trunk layer where 536 MB would do).  Here the workgroups of one (sample, 32-channel block) -- a GROUP of NPIECE workgroups with
// consecutive block indices -- keep their share of dy and x IN REGISTERS (32 pixel lanes x 8 channel quads per workgroup, NIT pixels
// per thread: 2 x NIT float4), publish their partial sums, meet at a GROUP-LOCAL barrier (a counter in global memory), add the
// group's partials in piece order (fp64, the same in every workgroup: deterministic) and apply from the registers.
// MEASURED AND NOT THE DEFAULT (ss_config norm_bwd_resident = 1 switches it on; tools/norm_resident_probe.py): 229 us against the two
// passes' 105 us at [8, 128, 128, 256].  What a workgroup streams takes ~8 us (128 KB in, 64 KB out); what it waits for between the two
// halves is a chain of dependent device-scope round trips -- partial store (write-through), arrival atomic, polling, L2 invalidate,
// partial loads: ~12 us -- and the register file holds two such workgroups per CU, so the chain is exposed, not hidden.  (First
// version: plain stores + __threadfence(): an agent-scope release writes back the whole L2, full of other workgroups' dx: 604 us.)
// Progress: a group's workgroups are dispatched in index order, so at most ONE group per resident launch is partly on the chip; its
// members spin on <= NPIECE <= 32 of the 512 workgroup slots (two launches of this kind run side by side in the dual-chain step),
// every slot beside them belongs to fully dispatched groups, which finish.  A spinning workgroup gives up after ~2^24 polls (seconds)
// and counts it in g_norm_resident_timeouts (ss_norm_resident_timeouts): a wrong result that the tests see, never a hung GPU.
__device__ unsigned int g_norm_resident_timeouts = 0;

struct NormRes {
    double* part;            // [groups][NPIECE][32][2] partial (sum g, sum g * xhat) of every workgroup
    unsigned int* counter;   // [groups] arrivals (zero on entry), then [C / 32] tickets of the parameter-gradient pass (zero on entry)
    double* rt;              // [n][C][2] the groups' totals (parameter gradients)
    float* dgamma;
    float* dbeta;
    int acc_params, n, npiece;
};

template <int NIT>
__global__ __launch_bounds__(256, 2) void norm_bwd_resident_kernel(const float* __restrict__ dy, int dy_cs, const float* __restrict__ x, int x_cs,
                                                                   const float* __restrict__ y, int y_cs, const float* __restrict__ gamma,
                                                                   const float* __restrict__ rbeta, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* __restrict__ dx, int dx_cs,
                                                                   float* __restrict__ dres, int dres_cs, int act, float alpha, int C, int P,
                                                                   unsigned int* __restrict__ amax, NormRes r) {
    __shared__ double red[64][33];          // [channel of the block x {s1, s2}][pixel lane] (+1: no bank aliasing between lanes)
    __shared__ float stat[2][32];
    const int tid = threadIdx.x;
    const int cq = tid & 7, pl = tid >> 3;
    const int ncb = C / 32;
    const int grp = blockIdx.x / r.npiece, piece = blockIdx.x - grp * r.npiece;
    const int n = grp / ncb, cb = grp - n * ncb;
    const int c = cb * 32 + cq * 4;
    const long gi = (long)n * C + c;
    const f32x4 mu = *(const f32x4*)(mean + gi), rs = *(const f32x4*)(rstd + gi);
    f32x4 gm = {1.f, 1.f, 1.f, 1.f}, bt = {0.f, 0.f, 0.f, 0.f};
    if (gamma) gm = *(const f32x4*)(gamma + c);
    const bool recompute = act != SS_ACT_NONE && y == nullptr;          // mask recomputed from x with the forward's expression
    const bool from_y = act != SS_ACT_NONE && y != nullptr;
    if (recompute) bt = *(const f32x4*)(rbeta + c);
    const bool relu = act == SS_ACT_RELU;
    const float slope = act == SS_ACT_LRELU ? alpha : 1.f;
    f32x4 kk;
#pragma unroll
    for (int v = 0; v < 4; ++v) kk[v] = rs[v] * gm[v];
    const int p0 = piece * (32 * NIT) + pl;
    const float* const gb = dy + (long)n * P * dy_cs + c;
    const float* const xb = x + (long)n * P * x_cs + c;

    f32x4 gv[NIT], xv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = p0 + 32 * it;
        const bool ok = p < P;
        gv[it] = ok ? *(const f32x4*)(gb + (long)p * dy_cs) : f32x4{0.f, 0.f, 0.f, 0.f};
        xv[it] = ok ? *(const f32x4*)(xb + (long)p * x_cs) : mu;          // (xhat = 0 beyond the sample)
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (recompute) {
#pragma unroll
            for (int v = 0; v < 4; ++v) gv[it][v] *= ss_act_grad_pwl(__builtin_fmaf(xv[it][v] - mu[v], kk[v], bt[v]), relu, slope);
        } else if (from_y) {
            const int p = p0 + 32 * it;
            if (p < P) {
                const f32x4 yv = *(const f32x4*)(y + ((long)n * P + p) * y_cs + c);
#pragma unroll
                for (int v = 0; v < 4; ++v) gv[it][v] *= ss_act_grad_pwl(yv[v], relu, slope);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float xh = (xv[it][v] - mu[v]) * rs[v];
            xv[it][v] = xh;          // the registers keep g and xhat for the apply
            s1[v] += gv[it][v];
            s2[v] = fmaf(gv[it][v], xh, s2[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { red[(cq * 4 + v) * 2][pl] = (double)s1[v]; red[(cq * 4 + v) * 2 + 1][pl] = (double)s2[v]; }
    __syncthreads();
    double* const mypart = r.part + ((long)grp * r.npiece + piece) * 64;
    if (tid < 64) {          // fixed-order sum over the 32 pixel lanes
        double a = 0.0;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) a += red[tid][q];
        // a device-scope STORE (write-through past this XCD's L2), not a plain store + __threadfence(): an agent-scope release writes
        // back the whole L2 -- full of the other workgroups' dx lines -- and with every workgroup fencing twice the kernel ran at 0.65 TB/s
        __hip_atomic_store((unsigned long long*)mypart + tid, __builtin_bit_cast(unsigned long long, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the stores above have been acknowledged (vmcnt) before the arrival is counted
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(r.counter + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        while (__hip_atomic_load(r.counter + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r.npiece) {
            __builtin_amdgcn_s_sleep(8);
            if (++polls > (1u << 24)) { atomicAdd(&g_norm_resident_timeouts, 1u); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (plain loads behind it see what the arrivals released)
    // the group's totals: every thread fetches 8 pieces' worth of one (channel, sum) -- 32 x 64 independent loads in flight instead of a
    // chain of 32 device-scope loads per thread (~1.5 us each: that chain alone was 50 us per workgroup) --, then the pieces are added
    // in index order, the same sum in every workgroup of the group
    {
        const double* src = r.part + (long)grp * r.npiece * 64;
        const int t64 = tid & 63, kq = tid >> 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kq * 8 + j;
            red[t64][k] = k < r.npiece ? src[(long)k * 64 + t64] : 0.0;
        }
    }
    __syncthreads();
    if (tid < 64) {
        double a = 0.0;
        for (int k = 0; k < r.npiece; ++k) a += red[tid][k];
        stat[tid & 1][tid >> 1] = (float)(a / (double)P);
        if (piece == 0 && r.rt)
            __hip_atomic_store((unsigned long long*)(r.rt + ((long)n * C + cb * 32 + (tid >> 1)) * 2 + (tid & 1)), __builtin_bit_cast(unsigned long long, a),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    // parameter gradients: the piece-0 workgroup of the LAST group of this channel block to get here adds the samples' totals in sample order
    if (piece == 0 && r.rt && (r.dgamma || r.dbeta)) {
        __shared__ int last;
        if (tid == 0) last = __hip_atomic_fetch_add(r.counter + (long)gridDim.x / r.npiece + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(r.n - 1);
        __syncthreads();
        if (last && tid < 64) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int cc = cb * 32 + (tid >> 1), k = tid & 1;
            double t = 0.0;
            for (int gg = 0; gg < r.n; ++gg) t += r.rt[((long)gg * C + cc) * 2 + k];
            float* dst = k ? r.dgamma : r.dbeta;
            if (dst) dst[cc] = r.acc_params ? dst[cc] + (float)t : (float)t;
        }
    }
    float am = 0.f;
    float m1[4], m2[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { m1[v] = stat[0][cq * 4 + v]; m2[v] = stat[1][cq * 4 + v]; }
    float* const ob = dx + (long)n * P * dx_cs + c;
    float* const rb = dres ? dres + (long)n * P * dres_cs + c : nullptr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int p = p0 + 32 * it;
        if (p < P) {
            f32x4 o;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                o[v] = kk[v] * (gv[it][v] - m1[v] - xv[it][v] * m2[v]);
                am = fmaxf(am, fabsf(o[v]));
            }
            *(f32x4*)(ob + (long)p * dx_cs) = o;
            if (rb) *(f32x4*)(rb + (long)p * dres_cs) = gv[it];
        }
    }
    if (amax) ss_block_amax_to_slot(am, amax);
}

// sums[(g*C+c)*2+k] = sum over chunks of part (raw sums, fp64 combine)
__global__ __launch_bounds__(256) void norm_collapse_kernel(const float* __restrict__ part, int chunks, int G, int C, float* __restrict__ sums) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)G * C) return;
    const int g = (int)(i / C), c = (int)(i % C);
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const float* o = part + (((long)g * chunks + k) * C + c) * 2;
        s1 += o[0];
        s2 += o[1];
    }
    sums[i * 2] = (float)s1;
    sums[i * 2 + 1] = (float)s2;
}

// cross-rank backward finalize: means from the GLOBAL raw sums / global count, parameter gradients from the LOCAL sums
__global__ __launch_bounds__(256) void norm_finalize_bwd_sync(const float* __restrict__ gsums, const float* __restrict__ lsums, int G, int C,
                                                              double count, float* __restrict__ means, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double tg = 0.0, tgx = 0.0;
    for (int g = 0; g < G; ++g) {
        const long i = ((long)g * C + c) * 2;
        means[i] = (float)((double)gsums[i] / count);
        means[i + 1] = (float)((double)gsums[i + 1] / count);
        tg += lsums[i];
        tgx += lsums[i + 1];
    }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)tg : (float)tg;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)tgx : (float)tgx;
}

// ---- small tensors: statistics + finalize + apply in ONE launch --------------------------------------------------------------
// At per-GPU batch 1-2 a norm layer is three ~10 us launches whose cost is launch latency, not bandwidth.  Here a block owns
// CL*V channels of ONE group and walks that group's pixels twice (statistics, then apply: the second walk hits L2); groups are
// independent blocks in forward, sequential inside the block in backward (dgamma / dbeta are sums over the groups).
// Block = CL channel lanes x 256/CL pixel lanes; fp32 lane partials, fixed-order tree over the pixel lanes, fp64 finalize.
constexpr long NORM_SMALL_ELEMS = 4L << 20;

template <typename T, int V>
__global__ __launch_bounds__(256) void norm_small_fwd_kernel(const T* __restrict__ x, int x_cs,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const T* __restrict__ res, int res_cs,
                                                             T* __restrict__ y, int y_cs,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ mm, float* __restrict__ mv, float momentum, float eps,
                                                             int act, float alpha, int C, long P, int CL) {
    __shared__ float red[2 * V][256];
    __shared__ float stat[2 * V][64];
    const int PT = 256 / CL;
    const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
    const int c = (blockIdx.x * CL + cl) * V;
    const int g = blockIdx.y;
    const bool cval = c < C;
    const long base = (long)g * P;
    float s1[V], s2[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { s1[v] = 0.f; s2[v] = 0.f; }
    if (cval) {
#pragma unroll 4
        for (long p = pl; p < P; p += PT) {
            float xv[V];
            ldv<V>(x + (base + p) * x_cs + c, xv);
#pragma unroll
            for (int v = 0; v < V; ++v) { s1[v] += xv[v]; s2[v] = fmaf(xv[v], xv[v], s2[v]); }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) { red[2 * v][threadIdx.x] = s1[v]; red[2 * v + 1][threadIdx.x] = s2[v]; }
    __syncthreads();
    for (int off = PT / 2; off >= 1; off >>= 1) {        // PT is a power of two
        if (pl < off) {
#pragma unroll
            for (int k = 0; k < 2 * V; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off * CL];
        }
        __syncthreads();
    }
    if (pl == 0 && cval) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const double mu = (double)red[2 * v][threadIdx.x] / (double)P;
            double var = (double)red[2 * v + 1][threadIdx.x] / (double)P - mu * mu;
            if (var < 0.0) var = 0.0;
            const float muf = (float)mu, rsf = (float)(1.0 / sqrt(var + (double)eps));
            const long i = (long)g * C + c + v;
            mean[i] = muf;
            rstd[i] = rsf;
            if (mm) {
                mm[c + v] = mm[c + v] * momentum + muf * (1.f - momentum);
                mv[c + v] = mv[c + v] * momentum + (float)var * (1.f - momentum);
            }
            stat[2 * v][cl] = muf;
            stat[2 * v + 1][cl] = rsf;
        }
    }
    __syncthreads();
    if (!cval) return;
    float mu[V], sc[V], bt[V];
    {
        float gm[V];
#pragma unroll
        for (int v = 0; v < V; ++v) gm[v] = 1.f;
        if (gamma) ldv<V>(gamma + c, gm);
        ldv<V>(beta + c, bt);
#pragma unroll
        for (int v = 0; v < V; ++v) { mu[v] = stat[2 * v][cl]; sc[v] = stat[2 * v + 1][cl] * gm[v]; }
    }
#pragma unroll 4
    for (long p = pl; p < P; p += PT) {
        float xv[V], rv[V], o[V];
        ldv<V>(x + (base + p) * x_cs + c, xv);
        if (res) ldv<V>(res + (base + p) * res_cs + c, rv);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float t = __builtin_fmaf(xv[v] - mu[v], sc[v], bt[v]);          // explicit: the backward recomputes the mask with this expression
            if (res) t += rv[v];
            o[v] = ss_apply_act(t, act, alpha);
        }
        stv<V>(y + (base + p) * y_cs + c, o);
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void norm_small_bwd_kernel(const T* __restrict__ dy, int dy_cs, const T* __restrict__ x, int x_cs,
                                                             const T* __restrict__ y, int y_cs,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             T* __restrict__ dx, int dx_cs, int acc_dx,
                                                             T* __restrict__ dres, int dres_cs, int acc_dres,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int acc_params,
                                                             int act, float alpha, int G, int C, long P, int CL) {
    __shared__ float red[2 * V][256];
    __shared__ float stat[2 * V][64];
    const int PT = 256 / CL;
    const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
    const int c = (blockIdx.x * CL + cl) * V;
    const bool cval = c < C;
    const bool recompute = act != SS_ACT_NONE && y == nullptr;
    float gm[V], bt[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { gm[v] = 1.f; bt[v] = 0.f; }
    if (cval) {
        if (gamma) ldv<V>(gamma + c, gm);
        if (recompute) ldv<V>(beta + c, bt);
    }
    double tg[V], tgx[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { tg[v] = 0.0; tgx[v] = 0.0; }
    for (int g = 0; g < G; ++g) {
        const long base = (long)g * P;
        float mu[V], rs[V], s1[V], s2[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { mu[v] = 0.f; rs[v] = 0.f; s1[v] = 0.f; s2[v] = 0.f; }
        if (cval) {
            ldv<V>(mean + (long)g * C + c, mu);
            ldv<V>(rstd + (long)g * C + c, rs);
#pragma unroll 2
            for (long p = pl; p < P; p += PT) {
                float xv[V], gv[V];
                ldv<V>(x + (base + p) * x_cs + c, xv);
                ldv<V>(dy + (base + p) * dy_cs + c, gv);
                if (recompute) {
#pragma unroll
                    for (int v = 0; v < V; ++v) gv[v] *= ss_act_grad_from_out(__builtin_fmaf(xv[v] - mu[v], rs[v] * gm[v], bt[v]), act, alpha);
                } else if (act != SS_ACT_NONE) {
                    float yv[V];
                    ldv<V>(y + (base + p) * y_cs + c, yv);
#pragma unroll
                    for (int v = 0; v < V; ++v) gv[v] *= ss_act_grad_from_out(yv[v], act, alpha);
                }
#pragma unroll
                for (int v = 0; v < V; ++v) { s1[v] += gv[v]; s2[v] = fmaf(gv[v], (xv[v] - mu[v]) * rs[v], s2[v]); }
            }
        }
        __syncthreads();                 // previous group's stat[] reads are done
#pragma unroll
        for (int v = 0; v < V; ++v) { red[2 * v][threadIdx.x] = s1[v]; red[2 * v + 1][threadIdx.x] = s2[v]; }
        __syncthreads();
        for (int off = PT / 2; off >= 1; off >>= 1) {
            if (pl < off) {
#pragma unroll
                for (int k = 0; k < 2 * V; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off * CL];
            }
            __syncthreads();
        }
        if (pl == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double a = (double)red[2 * v][threadIdx.x], b = (double)red[2 * v + 1][threadIdx.x];
                stat[2 * v][cl] = (float)(a / (double)P);
                stat[2 * v + 1][cl] = (float)(b / (double)P);
                tg[v] += a;
                tgx[v] += b;
            }
        }
        __syncthreads();
        if (cval) {
            float m1[V], m2[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { m1[v] = stat[2 * v][cl]; m2[v] = stat[2 * v + 1][cl]; }
#pragma unroll 2
            for (long p = pl; p < P; p += PT) {
                float xv[V], gv[V], o[V], r[V];
                ldv<V>(x + (base + p) * x_cs + c, xv);
                ldv<V>(dy + (base + p) * dy_cs + c, gv);
                if (recompute) {
#pragma unroll
                    for (int v = 0; v < V; ++v) gv[v] *= ss_act_grad_from_out(__builtin_fmaf(xv[v] - mu[v], rs[v] * gm[v], bt[v]), act, alpha);
                } else if (act != SS_ACT_NONE) {
                    float yv[V];
                    ldv<V>(y + (base + p) * y_cs + c, yv);
#pragma unroll
                    for (int v = 0; v < V; ++v) gv[v] *= ss_act_grad_from_out(yv[v], act, alpha);
                }
                if (acc_dx) ldv<V>(dx + (base + p) * dx_cs + c, o);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float xh = (xv[v] - mu[v]) * rs[v];
                    const float dv = rs[v] * gm[v] * (gv[v] - m1[v] - xh * m2[v]);
                    o[v] = acc_dx ? o[v] + dv : dv;
                }
                stv<V>(dx + (base + p) * dx_cs + c, o);
                if (dres) {
                    if (acc_dres) {
                        ldv<V>(dres + (base + p) * dres_cs + c, r);
#pragma unroll
                        for (int v = 0; v < V; ++v) r[v] += gv[v];
                        stv<V>(dres + (base + p) * dres_cs + c, r);
                    } else {
                        stv<V>(dres + (base + p) * dres_cs + c, gv);
                    }
                }
            }
        }
    }
    if (pl == 0 && cval) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (dbeta) dbeta[c + v] = acc_params ? dbeta[c + v] + (float)tg[v] : (float)tg[v];
            if (dgamma) dgamma[c + v] = acc_params ? dgamma[c + v] + (float)tgx[v] : (float)tgx[v];
        }
    }
}

// channel lanes of the small-tensor kernels: 8 channels per block (V = 4: 2 lanes, V = 1: 8 lanes)
inline int small_cl(int V) { return V == 4 ? 2 : 8; }
// One block walks a whole group's pixels for its 8 channels: worth it while that walk is short (launch latency dominates).
inline bool norm_small(const ss_norm_desc* d) {
    const long max_pix = ss_tuning().norm_fused_pix;   // 0 disables
    const long P = (long)d->n * d->h * d->w / (d->groups > 0 ? d->groups : 1);
    return P <= max_pix && P * d->groups * d->c <= NORM_SMALL_ELEMS;
}

inline bool al16(const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }
// grid of the channel-stationary apply kernels: (row chunks, channel blocks, groups); CT channel lanes (of V channels) x PT row lanes
struct ApplyGeom { int CT, PT; long rows_per_chunk; dim3 grid; };
inline ApplyGeom apply_geom(const NormGeom& g, int V) {
    ApplyGeom a;
    const int cv = (g.C + V - 1) / V;
    const int cblocks = (cv + 255) / 256;
    a.CT = (cv + cblocks - 1) / cblocks;
    a.PT = 256 / a.CT;
    long want = 2048 / ((long)g.G * cblocks);              // ~8 blocks per CU in total
    if (want < 1) want = 1;
    const long sweep = 4L * a.PT;                          // rows one block takes per (4-fold unrolled) iteration
    long maxchunks = (g.P + 4 * sweep - 1) / (4 * sweep);  // at least four iterations per thread: the operand set-up is amortised
    if (maxchunks < 1) maxchunks = 1;
    long chunks = want < maxchunks ? want : maxchunks;
    long rpc = (g.P + chunks - 1) / chunks;
    rpc = (rpc + a.PT - 1) / a.PT * a.PT;
    chunks = (g.P + rpc - 1) / rpc;
    a.rows_per_chunk = rpc;
    a.grid = dim3((unsigned)chunks, (unsigned)cblocks, (unsigned)g.G);
    return a;
}
// geometry of the fused-finalize form (see NormFin): channel blocks of <= NORM_FUSE_CH channels, <= NORM_FUSE_MAX_CHUNKS pixel chunks
struct FuseGeom { bool ok; int CT, PT, cblocks, chunks; long pix_per_chunk; };
inline FuseGeom fuse_geom(const ss_norm_desc* d, int V, int pass_bit = 1) {
    FuseGeom f{};
    if (!(ss_tuning().norm_fuse_fin & pass_bit)) return f;          // bit 0: forward, bit 1: backward
    const int G = d->groups;
    const long P = (long)d->n * d->h * d->w / G;
    if ((long)d->n * d->h * d->w * d->c > NORM_FUSE_ELEMS) return f;
    const int cv = (d->c + V - 1) / V, lanes_max = NORM_FUSE_CH / V;
    f.cblocks = (cv + lanes_max - 1) / lanes_max;
    f.CT = (cv + f.cblocks - 1) / f.cblocks;
    f.PT = 256 / f.CT;
    long chunks = 512 / ((long)G * f.cblocks);                 // ~512 workgroups in the statistics pass ...
    if (chunks > NORM_FUSE_MAX_CHUNKS) chunks = NORM_FUSE_MAX_CHUNKS;
    const long maxc = P / (4L * f.PT);                         // ... of at least four rows per thread
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    f.pix_per_chunk = (P + chunks - 1) / chunks;
    f.chunks = (int)((P + f.pix_per_chunk - 1) / f.pix_per_chunk);
    f.ok = true;
    return f;
}
inline ApplyGeom apply_geom_fused(const NormGeom& g, const FuseGeom& f) {
    ApplyGeom a;
    a.CT = f.CT;
    a.PT = f.PT;
    long want = 1024 / ((long)g.G * f.cblocks);              // every workgroup repeats the finalize prologue: fewer, longer ones
    if (want < 1) want = 1;
    const long sweep = 4L * a.PT;
    long maxchunks = (g.P + 4 * sweep - 1) / (4 * sweep);
    if (maxchunks < 1) maxchunks = 1;
    long chunks = want < maxchunks ? want : maxchunks;
    long rpc = (g.P + chunks - 1) / chunks;
    rpc = (rpc + a.PT - 1) / a.PT * a.PT;
    chunks = (g.P + rpc - 1) / rpc;
    a.rows_per_chunk = rpc;
    a.grid = dim3((unsigned)chunks, (unsigned)f.cblocks, (unsigned)g.G);
    return a;
}
inline unsigned apply_grid(long total) {
    long b = (total + 255) / 256;
    const long cap = 256L * 32;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

bool valid(const ss_norm_desc* d) {
    if (d && d->struct_size != sizeof(ss_norm_desc)) {
        ss_set_error("ss_norm_desc.struct_size = %u, this library expects %zu", d->struct_size, sizeof(ss_norm_desc));
        return false;
    }
    if (d && (d->dtype < SS_DTYPE_F32 || d->dtype > SS_DTYPE_F16)) { ss_set_error("ss_norm_desc.dtype = %d unknown", d->dtype); return false; }
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0) return false;
    if (d->groups != 1 && d->groups != d->n) return false;
    if (d->x_cstride < d->c || d->y_cstride < d->c) return false;
    return true;
}

// vector width usable for this call: every view pointer 16-B aligned and every stride a multiple of 4
int pick_v(int c, std::initializer_list<int> strides, std::initializer_list<const void*> ptrs) {
    if (c % 4) return 1;
    for (int st : strides) if (st % 4) return 1;
    for (const void* q : ptrs) if (!al16(q)) return 1;
    return 4;
}
// same for 16-bit activation views: four channels are 8 bytes
int pick_v16(int c, std::initializer_list<int> strides, std::initializer_list<const void*> act_ptrs, std::initializer_list<const void*> f32_ptrs) {
    if (c % 4) return 1;
    for (int st : strides) if (st % 4) return 1;
    for (const void* q : act_ptrs) if (q && (((uintptr_t)q) & 7)) return 1;
    for (const void* q : f32_ptrs) if (!al16(q)) return 1;
    return 4;
}

// geometry of the one-pass (register-resident) InstanceNorm backward: groups of npiece workgroups, 32 x nit pixels each
struct ResGeom { bool ok; int nit, npiece; };
inline ResGeom res_geom(const ss_norm_desc* d, int V, int acc_dx, bool acc_dres) {
    ResGeom r{false, 0, 0};
    if (!ss_tuning().norm_bwd_resident || d->dtype != SS_DTYPE_F32 || V != 4 || d->groups != d->n || d->c % 32 || acc_dx || acc_dres) return r;
    if (d->act != SS_ACT_NONE && d->act != SS_ACT_RELU && d->act != SS_ACT_LRELU) return r;
    const long P = (long)d->h * d->w;
    if (P > 16384 || P < 1024) return r;
    r.nit = P > 8192 ? 16 : (P > 4096 ? 8 : 4);          // <= 32 workgroups per group (the progress argument at the kernel)
    r.npiece = (int)((P + 32 * r.nit - 1) / (32 * r.nit));
    r.ok = r.npiece <= 32 && (long)d->n * (d->c / 32) * r.npiece >= 64;
    return r;
}
size_t res_part_bytes(const ss_norm_desc* d) {          // the groups' fp64 partials; the arrival counters and tickets follow
    return ss_align_up((size_t)d->n * (d->c / 32 + 1) * 32 * 64 * sizeof(double), 256);
}
size_t fused_part_bytes(const ss_norm_desc* d) {          // the fused form's fp64 partials; its pivots ([G][C] floats) follow
    return ss_align_up((size_t)d->groups * NORM_FUSE_MAX_CHUNKS * d->c * 2 * sizeof(double), 256);
}
size_t part_bytes(const ss_norm_desc* d) {
    const size_t a = (size_t)d->groups * norm_max_chunks(d->groups) * d->c * 2 * sizeof(float);
    const size_t b = fused_part_bytes(d) + ss_align_up((size_t)d->groups * d->c * sizeof(float), 256);
    const size_t c = d->groups == d->n && d->c % 32 == 0 ? res_part_bytes(d) + ss_align_up((size_t)(d->n + 1) * (d->c / 32 + 1) * sizeof(unsigned int), 256) : 0;
    const size_t ab = a > b ? a : b;
    return ss_align_up(ab > c ? ab : c, 256);
}

}  // namespace

namespace {


template <typename T>
int norm_fwd_t(const ss_norm_desc* d, const T* x, const float* gamma, const float* beta,
                const T* residual, T* y, float* mean, float* rstd,
                float* moving_mean, float* moving_var, float momentum,
                void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !x || !beta || !mean || !rstd) return SS_ERR_INVALID;          // y == NULL: statistics only (the consumer applies)
    if ((moving_mean != nullptr) != (moving_var != nullptr)) return SS_ERR_INVALID;
    if (moving_mean && d->groups != 1) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride, y ? d->y_cstride : 0, residual ? d->res_cstride : 0},
                         {x, y, residual, gamma, beta, mean, rstd});
    const NormGeom g = geom(d, V);
    unsigned int* yam = (unsigned int*)d->y_amax;        // max|y| (of the STORED, i.e. rounded, values) for the next conv's x3h scale
    if (norm_small(d) && y) {
        const int CL = small_cl(V);
        const dim3 grid((g.C + CL * V - 1) / (CL * V), g.G);
        if (V == 4)
            hipLaunchKernelGGL((norm_small_fwd_kernel<T, 4>), grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, residual, d->res_cstride, y,
                               d->y_cstride, mean, rstd, moving_mean, moving_var, momentum, d->eps, d->act, d->act_alpha, g.C, g.P, CL);
        else
            hipLaunchKernelGGL((norm_small_fwd_kernel<T, 1>), grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, residual, d->res_cstride, y,
                               d->y_cstride, mean, rstd, moving_mean, moving_var, momentum, d->eps, d->act, d->act_alpha, g.C, g.P, CL);
        SS_LAUNCH_CHECK();
        return SS_OK;          // small groups: no maximum reported (ss_norm_reports_amax == 0)
    }
    const float* part = (const float*)ws;
    int chunks = g.chunks;
    const double elems = (double)g.G * g.P * g.C;
    const bool ext_stats = d->x_stats && d->x_stats_chunks > 0;
    // (the geometry of the statistics must not depend on whether y is wanted: a deferred norm and its materialised twin agree bit for bit)
    const FuseGeom fz = fuse_geom(d, V);
    if (fz.ok && (!ext_stats || d->x_stats_chunks * (d->n / g.G) <= NORM_FUSE_EXT_CHUNKS)) {
        // consumer-side finalize: [statistics with the fused geometry ->] apply, which reduces the partials of its own channels
        float* pivot = nullptr;
        if (ext_stats) {
            part = (const float*)d->x_stats;
            chunks = d->x_stats_chunks * (d->n / g.G);
        } else {
            const dim3 sgrid(fz.chunks, fz.cblocks, g.G);
            chunks = fz.chunks;
            pivot = (float*)((char*)ws + fused_part_bytes(d));
            SsProfScope prof("norm_stats_kernel<fwd>", 0.0, elems * sizeof(T), s);
            if (V == 4)
                hipLaunchKernelGGL((norm_stats_kernel<T, 0, 4, true>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                                   0, 0.f, g.C, g.P, fz.pix_per_chunk, fz.CT, fz.PT, (float*)ws, nullptr, nullptr, 0, pivot);
            else
                hipLaunchKernelGGL((norm_stats_kernel<T, 0, 1, true>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                                   0, 0.f, g.C, g.P, fz.pix_per_chunk, fz.CT, fz.PT, (float*)ws, nullptr, nullptr, 0, pivot);
            SS_LAUNCH_CHECK();
        }
        NormFin fin{};
        fin.part = part; fin.part64 = ext_stats ? 0 : 1; fin.chunks = chunks; fin.G = g.G; fin.P = (double)g.P; fin.eps = d->eps; fin.momentum = momentum;
        fin.mean = mean; fin.rstd = rstd; fin.mm = moving_mean; fin.mv = moving_var; fin.pivot = pivot;
        if (!y) {          // statistics only
            hipLaunchKernelGGL(norm_finalize_fused_kernel, dim3(fz.cblocks, g.G), dim3(256), 0, s, fin, g.C, fz.CT * V);
            SS_LAUNCH_CHECK();
            return SS_OK;
        }
        const ApplyGeom ag = apply_geom_fused(g, fz);
        SsProfScope prof("norm_apply_kernel", 0.0, elems * sizeof(T) * (residual ? 3 : 2), s);
        if (V == 4)
            hipLaunchKernelGGL((norm_apply_kernel<T, 4, true>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                               residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, 0, fin);
        else
            hipLaunchKernelGGL((norm_apply_kernel<T, 1, true>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                               residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, 0, fin);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    if (ext_stats) {
        // the producing convolution's epilogue already summed x and x^2 ([n][chunks][c][2]; groups = 1: the samples' chunks follow
        // one another, i.e. n * chunks chunks of the one group)
        part = (const float*)d->x_stats;
        chunks = d->x_stats_chunks * (d->n / g.G);
    } else {
        const dim3 sgrid(g.chunks, g.cblocks, g.G);
        SsProfScope prof("norm_stats_kernel<fwd>", 0.0, elems * sizeof(T), s);          // algorithmic bytes: one read of x
        if (V == 4)
            hipLaunchKernelGGL((norm_stats_kernel<T, 0, 4>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                               0, 0.f, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, (float*)ws);
        else
            hipLaunchKernelGGL((norm_stats_kernel<T, 0, 1>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                               0, 0.f, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, (float*)ws);
        SS_LAUNCH_CHECK();
    }
    const int fcl = fin_cl(g.C, g.G);
    hipLaunchKernelGGL(norm_finalize_fwd, dim3((g.C + fcl - 1) / fcl, g.G), dim3(256), 0, s,
                       part, chunks, g.G, g.C, g.P, d->eps, mean, rstd, moving_mean, moving_var, momentum, fcl);
    SS_LAUNCH_CHECK();
    if (!y) return SS_OK;
    const ApplyGeom ag = apply_geom(g, V);
    SsProfScope prof("norm_apply_kernel", 0.0, elems * sizeof(T) * (residual ? 3 : 2), s);          // read x (+ residual), write y
    if (V == 4)
        hipLaunchKernelGGL((norm_apply_kernel<T, 4>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, ss_tuning().norm_order);
    else
        hipLaunchKernelGGL((norm_apply_kernel<T, 1>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, ss_tuning().norm_order);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// the apply pass alone (statistics were taken earlier: ss_norm_fwd with y == NULL)
template <typename T>
int norm_apply_t(const ss_norm_desc* d, const T* x, const float* gamma, const float* beta, const T* residual, T* y,
                 const float* mean, const float* rstd, void* stream) {
    if (!valid(d) || !x || !beta || !y || !mean || !rstd) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride, d->y_cstride, residual ? d->res_cstride : 0}, {x, y, residual, gamma, beta, mean, rstd});
    const NormGeom g = geom(d, V);
    unsigned int* yam = (unsigned int*)d->y_amax;
    const ApplyGeom ag = apply_geom(g, V);
    SsProfScope prof("norm_apply_kernel", 0.0, (double)g.G * g.P * g.C * sizeof(T) * (residual ? 3 : 2), s);
    if (V == 4)
        hipLaunchKernelGGL((norm_apply_kernel<T, 4>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, ss_tuning().norm_order);
    else
        hipLaunchKernelGGL((norm_apply_kernel<T, 1>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, yam, ss_tuning().norm_order);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T>
int norm_infer_t(const ss_norm_desc* d, const T* x, const float* gamma, const float* beta,
                  const float* moving_mean, const float* moving_var, const T* residual, T* y, void* stream) {
    if (!valid(d) || !x || !beta || !y || !moving_mean || !moving_var) return SS_ERR_INVALID;
    const long rows = (long)d->n * d->h * d->w;
    hipLaunchKernelGGL(norm_infer_kernel<T>, dim3((unsigned)((rows * d->c + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, d->x_cstride, gamma, beta, moving_mean, moving_var, d->eps, residual, d->res_cstride,
                       y, d->y_cstride, d->act, d->act_alpha, d->c, rows);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T>
int norm_bwd_t(const ss_norm_desc* d, const T* dy, int32_t dy_cstride, const T* x, const T* y,
                const float* gamma, const float* beta, const float* mean, const float* rstd,
                T* dx, int32_t dx_cstride, int accumulate_dx, T* dres, int accumulate_dres,
                float* dgamma, float* dbeta, int accumulate_params,
                void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !dy || !x || !mean || !rstd || !dx) return SS_ERR_INVALID;
    if (d->act != SS_ACT_NONE && !y) {      // y may be omitted only where its sign can be recomputed from x: relu / lrelu, no residual
        if ((d->act != SS_ACT_RELU && d->act != SS_ACT_LRELU) || !beta || dres) return SS_ERR_INVALID;
    }
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool use_y = d->act != SS_ACT_NONE && y != nullptr;
    const int V = pick_v(d->c, {d->x_cstride, dy_cstride, dx_cstride, use_y ? d->y_cstride : 0, dres ? d->res_cstride : 0},
                         {x, dy, dx, use_y ? y : nullptr, dres, gamma, beta, mean, rstd});
    const NormGeom g = geom(d, V);
    unsigned int* dxam = (unsigned int*)d->dx_amax;      // max|dx| (stored values) for the previous conv's x3h scales
    if (norm_small(d)) {
        const int CL = small_cl(V);
        const dim3 grid((g.C + CL * V - 1) / (CL * V));
        if (V == 4)
            hipLaunchKernelGGL((norm_small_bwd_kernel<T, 4>), grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride, gamma, beta,
                               mean, rstd, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres, dgamma, dbeta,
                               accumulate_params, d->act, d->act_alpha, g.G, g.C, g.P, CL);
        else
            hipLaunchKernelGGL((norm_small_bwd_kernel<T, 1>), grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride, gamma, beta,
                               mean, rstd, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres, dgamma, dbeta,
                               accumulate_params, d->act, d->act_alpha, g.G, g.C, g.P, CL);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + part_bytes(d));
    const double elems = (double)g.G * g.P * g.C;
    if constexpr (std::is_same<T, float>::value) {
        const ResGeom rg = res_geom(d, V, accumulate_dx, dres != nullptr && accumulate_dres);
        if (rg.ok) {          // one pass: the (sample, 32-channel block) groups keep their share of dy and x in registers across a group-local barrier
            NormRes r{};
            r.part = (double*)ws;
            r.counter = (unsigned int*)((char*)ws + res_part_bytes(d));
            r.rt = (double*)((char*)ws + part_bytes(d) + ss_align_up((size_t)g.G * g.C * 2 * sizeof(float), 256));
            r.dgamma = dgamma; r.dbeta = dbeta; r.acc_params = accumulate_params; r.n = g.G; r.npiece = rg.npiece;
            const int groups = g.G * (g.C / 32);
            (void)hipMemsetAsync(r.counter, 0, (size_t)(groups + g.C / 32) * sizeof(unsigned int), s);
            SsProfScope prof("norm_bwd_resident_kernel", 0.0, elems * sizeof(T) * ((use_y ? 3 : 2) + 1 + (dres ? 1 : 0)), s);
            const dim3 grid((unsigned)(groups * rg.npiece));
#define SS_NBR(NIT_) hipLaunchKernelGGL((norm_bwd_resident_kernel<NIT_>), grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride, gamma, beta, \
                                        mean, rstd, dx, dx_cstride, dres, d->res_cstride, d->act, d->act_alpha, g.C, (int)g.P, dxam, r)
            if (rg.nit == 16) SS_NBR(16); else if (rg.nit == 8) SS_NBR(8); else SS_NBR(4);
#undef SS_NBR
            SS_LAUNCH_CHECK();
            return SS_OK;
        }
    }
    const FuseGeom fz = fuse_geom(d, V, 2);
    if (fz.ok) {          // consumer-side finalize: statistics with the fused geometry -> apply, which reduces the partials of its own channels
        const dim3 fgrid(fz.chunks, fz.cblocks, g.G);
        {
            SsProfScope prof("norm_stats_kernel<bwd>", 0.0, elems * sizeof(T) * (use_y ? 3 : 2), s);
            if (V == 4)
                hipLaunchKernelGGL((norm_stats_kernel<T, 1, 4, true>), fgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                                   d->act, d->act_alpha, g.C, g.P, fz.pix_per_chunk, fz.CT, fz.PT, part, gamma, beta, 0);
            else
                hipLaunchKernelGGL((norm_stats_kernel<T, 1, 1, true>), fgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                                   d->act, d->act_alpha, g.C, g.P, fz.pix_per_chunk, fz.CT, fz.PT, part, gamma, beta, 0);
            SS_LAUNCH_CHECK();
        }
        const ApplyGeom ag = apply_geom_fused(g, fz);
        NormFin fin{};
        fin.part = part; fin.part64 = 1; fin.chunks = fz.chunks; fin.G = g.G; fin.P = (double)g.P;
        fin.dgamma = dgamma; fin.dbeta = dbeta; fin.acc_params = accumulate_params;
        SsProfScope prof("norm_bwd_apply_kernel", 0.0,
                         elems * sizeof(T) * ((use_y ? 3 : 2) + 1 + (accumulate_dx ? 1 : 0) + (dres ? (accumulate_dres ? 2 : 1) : 0)), s);
        if (V == 4)
            hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 4, true>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                               gamma, mean, rstd, nullptr, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                               d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, beta, dxam, nullptr, g.G, nullptr, nullptr, 0, 0, fin);
        else
            hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1, true>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                               gamma, mean, rstd, nullptr, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                               d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, beta, dxam, nullptr, g.G, nullptr, nullptr, 0, 0, fin);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }
    const dim3 sgrid(g.chunks, g.cblocks, g.G);
    {
        SsProfScope prof("norm_stats_kernel<bwd>", 0.0, elems * sizeof(T) * (use_y ? 3 : 2), s);          // read dy, x (+ y)
        if (V == 4)
            hipLaunchKernelGGL((norm_stats_kernel<T, 1, 4>), sgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                               d->act, d->act_alpha, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part, gamma, beta, ss_tuning().norm_order);
        else
            hipLaunchKernelGGL((norm_stats_kernel<T, 1, 1>), sgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                               d->act, d->act_alpha, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part, gamma, beta, ss_tuning().norm_order);
        SS_LAUNCH_CHECK();
    }
    double* rt = (double*)((char*)ws + part_bytes(d) + ss_align_up((size_t)g.G * g.C * 2 * sizeof(float), 256));
    const int fcl = fin_cl(g.C, g.G);
    hipLaunchKernelGGL(norm_finalize_bwd, dim3((g.C + fcl - 1) / fcl, g.G), dim3(256), 0, s,
                       part, g.chunks, g.G, g.C, g.P, sums, rt, fcl);
    SS_LAUNCH_CHECK();
    const ApplyGeom ag = apply_geom(g, V);
    const double* prt = (dgamma || dbeta) ? rt : nullptr;
    // read dy, x (+ y), write dx (+ read when accumulating), dres written (+ read when accumulating)
    SsProfScope prof("norm_bwd_apply_kernel", 0.0,
                     elems * sizeof(T) * ((use_y ? 3 : 2) + 1 + (accumulate_dx ? 1 : 0) + (dres ? (accumulate_dres ? 2 : 1) : 0)), s);
    if (V == 4)
        hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 4>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                           gamma, mean, rstd, sums, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                           d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, beta, dxam, prt, g.G, dgamma, dbeta, accumulate_params, ss_tuning().norm_order);
    else
        hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                           gamma, mean, rstd, sums, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                           d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk, beta, dxam, prt, g.G, dgamma, dbeta, accumulate_params, ss_tuning().norm_order);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

/* ---- two-phase forms for data-parallel (cross-rank) batch statistics: stats -> all-reduce(sum) of `sums` by the caller -> finish ---- */
template <typename T>
int norm_fwd_stats_t(const ss_norm_desc* d, const T* x, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !x || !sums) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride}, {x});
    const NormGeom g = geom(d, V);
    float* part = (float*)ws;
    const dim3 sgrid(g.chunks, g.cblocks, g.G);
    if (V == 4)
        hipLaunchKernelGGL((norm_stats_kernel<T, 0, 4>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                           0, 0.f, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    else
        hipLaunchKernelGGL((norm_stats_kernel<T, 0, 1>), sgrid, dim3(256), 0, s, x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr,
                           0, 0.f, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    SS_LAUNCH_CHECK();
    const long gc = (long)g.G * g.C;
    hipLaunchKernelGGL(norm_collapse_kernel, dim3((unsigned)((gc + 255) / 256)), dim3(256), 0, s, part, g.chunks, g.G, g.C, sums);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T>
int norm_fwd_finish_t(const ss_norm_desc* d, const T* x, const float* gamma, const float* beta, const T* residual, T* y,
                       const float* sums, int64_t total_count, float* mean, float* rstd,
                       float* moving_mean, float* moving_var, float momentum, void* stream) {
    if (!valid(d) || !x || !beta || !y || !sums || !mean || !rstd || total_count <= 0) return SS_ERR_INVALID;
    if ((moving_mean != nullptr) != (moving_var != nullptr)) return SS_ERR_INVALID;
    if (moving_mean && d->groups != 1) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride, d->y_cstride, residual ? d->res_cstride : 0}, {x, y, residual, gamma, beta, mean, rstd});
    const NormGeom g = geom(d, V);
    hipLaunchKernelGGL(norm_finalize_fwd, dim3((g.C + FIN_CL - 1) / FIN_CL, g.G), dim3(256), 0, s,
                       sums, 1, g.G, g.C, (long)total_count, d->eps, mean, rstd, moving_mean, moving_var, momentum, FIN_CL);
    SS_LAUNCH_CHECK();
    const ApplyGeom ag = apply_geom(g, V);
    if (V == 4)
        hipLaunchKernelGGL((norm_apply_kernel<T, 4>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk);
    else
        hipLaunchKernelGGL((norm_apply_kernel<T, 1>), ag.grid, dim3(256), 0, s, x, d->x_cstride, gamma, beta, mean, rstd,
                           residual, d->res_cstride, y, d->y_cstride, d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T>
int norm_bwd_stats_t(const ss_norm_desc* d, const T* dy, int32_t dy_cstride, const T* x, const T* y,
                      const float* mean, const float* rstd, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !dy || !x || !mean || !rstd || !sums) return SS_ERR_INVALID;
    if (d->act != SS_ACT_NONE && !y) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride, dy_cstride, d->act != SS_ACT_NONE ? d->y_cstride : 0},
                         {x, dy, d->act != SS_ACT_NONE ? y : nullptr, mean, rstd});
    const NormGeom g = geom(d, V);
    float* part = (float*)ws;
    const dim3 sgrid(g.chunks, g.cblocks, g.G);
    if (V == 4)
        hipLaunchKernelGGL((norm_stats_kernel<T, 1, 4>), sgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                           d->act, d->act_alpha, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    else
        hipLaunchKernelGGL((norm_stats_kernel<T, 1, 1>), sgrid, dim3(256), 0, s, x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd,
                           d->act, d->act_alpha, g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    SS_LAUNCH_CHECK();
    const long gc = (long)g.G * g.C;
    hipLaunchKernelGGL(norm_collapse_kernel, dim3((unsigned)((gc + 255) / 256)), dim3(256), 0, s, part, g.chunks, g.G, g.C, sums);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

template <typename T>
int norm_bwd_finish_t(const ss_norm_desc* d, const T* dy, int32_t dy_cstride, const T* x, const T* y,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* global_sums, const float* local_sums, int64_t total_count,
                       T* dx, int32_t dx_cstride, int accumulate_dx, T* dres, int accumulate_dres,
                       float* dgamma, float* dbeta, int accumulate_params, void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !dy || !x || !mean || !rstd || !dx || !global_sums || !local_sums || total_count <= 0) return SS_ERR_INVALID;
    if (d->act != SS_ACT_NONE && !y) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int V = pick_v(d->c, {d->x_cstride, dy_cstride, dx_cstride, d->act != SS_ACT_NONE ? d->y_cstride : 0, dres ? d->res_cstride : 0},
                         {x, dy, dx, d->act != SS_ACT_NONE ? y : nullptr, dres, gamma, mean, rstd});
    const NormGeom g = geom(d, V);
    float* means = (float*)((char*)ws + part_bytes(d));
    hipLaunchKernelGGL(norm_finalize_bwd_sync, dim3((g.C + 255) / 256), dim3(256), 0, s, global_sums, local_sums, g.G, g.C,
                       (double)total_count, means, dgamma, dbeta, accumulate_params);
    SS_LAUNCH_CHECK();
    const ApplyGeom ag = apply_geom(g, V);
    if (V == 4)
        hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 4>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                           gamma, mean, rstd, means, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                           d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk);
    else
        hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1>), ag.grid, dim3(256), 0, s, dy, dy_cstride, x, d->x_cstride, y, d->y_cstride,
                           gamma, mean, rstd, means, dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres,
                           d->act, d->act_alpha, g.C, g.P, ag.CT, ag.PT, ag.rows_per_chunk);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

}  // namespace

#define SS_NDT(dtype, ...)                                                       \
    switch (dtype) {                                                            \
        case SS_DTYPE_F32: { typedef float T; __VA_ARGS__; }                    \
        case SS_DTYPE_F16: { typedef _Float16 T; __VA_ARGS__; }                 \
        case SS_DTYPE_BF16: { typedef __bf16 T; __VA_ARGS__; }                  \
        default: return SS_ERR_INVALID;                                         \
    }

extern "C" {

/* Diagnostics: how many workgroups of the one-pass InstanceNorm backward gave up waiting at their group barrier since the library was
 * loaded (0 in every healthy run; such a launch produced wrong gradients).  Synchronises the device. */
int ss_norm_resident_timeouts(void) {
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_norm_resident_timeouts), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}

int ss_norm_reports_amax(const ss_norm_desc* d) { return valid(d) && !norm_small(d) ? 1 : 0; }

size_t ss_norm_workspace_bytes(const ss_norm_desc* d) {
    if (!valid(d)) return 0;
    // stats partials | per-(group, channel) means of the backward | their raw totals (double; parameter gradients)
    return part_bytes(d) + ss_align_up((size_t)d->groups * d->c * 2 * sizeof(float), 256) + ss_align_up((size_t)d->groups * d->c * 2 * sizeof(double), 256);
}

int ss_norm_fwd(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta,
                const void* residual, void* y, float* mean, float* rstd,
                float* moving_mean, float* moving_var, float momentum,
                void* ws, size_t ws_bytes, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_fwd_t<T>(d, (const T*)x, gamma, beta, (const T*)residual, (T*)y, mean, rstd, moving_mean, moving_var, momentum, ws, ws_bytes, stream));
}

int ss_norm_apply(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                  const float* mean, const float* rstd, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_apply_t<T>(d, (const T*)x, gamma, beta, (const T*)residual, (T*)y, mean, rstd, stream));
}

int ss_norm_infer(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta,
                  const float* moving_mean, const float* moving_var, const void* residual, void* y, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_infer_t<T>(d, (const T*)x, gamma, beta, moving_mean, moving_var, (const T*)residual, (T*)y, stream));
}

int ss_norm_bwd(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                const float* gamma, const float* beta, const float* mean, const float* rstd,
                void* dx, int32_t dx_cstride, int accumulate_dx, void* dres, int accumulate_dres,
                float* dgamma, float* dbeta, int accumulate_params,
                void* ws, size_t ws_bytes, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_bwd_t<T>(d, (const T*)dy, dy_cstride, (const T*)x, (const T*)y, gamma, beta, mean, rstd, (T*)dx, dx_cstride, accumulate_dx, (T*)dres, accumulate_dres, dgamma, dbeta, accumulate_params, ws, ws_bytes, stream));
}

int ss_norm_fwd_stats(const ss_norm_desc* d, const void* x, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_fwd_stats_t<T>(d, (const T*)x, sums, ws, ws_bytes, stream));
}

int ss_norm_fwd_finish(const ss_norm_desc* d, const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                       const float* sums, int64_t total_count, float* mean, float* rstd,
                       float* moving_mean, float* moving_var, float momentum, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_fwd_finish_t<T>(d, (const T*)x, gamma, beta, (const T*)residual, (T*)y, sums, total_count, mean, rstd, moving_mean, moving_var, momentum, stream));
}

int ss_norm_bwd_stats(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                      const float* mean, const float* rstd, float* sums, void* ws, size_t ws_bytes, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_bwd_stats_t<T>(d, (const T*)dy, dy_cstride, (const T*)x, (const T*)y, mean, rstd, sums, ws, ws_bytes, stream));
}

int ss_norm_bwd_finish(const ss_norm_desc* d, const void* dy, int32_t dy_cstride, const void* x, const void* y,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* global_sums, const float* local_sums, int64_t total_count,
                       void* dx, int32_t dx_cstride, int accumulate_dx, void* dres, int accumulate_dres,
                       float* dgamma, float* dbeta, int accumulate_params, void* ws, size_t ws_bytes, void* stream) {
    if (!d) return SS_ERR_INVALID;
    SS_NDT(d->dtype, return norm_bwd_finish_t<T>(d, (const T*)dy, dy_cstride, (const T*)x, (const T*)y, gamma, mean, rstd, global_sums, local_sums, total_count, (T*)dx, dx_cstride, accumulate_dx, (T*)dres, accumulate_dres, dgamma, dbeta, accumulate_params, ws, ws_bytes, stream));
}

}  // extern "C"
