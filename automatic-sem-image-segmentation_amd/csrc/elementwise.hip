// Element-wise, pooling, loss and optimizer kernels (all HBM-bound; NHWC views with pixel strides).
#include "common.h"

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(long total) {
    long b = (total + EW_BLOCK - 1) / EW_BLOCK;
    const long cap = 256L * 16;      // grid-stride beyond ~16 blocks per CU
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ImagePool.query on the device (ss_pool_query): blockIdx.y = image of the query; byte-exact copies, swap = load old / store new / emit old
struct PoolPlan { int mode[SS_POOL_MAX_QUERY]; int slot[SS_POOL_MAX_QUERY]; };
template <class U>
__global__ __launch_bounds__(256) void pool_query_kernel(U* __restrict__ pool, const U* __restrict__ images, U* __restrict__ out, long units, PoolPlan plan) {
    const int i = blockIdx.y;
    const int mode = plan.mode[i];
    const U* src = images + (long)i * units;
    U* dst = out + (long)i * units;
    U* keep = mode != SS_POOL_PASS ? pool + (long)plan.slot[i] * units : nullptr;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < units; e += (long)gridDim.x * blockDim.x) {
        const U cur = src[e];
        if (mode == SS_POOL_SWAP) { const U old = keep[e]; keep[e] = cur; dst[e] = old; }
        else { if (mode == SS_POOL_FILL) keep[e] = cur; dst[e] = cur; }
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void act_bwd_kernel(int act, float alpha, const T* __restrict__ dy, int dy_cs,
                                                           const T* __restrict__ y, int y_cs, T* __restrict__ dx, int dx_cs,
                                                           long rows, int C) {
    const long total = rows * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        dx[r * dx_cs + c] = (T)((float)dy[r * dy_cs + c] * ss_act_grad_from_out((float)y[r * y_cs + c], act, alpha));
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void axpby_kernel(float alpha, const T* __restrict__ a, int a_cs, float beta,
                                                         const T* __restrict__ b, int b_cs, T* __restrict__ out, int out_cs,
                                                         long rows, int C) {
    const long total = rows * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        float v = alpha * (float)a[r * a_cs + c];
        if (b) v = fmaf(beta, (float)b[r * b_cs + c], v);
        out[r * out_cs + c] = (T)v;
    }
}

// out = scale * a .* b  (Dropout with an explicit keep mask: WassersteinGAN.py:566-567,621)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void mul_kernel(float scale, const T* __restrict__ a, int a_cs, const T* __restrict__ b, int b_cs,
                                                       T* __restrict__ out, int out_cs, long rows, int C) {
    const long total = rows * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        out[r * out_cs + c] = (T)(scale * (float)a[r * a_cs + c] * (float)b[r * b_cs + c]);
    }
}

// WGAN-GP (WassersteinGAN.py:113-116): one block per sample; norm_i = sqrt(sum_j g_ij^2) (fixed-order tree: deterministic),
// gbar_ij = coef * 2 (norm_i - 1) / norm_i * g_ij = d/dg_ij of coef * (norm_i - 1)^2
__global__ __launch_bounds__(256) void gp_grad_kernel(const float* __restrict__ g, long per_sample, float coef, float* __restrict__ gbar,
                                                      float* __restrict__ norms) {
    __shared__ double red[256];
    const float* gi = g + (long)blockIdx.x * per_sample;
    double s = 0.0;
    for (long j = threadIdx.x; j < per_sample; j += 256) s += (double)gi[j] * (double)gi[j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const float norm = (float)sqrt(red[0]);
    if (threadIdx.x == 0) norms[blockIdx.x] = norm;
    if (gbar) {
        const float f = norm > 0.f ? coef * 2.f * (norm - 1.f) / norm : 0.f;
        float* o = gbar + (long)blockIdx.x * per_sample;
        for (long j = threadIdx.x; j < per_sample; j += 256) o[j] = f * gi[j];
    }
}

// out[i][j] = real[i][j] + alpha[i] * (fake[i][j] - real[i][j])   (WGAN_GP.gradient_penalty, WassersteinGAN.py:97-99)
__global__ __launch_bounds__(EW_BLOCK) void interpolate_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                               const float* __restrict__ alpha, float* __restrict__ out, long n, long per_sample) {
    const long total = n * per_sample;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const float r = real[e];
        out[e] = r + alpha[e / per_sample] * (fake[e] - r);
    }
}

// dst (type TD, view) = src (type TS, view): the storage-type boundary (fp32 <-> bf16 / fp16 activations)
template <typename TS, typename TD>
__global__ __launch_bounds__(EW_BLOCK) void convert_kernel(const TS* __restrict__ src, int src_cs, TD* __restrict__ dst, int dst_cs, long rows, int C) {
    const long total = rows * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / C;
        const int c = (int)(e - r * C);
        dst[r * dst_cs + c] = (TD)(float)src[r * src_cs + c];
    }
}

__global__ __launch_bounds__(EW_BLOCK) void fill_kernel(float* __restrict__ dst, float v, long count) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) dst[e] = v;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool_fwd_kernel(const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs,
                                                               int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * OH * OW * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        const T* ip = x + ((long)(n * H + 2 * oy) * W + 2 * ox) * x_cs + c;
        const float v00 = (float)ip[0], v01 = (float)ip[x_cs], v10 = (float)ip[(long)W * x_cs], v11 = (float)ip[(long)(W + 1) * x_cs];
        y[((long)(n * OH + oy) * OW + ox) * y_cs + c] = (T)fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
    }
}

// one thread per INPUT element: gets dy of its window iff it is the first maximum in row-major order
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool_bwd_kernel(const T* __restrict__ dy, int dy_cs, const T* __restrict__ x, int x_cs,
                                                               T* __restrict__ dx, int dx_cs, int accumulate,
                                                               int N, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    const long total = (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float g = 0.f;
        const int oy = iy >> 1, ox = ix >> 1;
        if (oy < OH && ox < OW) {
            const T* ip = x + ((long)(n * H + 2 * oy) * W + 2 * ox) * x_cs + c;
            const float v[4] = {(float)ip[0], (float)ip[x_cs], (float)ip[(long)W * x_cs], (float)ip[(long)(W + 1) * x_cs]};
            int arg = 0;
            float m = v[0];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k] > m) { m = v[k]; arg = k; }
            if (arg == ((iy & 1) * 2 + (ix & 1))) g = (float)dy[((long)(n * OH + oy) * OW + ox) * dy_cs + c];
        }
        T* o = dx + ((long)(n * H + iy) * W + ix) * dx_cs + c;
        *o = (T)(accumulate ? ((float)*o + g) : g);
    }
}


// y[n,py,px,c] = x[n, reflect(py-pt), reflect(px-pl), c]   (keras.ops.pad(mode="reflect"), CycleGAN.py:495-506)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void reflect_pad_fwd_kernel(const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs,
                                                                   int N, int H, int W, int C, int pt, int pl, int PH, int PW) {
    const long total = (long)N * PH * PW * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int px = (int)(r % PW); r /= PW;
        const int py = (int)(r % PH);
        const int n = (int)(r / PH);
        const int iy = ss_map_index(py - pt, H, 1), ix = ss_map_index(px - pl, W, 1);
        y[((long)(n * PH + py) * PW + px) * y_cs + c] = x[((long)(n * H + iy) * W + ix) * x_cs + c];
    }
}

// dx[n,iy,ix,c] (+)= sum of dy over the padded positions that reflect onto (iy,ix)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void reflect_pad_bwd_kernel(const T* __restrict__ dy, int dy_cs, T* __restrict__ dx, int dx_cs,
                                                                   int accumulate, int N, int H, int W, int C, int pt, int pl, int PH, int PW) {
    const long total = (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = iy + pt;
        if (iy >= 1 && pt - iy >= 0) ys[ny++] = pt - iy;
        { const int py = pt + 2 * (H - 1) - iy; if (iy <= H - 2 && py < PH) ys[ny++] = py; }
        xs[nx++] = ix + pl;
        if (ix >= 1 && pl - ix >= 0) xs[nx++] = pl - ix;
        { const int px = pl + 2 * (W - 1) - ix; if (ix <= W - 2 && px < PW) xs[nx++] = px; }
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += (float)dy[((long)(n * PH + ys[a]) * PW + xs[b]) * dy_cs + c];
        T* o = dx + ((long)(n * H + iy) * W + ix) * dx_cs + c;
        *o = (T)(accumulate ? ((float)*o + acc) : acc);
    }
}

// MODE 0: y = x[:, top:top+OH, left:left+OW]   MODE 1 (backward): dx (+)= dy placed at (top,left), zero elsewhere
template <typename T, int MODE>
__global__ __launch_bounds__(EW_BLOCK) void crop_kernel(const T* __restrict__ src, int src_cs, T* __restrict__ dst, int dst_cs,
                                                        int accumulate, int N, int H, int W, int C, int top, int left, int OH, int OW) {
    const long total = MODE == 0 ? (long)N * OH * OW * C : (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        if (MODE == 0) {
            const int ox = (int)(r % OW); r /= OW;
            const int oy = (int)(r % OH);
            const int n = (int)(r / OH);
            dst[((long)(n * OH + oy) * OW + ox) * dst_cs + c] = src[((long)(n * H + oy + top) * W + ox + left) * src_cs + c];
        } else {
            const int ix = (int)(r % W); r /= W;
            const int iy = (int)(r % H);
            const int n = (int)(r / H);
            const int oy = iy - top, ox = ix - left;
            const float g = (oy >= 0 && oy < OH && ox >= 0 && ox < OW) ? (float)src[((long)(n * OH + oy) * OW + ox) * src_cs + c] : 0.f;
            T* o = dst + ((long)(n * H + iy) * W + ix) * dst_cs + c;
            *o = (T)(accumulate ? ((float)*o + g) : g);
        }
    }
}

// MODE 0: nearest-neighbour 2x upsampling (keras.layers.UpSampling2D, CycleGAN.py:349); MODE 1: its backward (2x2 sums)
template <typename T, int MODE>
__global__ __launch_bounds__(EW_BLOCK) void upsample2x_kernel(const T* __restrict__ src, int src_cs, T* __restrict__ dst, int dst_cs,
                                                              int accumulate, int N, int H, int W, int C) {
    const long total = MODE == 0 ? (long)N * 2 * H * 2 * W * C : (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        if (MODE == 0) {
            const int ox = (int)(r % (2 * W)); r /= 2 * W;
            const int oy = (int)(r % (2 * H));
            const int n = (int)(r / (2 * H));
            dst[((long)(n * 2 * H + oy) * 2 * W + ox) * dst_cs + c] = src[((long)(n * H + oy / 2) * W + ox / 2) * src_cs + c];
        } else {
            const int ix = (int)(r % W); r /= W;
            const int iy = (int)(r % H);
            const int n = (int)(r / H);
            const T* q = src + ((long)(n * 2 * H + 2 * iy) * 2 * W + 2 * ix) * src_cs + c;
            const float g = (float)q[0] + (float)q[src_cs] + (float)q[(long)2 * W * src_cs] + (float)q[(long)(2 * W + 1) * src_cs];
            T* o = dst + ((long)(n * H + iy) * W + ix) * dst_cs + c;
            *o = (T)(accumulate ? ((float)*o + g) : g);
        }
    }
}

// ---- losses: stage 1 per-block partial sums (K values each), stage 2 single block finishes -------
constexpr int LOSS_MAX_BLOCKS = 1024;

template <int K>
__device__ __forceinline__ void block_reduce_store(float (&v)[K], float* part) {
    __shared__ float red[K][EW_BLOCK / 64];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = v[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
            for (int w = 0; w < EW_BLOCK / 64; ++w) s += red[k][w];
            part[(long)blockIdx.x * K + k] = s;
        }
    }
}

template <int K>
__global__ __launch_bounds__(EW_BLOCK) void loss_finish_kernel(const float* __restrict__ part, int nblocks, float inv_count, float* __restrict__ out) {
    __shared__ double red[K][EW_BLOCK];
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += EW_BLOCK)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += part[(long)b * K + k];
#pragma unroll
    for (int k = 0; k < K; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = EW_BLOCK / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off)
#pragma unroll
            for (int k = 0; k < K; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = (float)(red[k][0] * (double)inv_count);
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void mse_const_kernel(const T* __restrict__ pred, long count, float target, float gscale,
                                                             T* __restrict__ grad, float* __restrict__ part) {
    float v[1] = {0.f};
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const float d = (float)pred[e] - target;
        v[0] = fmaf(d, d, v[0]);
        if (grad) grad[e] = (T)(gscale * 2.f * d);
    }
    block_reduce_store<1>(v, part);
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void mae_kernel(const T* __restrict__ truth, const T* __restrict__ pred, long count, float gscale,
                                                       T* __restrict__ grad, float* __restrict__ part) {
    float v[1] = {0.f};
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const float d = (float)pred[e] - (float)truth[e];
        v[0] += fabsf(d);
        if (grad) grad[e] = (T)(gscale * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
    }
    block_reduce_store<1>(v, part);
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void wbce_kernel(const T* __restrict__ truth, const T* __restrict__ pred, long count,
                                                        float weighting, float gscale, T* __restrict__ grad, float* __restrict__ part) {
    float v[3] = {0.f, 0.f, 0.f};
    const float eps = 1e-7f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const float t = (float)truth[e], pr = (float)pred[e];
        const float pc = fminf(fmaxf(pr, eps), 1.f - eps);
        const float w = t * (weighting - 1.f) + 1.f;
        v[0] += -w * (t * logf(pc) + (1.f - t) * logf(1.f - pc));
        v[1] += fabsf(t - pr);
        v[2] += ((pr > 0.5f ? 1.f : 0.f) == t) ? 1.f : 0.f;
        if (grad) {
            // d/dp of -w[t log p + (1-t) log(1-p)], zero where the clip is active (as torch.clip backward)
            const bool inside = (pr >= eps) && (pr <= 1.f - eps);
            grad[e] = (T)(inside ? gscale * w * (-(t / pc) + (1.f - t) / (1.f - pc)) : 0.f);
        }
    }
    block_reduce_store<3>(v, part);
}

// ---- multi-class head of the MultiResUNet (UNet_Segmentation.py:558-560): softmax over the channels of a pixel, and the reference's
// loss closure on a multi-channel output (:379-384): BinaryCrossentropy(reduction none) averages over the channels, the result is
// broadcast back and weighted with y_true * (w - 1) + 1 ------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void softmax_fwd_kernel(const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs, long rows, int C) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const T* xr = x + r * x_cs;
        float m = (float)xr[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, (float)xr[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf((float)xr[c] - m);
        const float inv = 1.f / s;
        for (int c = 0; c < C; ++c) y[r * y_cs + c] = (T)(expf((float)xr[c] - m) * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void softmax_bwd_kernel(const T* __restrict__ dy, int dy_cs, const T* __restrict__ y, int y_cs,
                                                               T* __restrict__ dx, int dx_cs, long rows, int C) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf((float)dy[r * dy_cs + c], (float)y[r * y_cs + c], dot);
        for (int c = 0; c < C; ++c) dx[r * dx_cs + c] = (T)((float)y[r * y_cs + c] * ((float)dy[r * dy_cs + c] - dot));
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void wbce_mc_kernel(const T* __restrict__ truth, const T* __restrict__ pred, long rows, int C,
                                                           float weighting, float gscale, T* __restrict__ grad, float* __restrict__ part) {
    float v[3] = {0.f, 0.f, 0.f};
    const float eps = 1e-7f, invc = 1.f / (float)C;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const T* t_ = truth + r * C;
        const T* p_ = pred + r * C;
        float b = 0.f, wsum = 0.f, tmax = -1.f, pmax = -1.f;
        int ta = 0, pa = 0;
        for (int c = 0; c < C; ++c) {
            const float t = (float)t_[c], pr = (float)p_[c];
            const float pc = fminf(fmaxf(pr, eps), 1.f - eps);
            b += -(t * logf(pc) + (1.f - t) * logf(1.f - pc));
            wsum += t * (weighting - 1.f) + 1.f;
            v[1] += fabsf(t - pr);
            if (t > tmax) { tmax = t; ta = c; }          // first maximum, like argmax
            if (pr > pmax) { pmax = pr; pa = c; }
        }
        b *= invc;
        v[0] += wsum * b;
        v[2] += ta == pa ? (float)C : 0.f;               // categorical accuracy per PIXEL; the finish divides by rows * C
        if (grad) {
            for (int c = 0; c < C; ++c) {
                const float t = (float)t_[c], pr = (float)p_[c];
                const float pc = fminf(fmaxf(pr, eps), 1.f - eps);
                const bool inside = (pr >= eps) && (pr <= 1.f - eps);
                grad[r * C + c] = (T)(inside ? gscale * wsum * invc * (-(t / pc) + (1.f - t) / (1.f - pc)) : 0.f);
            }
        }
    }
    block_reduce_store<3>(v, part);
}

__global__ __launch_bounds__(EW_BLOCK) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long count, float alpha, float b1, float b2, float eps,
                                                        float gscale, const float* __restrict__ alpha_dev) {
    if (alpha_dev) alpha = *alpha_dev;          // step size from device memory: a captured hipGraph replays with a new one every step
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
        const float gv = g[e] * gscale;
        const float mv = m[e] + (gv - m[e]) * b1;       // b1, b2 hold (1 - beta), see adam_kernel_v4
        const float vv = v[e] + (gv * gv - v[e]) * b2;
        m[e] = mv;
        v[e] = vv;
        p[e] -= alpha * mv / (sqrtf(vv) + eps);
    }
}

__global__ __launch_bounds__(EW_BLOCK) void adam_kernel_v4(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m,
                                                           f32x4* __restrict__ v, long count4, float alpha, float b1, float b2, float eps,
                                                           float gscale, const float* __restrict__ alpha_dev) {
    if (alpha_dev) alpha = *alpha_dev;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count4; e += (long)gridDim.x * blockDim.x) {
        const f32x4 gv = g[e] * gscale;
        f32x4 mv = m[e], vv = v[e], pv = p[e];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mv[k] += (gv[k] - mv[k]) * b1;                  // b1, b2 hold (1 - beta): keras  m.assign_add((g - m) * (1 - beta_1))
            vv[k] += (gv[k] * gv[k] - vv[k]) * b2;
            pv[k] -= alpha * mv[k] / (sqrtf(vv[k]) + eps);
        }
        m[e] = mv;
        v[e] = vv;
        p[e] = pv;
    }
}

inline int loss_blocks(long count) {
    long b = (count + EW_BLOCK * 4 - 1) / (EW_BLOCK * 4);
    if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

// dtype dispatch of the activation storage type: T = float / _Float16 / __bf16 inside the statement
#define SS_DT(dtype, ...)                                                                                     \
    switch (dtype) {                                                                                          \
        case SS_DTYPE_F32: { typedef float T; __VA_ARGS__; } break;                                          \
        case SS_DTYPE_F16: { typedef _Float16 T; __VA_ARGS__; } break;                                       \
        case SS_DTYPE_BF16: { typedef __bf16 T; __VA_ARGS__; } break;                                        \
        default: ss_set_error("unknown ss_dtype %d", (int)(dtype)); return SS_ERR_INVALID;                   \
    }

int ss_convert_launch(const void* src, int src_dtype, int src_cs, void* dst, int dst_dtype, int dst_cs, long rows, int c, hipStream_t s) {
    if (!src || !dst || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    const dim3 grid(ew_grid(rows * c)), block(EW_BLOCK);
#define CV(TS, TD) hipLaunchKernelGGL((convert_kernel<TS, TD>), grid, block, 0, s, (const TS*)src, src_cs, (TD*)dst, dst_cs, rows, c)
    if (src_dtype == SS_DTYPE_F32 && dst_dtype == SS_DTYPE_F32) CV(float, float);
    else if (src_dtype == SS_DTYPE_F32 && dst_dtype == SS_DTYPE_F16) CV(float, _Float16);
    else if (src_dtype == SS_DTYPE_F32 && dst_dtype == SS_DTYPE_BF16) CV(float, __bf16);
    else if (src_dtype == SS_DTYPE_F16 && dst_dtype == SS_DTYPE_F32) CV(_Float16, float);
    else if (src_dtype == SS_DTYPE_BF16 && dst_dtype == SS_DTYPE_F32) CV(__bf16, float);
    else if (src_dtype == SS_DTYPE_F16 && dst_dtype == SS_DTYPE_F16) CV(_Float16, _Float16);
    else if (src_dtype == SS_DTYPE_BF16 && dst_dtype == SS_DTYPE_BF16) CV(__bf16, __bf16);
    else { ss_set_error("ss_convert: %d -> %d not supported", src_dtype, dst_dtype); return SS_ERR_UNSUPPORTED; }
#undef CV
    SS_LAUNCH_CHECK();
    return SS_OK;
}

extern "C" {

int ss_convert(const void* src, int32_t src_dtype, int32_t src_cstride, void* dst, int32_t dst_dtype, int32_t dst_cstride,
               int64_t rows, int32_t c, void* stream) {
    return ss_convert_launch(src, src_dtype, src_cstride, dst, dst_dtype, dst_cstride, (long)rows, c, (hipStream_t)stream);
}

int ss_act_bwd_t(int32_t dtype, int act, float act_alpha, const void* dy, int32_t dy_cstride, const void* y, int32_t y_cstride,
                 void* dx, int32_t dx_cstride, int64_t rows, int32_t c, void* stream) {
    if (!dy || !y || !dx || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    SS_DT(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(rows * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    act, act_alpha, (const T*)dy, dy_cstride, (const T*)y, y_cstride, (T*)dx, dx_cstride, (long)rows, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_axpby_t(int32_t dtype, float alpha, const void* a, int32_t a_cstride, float beta, const void* b, int32_t b_cstride,
               void* out, int32_t out_cstride, int64_t rows, int32_t c, void* stream) {
    if (!a || !out || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    SS_DT(dtype, hipLaunchKernelGGL(axpby_kernel<T>, dim3(ew_grid(rows * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    alpha, (const T*)a, a_cstride, beta, (const T*)b, b_cstride, (T*)out, out_cstride, (long)rows, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_mul_t(int32_t dtype, float scale, const void* a, int32_t a_cstride, const void* b, int32_t b_cstride, void* out, int32_t out_cstride,
             int64_t rows, int32_t c, void* stream) {
    if (!a || !b || !out || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    SS_DT(dtype, hipLaunchKernelGGL(mul_kernel<T>, dim3(ew_grid(rows * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    scale, (const T*)a, a_cstride, (const T*)b, b_cstride, (T*)out, out_cstride, (long)rows, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_wgan_interpolate(const float* real, const float* fake, const float* alpha, float* out, int64_t n, int64_t per_sample, void* stream) {
    if (!real || !fake || !alpha || !out || n < 0 || per_sample <= 0) return SS_ERR_INVALID;
    if (n == 0) return SS_OK;
    hipLaunchKernelGGL(interpolate_kernel, dim3(ew_grid(n * per_sample)), dim3(EW_BLOCK), 0, (hipStream_t)stream, real, fake, alpha, out,
                       (long)n, (long)per_sample);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_wgan_gp_grad(const float* g, int64_t n, int64_t per_sample, float coef, float* gbar, float* norms, void* stream) {
    if (!g || !norms || n < 0 || per_sample <= 0) return SS_ERR_INVALID;
    if (n == 0) return SS_OK;
    hipLaunchKernelGGL(gp_grad_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, g, (long)per_sample, coef, gbar, norms);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_copy_t(int32_t dtype, const void* src, int32_t src_cstride, void* dst, int32_t dst_cstride, int64_t rows, int32_t c, void* stream) {
    return ss_axpby_t(dtype, 1.f, src, src_cstride, 0.f, nullptr, 0, dst, dst_cstride, rows, c, stream);
}

int ss_pool_query(void* pool, const void* images, void* out, int64_t bytes_per_image, int32_t k, const int32_t* mode,
                  const int32_t* slot, int32_t pool_size, void* stream) {
    if (!images || !out || !mode || !slot || bytes_per_image <= 0 || k < 0 || k > SS_POOL_MAX_QUERY) return SS_ERR_INVALID;
    if (k == 0) return SS_OK;
    PoolPlan plan{};
    for (int i = 0; i < k; ++i) {
        if (mode[i] < SS_POOL_PASS || mode[i] > SS_POOL_SWAP) return SS_ERR_INVALID;
        if (mode[i] != SS_POOL_PASS) {
            if (!pool || slot[i] < 0 || slot[i] >= pool_size) { ss_set_error("ss_pool_query: slot %d outside the buffer of %d images", slot[i], pool_size); return SS_ERR_INVALID; }
            for (int j = 0; j < i; ++j)
                if (mode[j] != SS_POOL_PASS && slot[j] == slot[i]) { ss_set_error("ss_pool_query: two images of one query name slot %d", slot[i]); return SS_ERR_INVALID; }
        }
        plan.mode[i] = mode[i];
        plan.slot[i] = slot[i];
    }
    const bool v16 = bytes_per_image % 16 == 0 && (((uintptr_t)pool | (uintptr_t)images | (uintptr_t)out) & 15) == 0;
    const long units = v16 ? bytes_per_image / 16 : bytes_per_image;
    const unsigned gx = (unsigned)((units + EW_BLOCK - 1) / EW_BLOCK > 4096 ? 4096 : (units + EW_BLOCK - 1) / EW_BLOCK);
    if (v16) hipLaunchKernelGGL(pool_query_kernel<f32x4>, dim3(gx, k), dim3(EW_BLOCK), 0, (hipStream_t)stream, (f32x4*)pool, (const f32x4*)images, (f32x4*)out, units, plan);
    else hipLaunchKernelGGL(pool_query_kernel<unsigned char>, dim3(gx, k), dim3(EW_BLOCK), 0, (hipStream_t)stream, (unsigned char*)pool, (const unsigned char*)images, (unsigned char*)out, units, plan);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_zero(void* dst, size_t bytes, void* stream) {
    if (!dst && bytes) return SS_ERR_INVALID;
    if (bytes && hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream) != hipSuccess) { ss_set_error("hipMemsetAsync failed"); return SS_ERR_LAUNCH; }
    return SS_OK;
}

int ss_fill(float* dst, float value, int64_t count, void* stream) {
    if (!dst || count < 0) return SS_ERR_INVALID;
    if (count == 0) return SS_OK;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(count)), dim3(EW_BLOCK), 0, (hipStream_t)stream, dst, value, (long)count);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_maxpool2x2_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride,
                        int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!x || !y || n <= 0 || h < 2 || w < 2 || c <= 0) return SS_ERR_INVALID;
    const long total = (long)n * (h / 2) * (w / 2) * c;
    SS_DT(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)x, x_cstride, (T*)y, y_cstride, n, h, w, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_maxpool2x2_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, const void* x, int32_t x_cstride,
                        void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    if (!dy || !x || !dx || n <= 0 || h < 2 || w < 2 || c <= 0) return SS_ERR_INVALID;
    const long total = (long)n * h * w * c;
    SS_DT(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)dy, dy_cstride, (const T*)x, x_cstride, (T*)dx, dx_cstride, accumulate, n, h, w, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_reflect_pad2d_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w,
                           int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0 || pad_top < 0 || pad_bottom < 0 || pad_left < 0 || pad_right < 0) return SS_ERR_INVALID;
    if (pad_top >= h || pad_bottom >= h || pad_left >= w || pad_right >= w) return SS_ERR_INVALID;
    const int PH = h + pad_top + pad_bottom, PW = w + pad_left + pad_right;
    SS_DT(dtype, hipLaunchKernelGGL(reflect_pad_fwd_kernel<T>, dim3(ew_grid((long)n * PH * PW * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)x, x_cstride, (T*)y, y_cstride, n, h, w, c, pad_top, pad_left, PH, PW));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_reflect_pad2d_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n,
                           int32_t h, int32_t w, int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream) {
    if (!dy || !dx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return SS_ERR_INVALID;
    const int PH = h + pad_top + pad_bottom, PW = w + pad_left + pad_right;
    SS_DT(dtype, hipLaunchKernelGGL(reflect_pad_bwd_kernel<T>, dim3(ew_grid((long)n * h * w * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)dy, dy_cstride, (T*)dx, dx_cstride, accumulate, n, h, w, c, pad_top, pad_left, PH, PW));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_crop2d_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                    int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || top < 0 || left < 0 || oh <= 0 || ow <= 0 || top + oh > h || left + ow > w) return SS_ERR_INVALID;
    SS_DT(dtype, hipLaunchKernelGGL((crop_kernel<T, 0>), dim3(ew_grid((long)n * oh * ow * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)x, x_cstride, (T*)y, y_cstride, 0, n, h, w, c, top, left, oh, ow));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_crop2d_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h,
                    int32_t w, int32_t c, int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream) {
    if (!dy || !dx || n <= 0 || c <= 0 || top < 0 || left < 0 || oh <= 0 || ow <= 0 || top + oh > h || left + ow > w) return SS_ERR_INVALID;
    SS_DT(dtype, hipLaunchKernelGGL((crop_kernel<T, 1>), dim3(ew_grid((long)n * h * w * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)dy, dy_cstride, (T*)dx, dx_cstride, accumulate, n, h, w, c, top, left, oh, ow));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_upsample2x_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w,
                        int32_t c, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || c <= 0) return SS_ERR_INVALID;
    SS_DT(dtype, hipLaunchKernelGGL((upsample2x_kernel<T, 0>), dim3(ew_grid((long)n * 4 * h * w * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)x, x_cstride, (T*)y, y_cstride, 0, n, h, w, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_upsample2x_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, void* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h,
                        int32_t w, int32_t c, void* stream) {
    if (!dy || !dx || n <= 0 || h <= 0 || w <= 0 || c <= 0) return SS_ERR_INVALID;
    SS_DT(dtype, hipLaunchKernelGGL((upsample2x_kernel<T, 1>), dim3(ew_grid((long)n * h * w * c)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                                    (const T*)dy, dy_cstride, (T*)dx, dx_cstride, accumulate, n, h, w, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

size_t ss_loss_workspace_bytes(int64_t count) { (void)count; return (size_t)LOSS_MAX_BLOCKS * 3 * sizeof(float); }

int ss_loss_mse_const_t(int32_t dtype, const void* pred, int64_t count, float target, float grad_scale,
                        float* loss_out, void* grad, void* ws, size_t ws_bytes, void* stream) {
    if (!pred || !loss_out || count <= 0) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_loss_workspace_bytes(count)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(count);
    SS_DT(dtype, hipLaunchKernelGGL(mse_const_kernel<T>, dim3(nb), dim3(EW_BLOCK), 0, s, (const T*)pred, (long)count, target,
                                    grad_scale / (float)count, (T*)grad, (float*)ws));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel<1>, dim3(1), dim3(EW_BLOCK), 0, s, (const float*)ws, nb, 1.f / (float)count, loss_out);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_loss_mae_t(int32_t dtype, const void* truth, const void* pred, int64_t count, float grad_scale,
                  float* loss_out, void* grad, void* ws, size_t ws_bytes, void* stream) {
    if (!truth || !pred || !loss_out || count <= 0) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_loss_workspace_bytes(count)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(count);
    SS_DT(dtype, hipLaunchKernelGGL(mae_kernel<T>, dim3(nb), dim3(EW_BLOCK), 0, s, (const T*)truth, (const T*)pred, (long)count,
                                    grad_scale / (float)count, (T*)grad, (float*)ws));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel<1>, dim3(1), dim3(EW_BLOCK), 0, s, (const float*)ws, nb, 1.f / (float)count, loss_out);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_loss_weighted_bce_t(int32_t dtype, const void* truth, const void* pred, int64_t count, float weighting, float grad_scale,
                           float* out3, void* grad, void* ws, size_t ws_bytes, void* stream) {
    if (!truth || !pred || !out3 || count <= 0) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_loss_workspace_bytes(count)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(count);
    SS_DT(dtype, hipLaunchKernelGGL(wbce_kernel<T>, dim3(nb), dim3(EW_BLOCK), 0, s, (const T*)truth, (const T*)pred, (long)count, weighting,
                                    grad_scale / (float)count, (T*)grad, (float*)ws));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel<3>, dim3(1), dim3(EW_BLOCK), 0, s, (const float*)ws, nb, 1.f / (float)count, out3);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_softmax_fwd_t(int32_t dtype, const void* x, int32_t x_cstride, void* y, int32_t y_cstride, int64_t rows, int32_t c, void* stream) {
    if (!x || !y || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    SS_DT(dtype, hipLaunchKernelGGL(softmax_fwd_kernel<T>, dim3(ew_grid(rows)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)x, x_cstride,
                                    (T*)y, y_cstride, (long)rows, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_softmax_bwd_t(int32_t dtype, const void* dy, int32_t dy_cstride, const void* y, int32_t y_cstride, void* dx, int32_t dx_cstride,
                     int64_t rows, int32_t c, void* stream) {
    if (!dy || !y || !dx || rows < 0 || c <= 0) return SS_ERR_INVALID;
    if (rows == 0) return SS_OK;
    SS_DT(dtype, hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3(ew_grid(rows)), dim3(EW_BLOCK), 0, (hipStream_t)stream, (const T*)dy, dy_cstride,
                                    (const T*)y, y_cstride, (T*)dx, dx_cstride, (long)rows, c));
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_loss_weighted_bce_mc_t(int32_t dtype, const void* truth, const void* pred, int64_t rows, int32_t c, float weighting, float grad_scale,
                              float* out3, void* grad, void* ws, size_t ws_bytes, void* stream) {
    if (!truth || !pred || !out3 || rows <= 0 || c <= 0) return SS_ERR_INVALID;
    const int64_t count = rows * c;
    if (!ws || ws_bytes < ss_loss_workspace_bytes(count)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(rows);
    SS_DT(dtype, hipLaunchKernelGGL(wbce_mc_kernel<T>, dim3(nb), dim3(EW_BLOCK), 0, s, (const T*)truth, (const T*)pred, (long)rows, c, weighting,
                                    grad_scale / (float)count, (T*)grad, (float*)ws));
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel<3>, dim3(1), dim3(EW_BLOCK), 0, s, (const float*)ws, nb, 1.f / (float)count, out3);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// ---- the fp32 entry points of the original ABI ----
int ss_act_bwd(int act, float act_alpha, const float* dy, int32_t dy_cstride, const float* y, int32_t y_cstride,
               float* dx, int32_t dx_cstride, int64_t rows, int32_t c, void* stream) {
    return ss_act_bwd_t(SS_DTYPE_F32, act, act_alpha, dy, dy_cstride, y, y_cstride, dx, dx_cstride, rows, c, stream);
}
int ss_axpby(float alpha, const float* a, int32_t a_cstride, float beta, const float* b, int32_t b_cstride,
             float* out, int32_t out_cstride, int64_t rows, int32_t c, void* stream) {
    return ss_axpby_t(SS_DTYPE_F32, alpha, a, a_cstride, beta, b, b_cstride, out, out_cstride, rows, c, stream);
}
int ss_copy(const float* src, int32_t src_cstride, float* dst, int32_t dst_cstride, int64_t rows, int32_t c, void* stream) {
    return ss_copy_t(SS_DTYPE_F32, src, src_cstride, dst, dst_cstride, rows, c, stream);
}
int ss_maxpool2x2_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    return ss_maxpool2x2_fwd_t(SS_DTYPE_F32, x, x_cstride, y, y_cstride, n, h, w, c, stream);
}
int ss_maxpool2x2_bwd(const float* dy, int32_t dy_cstride, const float* x, int32_t x_cstride, float* dx, int32_t dx_cstride, int accumulate,
                      int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    return ss_maxpool2x2_bwd_t(SS_DTYPE_F32, dy, dy_cstride, x, x_cstride, dx, dx_cstride, accumulate, n, h, w, c, stream);
}
int ss_reflect_pad2d_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                         int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream) {
    return ss_reflect_pad2d_fwd_t(SS_DTYPE_F32, x, x_cstride, y, y_cstride, n, h, w, c, pad_top, pad_bottom, pad_left, pad_right, stream);
}
int ss_reflect_pad2d_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                         int32_t c, int32_t pad_top, int32_t pad_bottom, int32_t pad_left, int32_t pad_right, void* stream) {
    return ss_reflect_pad2d_bwd_t(SS_DTYPE_F32, dy, dy_cstride, dx, dx_cstride, accumulate, n, h, w, c, pad_top, pad_bottom, pad_left, pad_right, stream);
}
int ss_crop2d_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c,
                  int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream) {
    return ss_crop2d_fwd_t(SS_DTYPE_F32, x, x_cstride, y, y_cstride, n, h, w, c, top, left, oh, ow, stream);
}
int ss_crop2d_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                  int32_t c, int32_t top, int32_t left, int32_t oh, int32_t ow, void* stream) {
    return ss_crop2d_bwd_t(SS_DTYPE_F32, dy, dy_cstride, dx, dx_cstride, accumulate, n, h, w, c, top, left, oh, ow, stream);
}
int ss_upsample2x_fwd(const float* x, int32_t x_cstride, float* y, int32_t y_cstride, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
    return ss_upsample2x_fwd_t(SS_DTYPE_F32, x, x_cstride, y, y_cstride, n, h, w, c, stream);
}
int ss_upsample2x_bwd(const float* dy, int32_t dy_cstride, float* dx, int32_t dx_cstride, int accumulate, int32_t n, int32_t h, int32_t w,
                      int32_t c, void* stream) {
    return ss_upsample2x_bwd_t(SS_DTYPE_F32, dy, dy_cstride, dx, dx_cstride, accumulate, n, h, w, c, stream);
}
int ss_loss_mse_const(const float* pred, int64_t count, float target, float grad_scale,
                      float* loss_out, float* grad, void* ws, size_t ws_bytes, void* stream) {
    return ss_loss_mse_const_t(SS_DTYPE_F32, pred, count, target, grad_scale, loss_out, grad, ws, ws_bytes, stream);
}
int ss_loss_mae(const float* truth, const float* pred, int64_t count, float grad_scale,
                float* loss_out, float* grad, void* ws, size_t ws_bytes, void* stream) {
    return ss_loss_mae_t(SS_DTYPE_F32, truth, pred, count, grad_scale, loss_out, grad, ws, ws_bytes, stream);
}
int ss_loss_weighted_bce(const float* truth, const float* pred, int64_t count, float weighting, float grad_scale,
                         float* out3, float* grad, void* ws, size_t ws_bytes, void* stream) {
    return ss_loss_weighted_bce_t(SS_DTYPE_F32, truth, pred, count, weighting, grad_scale, out3, grad, ws, ws_bytes, stream);
}

static int adam_launch(float* p, const float* g, float* m, float* v, int64_t count, float alpha, const float* alpha_dev,
                       double beta1_d, double beta2_d, double eps_d, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || count < 0) return SS_ERR_INVALID;
    if (count == 0) return SS_OK;
    // Keras forms (1 - beta) in python double precision and casts the RESULT to the variable dtype: 1 - 0.999 -> fp32(0.001), whereas
    // 1.f - fp32(0.999) = 0.00099998713 (1.3e-5 off; found by tests/test_direct_gpu.py::test_adam_keras_ten_iterations_vs_oracle)
    const float eps = (float)eps_d;
    const float beta1 = (float)(1.0 - beta1_d), beta2 = (float)(1.0 - beta2_d);
    hipStream_t s = (hipStream_t)stream;
    const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const long c4 = al ? count / 4 : 0;
    if (c4 > 0) {
        hipLaunchKernelGGL(adam_kernel_v4, dim3(ew_grid(c4)), dim3(EW_BLOCK), 0, s, (f32x4*)p, (const f32x4*)g, (f32x4*)m, (f32x4*)v,
                           c4, alpha, beta1, beta2, eps, grad_scale, alpha_dev);
        SS_LAUNCH_CHECK();
    }
    const long rem = count - c4 * 4;
    if (rem > 0) {
        hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(rem)), dim3(EW_BLOCK), 0, s, p + c4 * 4, g + c4 * 4, m + c4 * 4, v + c4 * 4,
                           rem, alpha, beta1, beta2, eps, grad_scale, alpha_dev);
        SS_LAUNCH_CHECK();
    }
    return SS_OK;
}

int ss_adam_keras(float* p, const float* g, float* m, float* v, int64_t count,
                  double alpha_d, double beta1_d, double beta2_d, double eps_d, float grad_scale, void* stream) {
    return adam_launch(p, g, m, v, count, (float)alpha_d, nullptr, beta1_d, beta2_d, eps_d, grad_scale, stream);
}

int ss_adam_keras_dev(float* p, const float* g, float* m, float* v, int64_t count,
                      const float* alpha_dev, double beta1_d, double beta2_d, double eps_d, float grad_scale, void* stream) {
    if (!alpha_dev) return SS_ERR_INVALID;
    return adam_launch(p, g, m, v, count, 0.f, alpha_dev, beta1_d, beta2_d, eps_d, grad_scale, stream);
}

}  // extern "C"
