// InstanceNorm (GroupNormalization groups=-1) and BatchNorm, forward / inference / backward, NHWC views.
// A tensor is seen as [G groups][P pixels][C channels]: G = n for instance norm, G = 1 (P = n*h*w) for batch norm.
// HBM-bound: statistics pass (read x), apply pass (read x, write y); backward: statistics pass
// (read dy, x[, y]) and apply pass (read dy, x[, y], write dx[, dres]).
// Statistics: per-thread fp32 partial sums over a pixel chunk, wave/LDS reduction over the pixel lanes,
// per-chunk partials in the workspace, fp64 combine in the finalize kernel.
#include "common.h"

namespace {

constexpr int NORM_MAX_CHUNKS = 128;

struct NormGeom {
    int G, C, CT, PT, chunks;   // CT channel lanes (pow2 <= 64), PT = 256/CT pixel lanes
    long P, pix_per_chunk;
};

NormGeom geom(const ss_norm_desc* d) {
    NormGeom g;
    g.G = d->groups;
    g.C = d->c;
    g.P = (long)d->n * d->h * d->w / d->groups;
    int ct = 1;
    while (ct < d->c && ct < 64) ct <<= 1;
    g.CT = ct;
    g.PT = 256 / ct;
    long chunks = (g.P + 1023) / 1024;
    long want = 2048 / ((long)g.G * ((g.C + ct - 1) / ct));   // aim at >= ~2048 blocks in total
    if (want < 1) want = 1;
    if (chunks > want) chunks = want;
    if (chunks > NORM_MAX_CHUNKS) chunks = NORM_MAX_CHUNKS;
    if (chunks < 1) chunks = 1;
    g.chunks = (int)chunks;
    g.pix_per_chunk = (g.P + chunks - 1) / chunks;
    return g;
}

// partial sums: part[((g*chunks + chunk)*C + c)*2 + {0,1}]
// MODE 0: (sum x, sum x^2)        MODE 1: (sum g, sum g*xhat) with g = dy * act'(y)
template <int MODE>
__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ dy, int dy_cs,
                                                         const float* __restrict__ y, int y_cs,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         int act, float alpha,
                                                         int C, long P, long pix_per_chunk, int CT, int PT,
                                                         float* __restrict__ part) {
    __shared__ float red[2][256];
    const int ct = threadIdx.x % CT, pt = threadIdx.x / CT;
    const int c = blockIdx.y * CT + ct;
    const int g = blockIdx.z;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = (p0 + pix_per_chunk < P) ? p0 + pix_per_chunk : P;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const long base = (long)g * P;
        float mu = 0.f, rs = 0.f;
        if (MODE == 1) { mu = mean[(long)g * C + c]; rs = rstd[(long)g * C + c]; }
        for (long p = p0 + pt; p < p1; p += PT) {
            const float xv = x[(base + p) * x_cs + c];
            if (MODE == 0) {
                s1 += xv;
                s2 = fmaf(xv, xv, s2);
            } else {
                float gv = dy[(base + p) * dy_cs + c];
                if (act != SS_ACT_NONE) gv *= ss_act_grad_from_out(y[(base + p) * y_cs + c], act, alpha);
                s1 += gv;
                s2 = fmaf(gv, (xv - mu) * rs, s2);
            }
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int off = PT / 2; off >= 1; off >>= 1) {
        if (pt < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off * CT];
            red[1][threadIdx.x] += red[1][threadIdx.x + off * CT];
        }
        __syncthreads();
    }
    if (pt == 0 && c < C) {
        float* o = part + (((long)g * gridDim.x + blockIdx.x) * C + c) * 2;
        o[0] = red[0][threadIdx.x];
        o[1] = red[1][threadIdx.x];
    }
}

// mean / rstd per (g,c); optional moving-average update (batch norm, G == 1)
__global__ __launch_bounds__(256) void norm_finalize_fwd(const float* __restrict__ part, int chunks, int G, int C, long P, float eps,
                                                         float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ mm, float* __restrict__ mv, float momentum) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)G * C) return;
    const int g = (int)(i / C), c = (int)(i % C);
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const float* o = part + (((long)g * chunks + k) * C + c) * 2;
        s1 += o[0];
        s2 += o[1];
    }
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

// y = act((x-mean)*rstd*gamma + beta + residual)
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ res, int res_cs,
                                                         float* __restrict__ y, int y_cs,
                                                         int act, float alpha, int C, long P, long rows) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const int c = (int)(e % C);
    const long row = e / C;
    const long gi = (row / P) * C + c;
    const float sc = rstd[gi] * (gamma ? gamma[c] : 1.f);
    float v = (x[row * x_cs + c] - mean[gi]) * sc + beta[c];
    if (res) v += res[row * res_cs + c];
    y[row * y_cs + c] = ss_apply_act(v, act, alpha);
}

// inference: statistics from moving mean / variance
__global__ __launch_bounds__(256) void norm_infer_kernel(const float* __restrict__ x, int x_cs,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mm, const float* __restrict__ mv, float eps,
                                                         const float* __restrict__ res, int res_cs,
                                                         float* __restrict__ y, int y_cs, int act, float alpha, int C, long rows) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const int c = (int)(e % C);
    const long row = e / C;
    const float sc = rsqrtf(mv[c] + eps) * (gamma ? gamma[c] : 1.f);
    float v = (x[row * x_cs + c] - mm[c]) * sc + beta[c];
    if (res) v += res[row * res_cs + c];
    y[row * y_cs + c] = ss_apply_act(v, act, alpha);
}

// backward finalize: per (g,c) means of g and g*xhat -> sg/sgx arrays; dgamma/dbeta summed over groups
__global__ __launch_bounds__(256) void norm_finalize_bwd(const float* __restrict__ part, int chunks, int G, int C, long P,
                                                         float* __restrict__ sums /* [G*C*2] */,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double tg = 0.0, tgx = 0.0;
    for (int g = 0; g < G; ++g) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < chunks; ++k) {
            const float* o = part + (((long)g * chunks + k) * C + c) * 2;
            s1 += o[0];
            s2 += o[1];
        }
        sums[((long)g * C + c) * 2 + 0] = (float)(s1 / (double)P);
        sums[((long)g * C + c) * 2 + 1] = (float)(s2 / (double)P);
        tg += s1;
        tgx += s2;
    }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)tg : (float)tg;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)tgx : (float)tgx;
}

// dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) ; dres = g
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ dy, int dy_cs,
                                                             const float* __restrict__ x, int x_cs,
                                                             const float* __restrict__ y, int y_cs,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ sums,
                                                             float* __restrict__ dx, int dx_cs, int acc_dx,
                                                             float* __restrict__ dres, int dres_cs, int acc_dres,
                                                             int act, float alpha, int C, long P, long rows) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * C) return;
    const int c = (int)(e % C);
    const long row = e / C;
    const long gi = (row / P) * C + c;
    float gv = dy[row * dy_cs + c];
    if (act != SS_ACT_NONE) gv *= ss_act_grad_from_out(y[row * y_cs + c], act, alpha);
    const float rs = rstd[gi];
    const float xh = (x[row * x_cs + c] - mean[gi]) * rs;
    const float sc = rs * (gamma ? gamma[c] : 1.f);
    const float dv = sc * (gv - sums[gi * 2] - xh * sums[gi * 2 + 1]);
    float* o = dx + row * dx_cs + c;
    *o = acc_dx ? (*o + dv) : dv;
    if (dres) {
        float* r = dres + row * dres_cs + c;
        *r = acc_dres ? (*r + gv) : gv;
    }
}

bool valid(const ss_norm_desc* d) {
    if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0 || d->c <= 0) return false;
    if (d->groups != 1 && d->groups != d->n) return false;
    if (d->x_cstride < d->c || d->y_cstride < d->c) return false;
    return true;
}

}  // namespace

extern "C" {

size_t ss_norm_workspace_bytes(const ss_norm_desc* d) {
    if (!valid(d)) return 0;
    const NormGeom g = geom(d);
    return ss_align_up((size_t)g.G * g.chunks * g.C * 2 * sizeof(float), 256) + ss_align_up((size_t)g.G * g.C * 2 * sizeof(float), 256);
}

int ss_norm_fwd(const ss_norm_desc* d, const float* x, const float* gamma, const float* beta,
                const float* residual, float* y, float* mean, float* rstd,
                float* moving_mean, float* moving_var, float momentum,
                void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !x || !beta || !y || !mean || !rstd) return SS_ERR_INVALID;
    if ((moving_mean != nullptr) != (moving_var != nullptr)) return SS_ERR_INVALID;
    if (moving_mean && d->groups != 1) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const NormGeom g = geom(d);
    float* part = (float*)ws;
    hipLaunchKernelGGL(norm_stats_kernel<0>, dim3(g.chunks, (g.C + g.CT - 1) / g.CT, g.G), dim3(256), 0, s,
                       x, d->x_cstride, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, 0.f,
                       g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    SS_LAUNCH_CHECK();
    const long gc = (long)g.G * g.C;
    hipLaunchKernelGGL(norm_finalize_fwd, dim3((unsigned)((gc + 255) / 256)), dim3(256), 0, s,
                       part, g.chunks, g.G, g.C, g.P, d->eps, mean, rstd, moving_mean, moving_var, momentum);
    SS_LAUNCH_CHECK();
    const long rows = (long)g.G * g.P;
    hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((rows * g.C + 255) / 256)), dim3(256), 0, s,
                       x, d->x_cstride, gamma, beta, mean, rstd, residual, d->res_cstride, y, d->y_cstride,
                       d->act, d->act_alpha, g.C, g.P, rows);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_norm_infer(const ss_norm_desc* d, const float* x, const float* gamma, const float* beta,
                  const float* moving_mean, const float* moving_var, const float* residual, float* y, void* stream) {
    if (!valid(d) || !x || !beta || !y || !moving_mean || !moving_var) return SS_ERR_INVALID;
    const long rows = (long)d->n * d->h * d->w;
    hipLaunchKernelGGL(norm_infer_kernel, dim3((unsigned)((rows * d->c + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, d->x_cstride, gamma, beta, moving_mean, moving_var, d->eps, residual, d->res_cstride,
                       y, d->y_cstride, d->act, d->act_alpha, d->c, rows);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

int ss_norm_bwd(const ss_norm_desc* d, const float* dy, int32_t dy_cstride, const float* x, const float* y,
                const float* gamma, const float* mean, const float* rstd,
                float* dx, int32_t dx_cstride, int accumulate_dx, float* dres, int accumulate_dres,
                float* dgamma, float* dbeta, int accumulate_params,
                void* ws, size_t ws_bytes, void* stream) {
    if (!valid(d) || !dy || !x || !mean || !rstd || !dx) return SS_ERR_INVALID;
    if (d->act != SS_ACT_NONE && !y) return SS_ERR_INVALID;
    if (!ws || ws_bytes < ss_norm_workspace_bytes(d)) return SS_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const NormGeom g = geom(d);
    float* part = (float*)ws;
    float* sums = (float*)((char*)ws + ss_align_up((size_t)g.G * g.chunks * g.C * 2 * sizeof(float), 256));
    hipLaunchKernelGGL(norm_stats_kernel<1>, dim3(g.chunks, (g.C + g.CT - 1) / g.CT, g.G), dim3(256), 0, s,
                       x, d->x_cstride, dy, dy_cstride, y, d->y_cstride, mean, rstd, d->act, d->act_alpha,
                       g.C, g.P, g.pix_per_chunk, g.CT, g.PT, part);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_finalize_bwd, dim3((g.C + 255) / 256), dim3(256), 0, s,
                       part, g.chunks, g.G, g.C, g.P, sums, dgamma, dbeta, accumulate_params);
    SS_LAUNCH_CHECK();
    const long rows = (long)g.G * g.P;
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3((unsigned)((rows * g.C + 255) / 256)), dim3(256), 0, s,
                       dy, dy_cstride, x, d->x_cstride, y, d->y_cstride, gamma, mean, rstd, sums,
                       dx, dx_cstride, accumulate_dx, dres, d->res_cstride, accumulate_dres, d->act, d->act_alpha, g.C, g.P, rows);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

}  // extern "C"
