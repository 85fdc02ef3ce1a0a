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
};

// ---- small helper kernels -----------------------------------------------------------------------
// WT[t][b][a] = W[t][a][b]
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ w, float* __restrict__ wt, int A, int B) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const float* src = w + (long)t * A * B;
    float* dst = wt + (long)t * A * B;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
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

// dx[n,iy,ix,c] (+)= sum over the padded positions that reflect onto (iy,ix) of dpad[n,py,px,c]
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dpad, float* __restrict__ dx,
                                                           int N, int IH, int IW, int C, int dx_cs,
                                                           int pt, int pl, int PH, int PW, int accumulate) {
    const long total = (long)N * IH * IW * C;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C);
    long r = e / C;
    const int ix = (int)(r % IW); r /= IW;
    const int iy = (int)(r % IH);
    const int n = (int)(r / IH);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = iy + pt;
    if (iy >= 1 && pt - iy >= 0) ys[ny++] = pt - iy;
    { const int py = pt + 2 * (IH - 1) - iy; if (iy <= IH - 2 && py < PH) ys[ny++] = py; }
    xs[nx++] = ix + pl;
    if (ix >= 1 && pl - ix >= 0) xs[nx++] = pl - ix;
    { const int px = pl + 2 * (IW - 1) - ix; if (ix <= IW - 2 && px < PW) xs[nx++] = px; }
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) acc += dpad[((long)(n * PH + ys[a]) * PW + xs[b]) * C + c];
    float* o = dx + ((long)(n * IH + iy) * IW + ix) * dx_cs + c;
    *o = accumulate ? (*o + acc) : acc;
}

// column sums of a [rows][C] view: stage 1 partials[chunk][c], stage 2 final
#define COLSUM_CHUNKS 240
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ v, long rows, int C, int cs, float* __restrict__ part) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;  // 4 row lanes
    __shared__ float red[4][64];
    float acc = 0.f;
    if (c < C) {
        const long per = (rows + gridDim.x - 1) / gridDim.x;
        const long r0 = (long)blockIdx.x * per;
        const long r1 = (r0 + per < rows) ? r0 + per : rows;
        for (long r = r0 + pl; r < r1; r += 4) acc += v[r * cs + c];
    }
    red[pl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (pl == 0 && c < C)
        part[(long)blockIdx.x * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int chunks, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    for (int k = 0; k < chunks; ++k) acc += part[(long)k * C + c];
    out[c] = accumulate ? out[c] + (float)acc : (float)acc;
}

int colsum(const float* v, long rows, int C, int cs, float* out, int accumulate, float* part, hipStream_t s) {
    int chunks = (int)((rows + 255) / 256);
    if (chunks > COLSUM_CHUNKS) chunks = COLSUM_CHUNKS;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(colsum_stage1, dim3(chunks, (C + 63) / 64), dim3(256), 0, s, v, rows, C, cs, part);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 255) / 256), dim3(256), 0, s, part, chunks, C, out, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// ---- problem builders ---------------------------------------------------------------------------
bool use_mfma(int algo, const GConvParams& p) {
    if (algo == SS_ALGO_DIRECT) return false;
    if (algo == SS_ALGO_MFMA) return true;
    return ss_gconv_mfma_ok(p);
}

int run_gconv(int algo, const GConvParams& p, hipStream_t s) {
    return use_mfma(algo, p) ? ss_launch_gconv_mfma(p, s) : ss_launch_gconv_direct(p, s);
}

int conv_fwd(const ConvProb& c, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
             int accumulate, int algo, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    GConvParams p{};
    p.in = x; p.w = w; p.bias = bias; p.out = y;
    p.N = c.n; p.IH = c.ih; p.IW = c.iw; p.Cin = c.cin; p.in_cs = c.in_cs;
    p.OHc = c.oh; p.OWc = c.ow;
    p.in_s = c.s; p.in_oy = -c.pt; p.in_ox = -c.pl;
    p.OH = c.oh; p.OW = c.ow; p.Cout = c.cout; p.out_cs = c.out_cs;
    p.out_s = 1; p.out_oy = 0; p.out_ox = 0;
    p.ldb = c.cout; p.reflect = c.reflect; p.act = act; p.alpha = alpha; p.accumulate = accumulate;
    p.ntaps = 0;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = p.taps[p.ntaps++];
            t.dy = (int16_t)a; t.dx = (int16_t)b; t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    return run_gconv(algo, p, s);
}

size_t bwd_data_ws(const ConvProb& c) {
    size_t b = ss_align_up((size_t)c.kh * c.kw * c.cin * c.cout * sizeof(float), 256);
    if (c.reflect) {
        const int PH = c.oh + c.kh - 1, PW = c.ow + c.kw - 1;
        b += ss_align_up((size_t)c.n * PH * PW * c.cin * sizeof(float), 256);
    }
    return b;
}

// dx = dC/dx for the plain conv `c`; bias/act only used when this implements a transposed-conv forward
int conv_bwd_data(const ConvProb& c, const float* dy, const float* w, float* dx, const float* bias, int act, float alpha,
                  int accumulate, int algo, void* ws, size_t ws_bytes, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    if (c.reflect && c.s != 1) return SS_ERR_UNSUPPORTED;
    if (ws_bytes < bwd_data_ws(c) || !ws) return SS_ERR_WORKSPACE;
    float* wt = (float*)ws;
    const int T = c.kh * c.kw;
    hipLaunchKernelGGL(transpose_last2_kernel, dim3((c.cout + 31) / 32, (c.cin + 31) / 32, T), dim3(256), 0, s, w, wt, c.cin, c.cout);
    SS_LAUNCH_CHECK();

    GConvParams p{};
    p.in = dy; p.w = wt; p.bias = bias;
    p.N = c.n; p.IH = c.oh; p.IW = c.ow; p.Cin = c.cout; p.in_cs = c.out_cs;
    p.in_s = 1; p.Cout = c.cin; p.ldb = c.cin; p.reflect = 0; p.act = act; p.alpha = alpha;

    if (c.reflect) {
        const int PH = c.oh + c.kh - 1, PW = c.ow + c.kw - 1;
        float* dpad = (float*)((char*)ws + ss_align_up((size_t)T * c.cin * c.cout * sizeof(float), 256));
        p.out = dpad; p.OH = PH; p.OW = PW; p.out_cs = c.cin; p.OHc = PH; p.OWc = PW;
        p.out_s = 1; p.out_oy = 0; p.out_ox = 0; p.in_oy = 0; p.in_ox = 0; p.accumulate = 0;
        p.ntaps = 0;
        for (int a = 0; a < c.kh; ++a)
            for (int b = 0; b < c.kw; ++b) {
                GTap& t = p.taps[p.ntaps++];
                t.dy = (int16_t)(-a); t.dx = (int16_t)(-b); t.woff = (a * c.kw + b) * c.cin * c.cout;
            }
        int rc = run_gconv(algo, p, s);
        if (rc != SS_OK) return rc;
        const long total = (long)c.n * c.ih * c.iw * c.cin;
        hipLaunchKernelGGL(reflect_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dpad, dx,
                           c.n, c.ih, c.iw, c.cin, c.in_cs, c.pt, c.pl, PH, PW, accumulate);
        SS_LAUNCH_CHECK();
        return SS_OK;
    }

    p.out = dx; p.OH = c.ih; p.OW = c.iw; p.out_cs = c.in_cs; p.accumulate = accumulate;
    p.out_s = c.s; p.in_oy = 0; p.in_ox = 0;
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
            int rc = run_gconv(algo, p, s);
            if (rc != SS_OK) return rc;
        }
    return SS_OK;
}

size_t bwd_weight_ws(const ConvProb& c) {
    int pps;
    const long P = (long)c.n * c.oh * c.ow;
    const int M = c.kh * c.kw * c.cin;
    const int splits = ss_wgrad_mfma_splits(P, M, c.cout, &pps);
    size_t b = ss_align_up((size_t)splits * M * c.cout * sizeof(float), 256);
    b += ss_align_up((size_t)COLSUM_CHUNKS * (c.cout > c.cin ? c.cout : c.cin) * sizeof(float), 256);
    return b;
}

int conv_bwd_weight(const ConvProb& c, const float* x, const float* dy, float* dw, int accumulate, int algo,
                    void* ws, size_t ws_bytes, hipStream_t s) {
    if (c.kh * c.kw > SS_MAX_TAPS) return SS_ERR_UNSUPPORTED;
    if (ws_bytes < bwd_weight_ws(c) || !ws) return SS_ERR_WORKSPACE;
    WGradParams p{};
    p.a = x; p.b = dy; p.part = (float*)ws;
    p.N = c.n; p.AH = c.ih; p.AW = c.iw; p.Ca = c.cin; p.a_cs = c.in_cs;
    p.GH = c.oh; p.GW = c.ow; p.Cb = c.cout; p.b_cs = c.out_cs;
    p.a_s = c.s; p.a_oy = -c.pt; p.a_ox = -c.pl; p.reflect = c.reflect;
    p.ntaps = 0;
    for (int a = 0; a < c.kh; ++a)
        for (int b = 0; b < c.kw; ++b) {
            GTap& t = p.taps[p.ntaps++];
            t.dy = (int16_t)a; t.dx = (int16_t)b; t.woff = (a * c.kw + b) * c.cin * c.cout;
        }
    if (algo == SS_ALGO_DIRECT) {
        p.splits = 1; p.pix_per_split = 0;
        return ss_launch_wgrad_direct(p, dw, c.cout, accumulate, s);
    }
    int pps;
    p.splits = ss_wgrad_mfma_splits((long)c.n * c.oh * c.ow, p.ntaps * p.Ca, p.Cb, &pps);
    p.pix_per_split = pps;
    return ss_launch_wgrad_mfma(p, dw, c.cout, accumulate, s);
}

bool valid_desc(const ss_conv_desc* d) {
    if (!d) return false;
    if (d->n <= 0 || d->ih <= 0 || d->iw <= 0 || d->cin <= 0 || d->oh <= 0 || d->ow <= 0 || d->cout <= 0) return false;
    if (d->kh <= 0 || d->kw <= 0 || d->stride <= 0) return false;
    if (d->in_cstride < d->cin || d->out_cstride < d->cout) return false;
    if (d->pad_mode == SS_PAD_REFLECT && (d->stride != 1 || d->transposed)) return false;
    if (d->pad_mode == SS_PAD_REFLECT && (d->pad_top >= d->ih || d->pad_left >= d->iw)) return false;
    return true;
}

ConvProb plain(const ss_conv_desc* d) {
    return ConvProb{d->n, d->ih, d->iw, d->cin, d->in_cstride, d->oh, d->ow, d->cout, d->out_cstride,
                    d->kh, d->kw, d->stride, d->pad_top, d->pad_left, d->pad_mode == SS_PAD_REFLECT};
}
// adjoint conv of a transposed conv: output space -> input space
ConvProb adjoint(const ss_conv_desc* d) {
    return ConvProb{d->n, d->oh, d->ow, d->cout, d->out_cstride, d->ih, d->iw, d->cin, d->in_cstride,
                    d->kh, d->kw, d->stride, d->pad_top, d->pad_left, 0};
}

}  // namespace

extern "C" {

int ss_version(void) { return 100; }

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

size_t ss_conv2d_workspace_bytes(const ss_conv_desc* d, int pass) {
    if (!valid_desc(d)) return 0;
    const size_t colsum_b = ss_align_up((size_t)COLSUM_CHUNKS * (d->cout > d->cin ? d->cout : d->cin) * sizeof(float), 256);
    if (!d->transposed) {
        if (pass == SS_PASS_FWD) return 256;
        if (pass == SS_PASS_BWD_DATA) return bwd_data_ws(plain(d));
        return bwd_weight_ws(plain(d)) + colsum_b;
    }
    if (pass == SS_PASS_FWD) return bwd_data_ws(adjoint(d));
    if (pass == SS_PASS_BWD_DATA) return 256;
    return bwd_weight_ws(adjoint(d)) + colsum_b;
}

int ss_conv2d_fwd(const ss_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                  void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc(d) || !x || !w || !y) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (!d->transposed) return conv_fwd(plain(d), x, w, bias, y, d->act, d->act_alpha, 0, d->algo, s);
    return conv_bwd_data(adjoint(d), x, w, y, bias, d->act, d->act_alpha, 0, d->algo, ws, ws_bytes, s);
}

int ss_conv2d_bwd_data(const ss_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc(d) || !dy || !w || !dx) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (!d->transposed)
        return conv_bwd_data(plain(d), dy, w, dx, nullptr, SS_ACT_NONE, 0.f, accumulate, d->algo, ws, ws_bytes, s);
    return conv_fwd(adjoint(d), dy, w, nullptr, dx, SS_ACT_NONE, 0.f, accumulate, d->algo, s);
}

int ss_conv2d_bwd_weight(const ss_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                         int accumulate, void* ws, size_t ws_bytes, void* stream) {
    if (!valid_desc(d) || !x || !dy || !dw) return SS_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (ws_bytes < ss_conv2d_workspace_bytes(d, SS_PASS_BWD_WEIGHT) || !ws) return SS_ERR_WORKSPACE;
    int rc;
    const ConvProb c = d->transposed ? adjoint(d) : plain(d);
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

}  // extern "C"
