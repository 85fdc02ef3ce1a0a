// Winograd F(2x2, 3x3) path for the stride-1 3x3 convolutions with wide channels (the generator trunk:
// 85 % of the CycleGAN FLOPs).  2.25x fewer multiply-adds than the direct implicit GEMM, all in fp32:
//   V = B^T d B   (4x4 input patch per 2x2 output tile; reflect / zero padding applied in the gather)
//   U = G g G^T   (weights, per step)
//   M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co]      16 independent GEMMs -> batched gconv_mfma (fp32 MFMA)
//   Y = A^T M A
// and for the weight gradient (F(3x3, 2x2)):
//   S_xi[ci][co] = sum_tiles V_xi[tile][ci] * E_xi[tile][co],  E = A e A^T (2x2 dy tile),   dW = G^T S G.
// The transforms are streaming (HBM-bound) kernels: one thread = one tile x 4 channels.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ f32x4 ldz(const float* base, long off, bool ok) {
    const f32x4 v = *(const f32x4*)(base + (ok ? off : 0));
    return ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
}

// V[xi][tile][c], tile = (n, ty, tx); patch d[i][j] = in[n, map(2ty + i - pt), map(2tx + j - pl), c]
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ in, int in_cs, int N, int H, int W, int C,
                                                         int TH, int TW, int pt, int pl, int reflect, float* __restrict__ V) {
    const int C4 = C / 4;
    const long tiles = (long)N * TH * TW;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tiles * C4) return;
    const int c = (int)(e % C4) * 4;
    const long tile = e / C4;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    int iy[4], ix[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        iy[i] = ss_map_index(2 * ty + i - pt, H, reflect);
        ix[i] = ss_map_index(2 * tx + i - pl, W, reflect);
    }
    f32x4 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = iy[i] >= 0 && ix[j] >= 0;
            d[i] = ldz(in, ((long)(n * H + iy[i]) * W + ix[j]) * in_cs + c, ok);
        }
        t[0][j] = d[0] - d[2];
        t[1][j] = d[1] + d[2];
        t[2][j] = d[2] - d[1];
        t[3][j] = d[1] - d[3];
    }
    const long xs = tiles * C;     // stride between transform positions
    float* o = V + tile * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(f32x4*)(o + (long)(i * 4 + 0) * xs) = t[i][0] - t[i][2];
        *(f32x4*)(o + (long)(i * 4 + 1) * xs) = t[i][1] + t[i][2];
        *(f32x4*)(o + (long)(i * 4 + 2) * xs) = t[i][2] - t[i][1];
        *(f32x4*)(o + (long)(i * 4 + 3) * xs) = t[i][1] - t[i][3];
    }
}

// E[xi][tile][c] = (A e A^T)_xi for the 2x2 tile e of dy (zero outside the dy extent)
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, int dy_cs, int N, int OH, int OW, int C,
                                                      int TH, int TW, float* __restrict__ E) {
    const int C4 = C / 4;
    const long tiles = (long)N * TH * TW;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tiles * C4) return;
    const int c = (int)(e % C4) * 4;
    const long tile = e / C4;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    f32x4 v[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = 2 * ty + i, ox = 2 * tx + j;
            v[i][j] = ldz(dy, ((long)(n * OH + oy) * OW + ox) * dy_cs + c, oy < OH && ox < OW);
        }
    // A = [[1,0],[1,1],[1,-1],[0,-1]]  (4x2):  rows of A e, then (A e) A^T
    f32x4 a[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        a[0][j] = v[0][j];
        a[1][j] = v[0][j] + v[1][j];
        a[2][j] = v[0][j] - v[1][j];
        a[3][j] = -v[1][j];
    }
    const long xs = tiles * C;
    float* o = E + tile * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(f32x4*)(o + (long)(i * 4 + 0) * xs) = a[i][0];
        *(f32x4*)(o + (long)(i * 4 + 1) * xs) = a[i][0] + a[i][1];
        *(f32x4*)(o + (long)(i * 4 + 2) * xs) = a[i][0] - a[i][1];
        *(f32x4*)(o + (long)(i * 4 + 3) * xs) = -a[i][1];
    }
}

// U[xi][kr][no] = (G g G^T)_xi with g[kh][kw] = w[kh'][kw'][..]; flip = 0: (kr,no) = (ci,co); flip = 1 (backward-data):
// (kh',kw') = (2-kh,2-kw), (kr,no) = (co,ci).  w is the Keras (3,3,cin,cout) kernel.
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int Cin, int Cout, int flip, float* __restrict__ U) {
    const int KR = flip ? Cout : Cin, NO = flip ? Cin : Cout;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)KR * NO) return;
    const int no = (int)(e % NO), kr = (int)(e / NO);
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int kh = flip ? 2 - a : a, kw = flip ? 2 - b : b;
            const int ci = flip ? no : kr, co = flip ? kr : no;
            g[a][b] = w[((long)(kh * 3 + kw) * Cin + ci) * Cout + co];
        }
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const long xs = (long)KR * NO;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        U[(long)(i * 4 + 0) * xs + e] = t[i][0];
        U[(long)(i * 4 + 1) * xs + e] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        U[(long)(i * 4 + 2) * xs + e] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        U[(long)(i * 4 + 3) * xs + e] = t[i][2];
    }
}

// y[n, 2ty+i, 2tx+j, c] = act(bias + (A^T M A)_ij)
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mx, int N, int OH, int OW, int C, int TH, int TW,
                                                          const float* __restrict__ bias, int act, float alpha,
                                                          float* __restrict__ y, int y_cs, int accumulate) {
    const int C4 = C / 4;
    const long tiles = (long)N * TH * TW;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tiles * C4) return;
    const int c = (int)(e % C4) * 4;
    const long tile = e / C4;
    const int tx = (int)(tile % TW);
    const long r = tile / TW;
    const int ty = (int)(r % TH);
    const int n = (int)(r / TH);
    const long xs = tiles * C;
    const float* m = Mx + tile * C + c;
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 m0 = *(const f32x4*)(m + (long)(0 * 4 + j) * xs), m1 = *(const f32x4*)(m + (long)(1 * 4 + j) * xs);
        const f32x4 m2 = *(const f32x4*)(m + (long)(2 * 4 + j) * xs), m3 = *(const f32x4*)(m + (long)(3 * 4 + j) * xs);
        s[0][j] = m0 + m1 + m2;
        s[1][j] = m1 - m2 - m3;
    }
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4*)(bias + c);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f32x4 o[2];
        o[0] = s[i][0] + s[i][1] + s[i][2] + bv;
        o[1] = s[i][1] - s[i][2] - s[i][3] + bv;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = 2 * ty + i, ox = 2 * tx + j;
            if (oy < OH && ox < OW) {
                f32x4 v = o[j];
                if (act != SS_ACT_NONE) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = ss_apply_act(v[k], act, alpha);
                }
                f32x4* dst = (f32x4*)(y + ((long)(n * OH + oy) * OW + ox) * y_cs + c);
                if (accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

// dw[kh][kw][ci][co] (+)= (G^T S G) with S_xi[ci][co] = sum_splits part[xi][split][ci][co]
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ part, int splits, int Cin, int Cout, float* __restrict__ dw,
                                                      int accumulate) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long cc = (long)Cin * Cout;
    if (e >= cc) return;
    float S[4][4];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
        float acc = 0.f;
        for (int sp = 0; sp < splits; ++sp) acc += part[((long)xi * splits + sp) * cc + e];
        S[xi >> 2][xi & 3] = acc;
    }
    // G^T (3x4) = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = S[0][j] + 0.5f * (S[1][j] + S[2][j]);
        t[1][j] = 0.5f * (S[1][j] - S[2][j]);
        t[2][j] = 0.5f * (S[1][j] + S[2][j]) + S[3][j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float o[3];
        o[0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
        o[1] = 0.5f * (t[a][1] - t[a][2]);
        o[2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float* d = dw + (long)(a * 3 + b) * cc + e;
            *d = accumulate ? (*d + o[b]) : o[b];
        }
    }
}

inline unsigned g256(long total) { return (unsigned)((total + 255) / 256); }

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------
bool ss_wino_ok(const WinoProb& q) {
    if (getenv("SS_NO_WINOGRAD")) return false;
    const int kr = q.cin, no = q.cout;
    return kr % 32 == 0 && no % 4 == 0 && kr >= 64 && no >= 64 && q.in_cs % 4 == 0 && q.out_cs % 4 == 0 &&
           (long)q.n * ((q.oh + 1) / 2) * ((q.ow + 1) / 2) >= 1024;
}

size_t ss_wino_fwd_ws(const WinoProb& q) {
    const long tiles = (long)q.n * ((q.oh + 1) / 2) * ((q.ow + 1) / 2);
    return ss_align_up((size_t)16 * q.cin * q.cout * 4, 256) + ss_align_up((size_t)16 * tiles * q.cin * 4, 256) +
           ss_align_up((size_t)16 * tiles * q.cout * 4, 256);
}

// y = act(bias + conv3x3_stride1(x)) with out[o] = sum_a in[map(o + a - pt)] * g[a];  flip = 1: g = rotated + transposed w
int ss_wino_conv_fwd(const WinoProb& q, const float* x, const float* w, int w_cin, int w_cout, int flip, const float* bias, float* y,
                     int act, float alpha, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_fwd_ws(q)) return SS_ERR_WORKSPACE;
    const int TH = (q.oh + 1) / 2, TW = (q.ow + 1) / 2;
    const long tiles = (long)q.n * TH * TW;
    float* U = (float*)ws;
    float* V = (float*)((char*)ws + ss_align_up((size_t)16 * q.cin * q.cout * 4, 256));
    float* Mx = (float*)((char*)V + ss_align_up((size_t)16 * tiles * q.cin * 4, 256));
    hipLaunchKernelGGL(wino_weight_kernel, dim3(g256((long)q.cin * q.cout)), dim3(256), 0, s, w, w_cin, w_cout, flip, U);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(wino_input_kernel, dim3(g256(tiles * (q.cin / 4))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                       q.pt, q.pl, q.reflect, V);
    SS_LAUNCH_CHECK();
    GConvParams g{};
    g.in = V; g.w = U; g.bias = nullptr; g.out = Mx;
    g.N = 1; g.IH = 1; g.IW = (int)tiles; g.Cin = q.cin; g.in_cs = q.cin;
    g.OHc = 1; g.OWc = (int)tiles; g.in_s = 1; g.in_oy = 0; g.in_ox = 0;
    g.OH = 1; g.OW = (int)tiles; g.Cout = q.cout; g.out_cs = q.cout; g.out_s = 1; g.out_oy = 0; g.out_ox = 0;
    g.ldb = q.cout; g.reflect = 0; g.act = SS_ACT_NONE; g.alpha = 0.f; g.accumulate = 0;
    g.nbatch = 16; g.in_bs = tiles * q.cin; g.w_bs = (long)q.cin * q.cout; g.out_bs = tiles * q.cout;
    g.ntaps = 1; g.taps[0].dy = 0; g.taps[0].dx = 0; g.taps[0].woff = 0;
    int rc = ss_launch_gconv_mfma(g, s);
    if (rc != SS_OK) return rc;
    hipLaunchKernelGGL(wino_output_kernel, dim3(g256(tiles * (q.cout / 4))), dim3(256), 0, s, Mx, q.n, q.oh, q.ow, q.cout, TH, TW,
                       bias, act, alpha, y, q.out_cs, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}

size_t ss_wino_wgrad_ws(const WinoProb& q) {
    const long tiles = (long)q.n * ((q.oh + 1) / 2) * ((q.ow + 1) / 2);
    int pps;
    const int splits = ss_wgrad_mfma_splits(tiles, q.cin, q.cout, &pps, 16);
    return ss_align_up((size_t)16 * tiles * q.cin * 4, 256) + ss_align_up((size_t)16 * tiles * q.cout * 4, 256) +
           ss_align_up((size_t)16 * (splits + 1) * q.cin * q.cout * 4, 256);
}

// dw (3,3,cin,cout) (+)= sum_pixels xpad[o + a] * dy[o]
int ss_wino_conv_wgrad(const WinoProb& q, const float* x, const float* dy, float* dw, int accumulate, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    if (!ws || ws_bytes < ss_wino_wgrad_ws(q)) return SS_ERR_WORKSPACE;
    const int TH = (q.oh + 1) / 2, TW = (q.ow + 1) / 2;
    const long tiles = (long)q.n * TH * TW;
    float* V = (float*)ws;
    float* E = (float*)((char*)ws + ss_align_up((size_t)16 * tiles * q.cin * 4, 256));
    float* part = (float*)((char*)E + ss_align_up((size_t)16 * tiles * q.cout * 4, 256));
    hipLaunchKernelGGL(wino_input_kernel, dim3(g256(tiles * (q.cin / 4))), dim3(256), 0, s, x, q.in_cs, q.n, q.h, q.w, q.cin, TH, TW,
                       q.pt, q.pl, q.reflect, V);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(wino_dy_kernel, dim3(g256(tiles * (q.cout / 4))), dim3(256), 0, s, dy, q.out_cs, q.n, q.oh, q.ow, q.cout, TH, TW, E);
    SS_LAUNCH_CHECK();
    WGradParams p{};
    p.a = V; p.b = E; p.part = part;
    p.N = 1; p.AH = 1; p.AW = (int)tiles; p.Ca = q.cin; p.a_cs = q.cin;
    p.GH = 1; p.GW = (int)tiles; p.Cb = q.cout; p.b_cs = q.cout;
    p.a_s = 1; p.a_oy = 0; p.a_ox = 0; p.reflect = 0;
    p.ntaps = 1; p.taps[0].dy = 0; p.taps[0].dx = 0; p.taps[0].woff = 0;
    int pps;
    p.splits = ss_wgrad_mfma_splits(tiles, q.cin, q.cout, &pps, 16);
    p.pix_per_split = pps;
    p.nbatch = 16; p.a_bs = tiles * q.cin; p.b_bs = tiles * q.cout;
    int rc = ss_launch_wgrad_mfma_partials(p, s);
    if (rc != SS_OK) return rc;
    hipLaunchKernelGGL(wino_dw_kernel, dim3(g256((long)q.cin * q.cout)), dim3(256), 0, s, part, p.splits, q.cin, q.cout, dw, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
