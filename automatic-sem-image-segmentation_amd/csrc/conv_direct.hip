// Direct (VALU) gather-convolution kernels: the always-correct path for shapes that are not
// matrix-core shaped (Cout == 1 heads, tiny test shapes) and the in-GPU cross-check for the MFMA path.
// One thread = one output pixel x COB output channels; weights are wave-uniform (scalar loads).
#include "common.h"

template <int COB>
__global__ __launch_bounds__(256) void gconv_direct_kernel(GConvParams p) {
    const long total = (long)p.N * p.OHc * p.OWc;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= total) return;
    const int co0 = blockIdx.y * COB;
    const int xc = (int)(pix % p.OWc);
    const long r = pix / p.OWc;
    const int yc = (int)(r % p.OHc);
    const int n = (int)(r / p.OHc);
    const int oy = yc * p.out_s + p.out_oy, ox = xc * p.out_s + p.out_ox;
    if (oy < 0 || oy >= p.OH || ox < 0 || ox >= p.OW) return;

    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;

    const int by = yc * p.in_s + p.in_oy, bx = xc * p.in_s + p.in_ox;
    for (int t = 0; t < p.ntaps; ++t) {
        const int iy = ss_map_index(by + p.taps[t].dy, p.IH, p.reflect);
        const int ix = ss_map_index(bx + p.taps[t].dx, p.IW, p.reflect);
        if (iy < 0 || ix < 0) continue;
        const float* ip = p.in + ((long)(n * p.IH + iy) * p.IW + ix) * p.in_cs;
        const float* wp = p.w + p.taps[t].woff + co0;
        for (int ci = 0; ci < p.Cin; ++ci) {
            const float v = ip[ci];
#pragma unroll
            for (int j = 0; j < COB; ++j) {
                const float wv = (co0 + j < p.Cout) ? wp[(long)ci * p.ldb + j] : 0.f;
                acc[j] = fmaf(v, wv, acc[j]);
            }
        }
    }
    float* op = p.out + ((long)(n * p.OH + oy) * p.OW + ox) * p.out_cs;
#pragma unroll
    for (int j = 0; j < COB; ++j) {
        const int co = co0 + j;
        if (co < p.Cout) {
            float v = acc[j] + (p.bias ? p.bias[co] : 0.f);
            v = ss_apply_act(v, p.act, p.alpha);
            if (p.accumulate) v += op[co];
            op[co] = v;
        }
    }
}

int ss_launch_gconv_direct(const GConvParams& p, hipStream_t s) {
    const long total = (long)p.N * p.OHc * p.OWc;
    if (total == 0) return SS_OK;
    dim3 block(256);
    if (p.Cout >= 8) {
        dim3 grid((unsigned)((total + 255) / 256), (p.Cout + 7) / 8);
        hipLaunchKernelGGL(gconv_direct_kernel<8>, grid, block, 0, s, p);
    } else if (p.Cout >= 2) {
        dim3 grid((unsigned)((total + 255) / 256), (p.Cout + 3) / 4);
        hipLaunchKernelGGL(gconv_direct_kernel<4>, grid, block, 0, s, p);
    } else {
        dim3 grid((unsigned)((total + 255) / 256), 1);
        hipLaunchKernelGGL(gconv_direct_kernel<1>, grid, block, 0, s, p);
    }
    SS_LAUNCH_CHECK();
    return SS_OK;
}

// dw[woff_t + ca*ldw + cb] (+)= sum_pixels a_gathered * b ; one thread per dw element (slow fallback)
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WGradParams p, float* dw, int ldw, int accumulate) {
    const long total = (long)p.ntaps * p.Ca * p.Cb;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int cb = (int)(e % p.Cb);
    const long r = e / p.Cb;
    const int ca = (int)(r % p.Ca);
    const int t = (int)(r / p.Ca);
    const int dy = p.taps[t].dy, dx = p.taps[t].dx;
    float acc = 0.f;
    for (int n = 0; n < p.N; ++n)
        for (int yc = 0; yc < p.GH; ++yc) {
            const int iy = ss_map_index(yc * p.a_s + p.a_oy + dy, p.AH, p.reflect);
            if (iy < 0) continue;
            for (int xc = 0; xc < p.GW; ++xc) {
                const int ix = ss_map_index(xc * p.a_s + p.a_ox + dx, p.AW, p.reflect);
                if (ix < 0) continue;
                const float av = p.a[((long)(n * p.AH + iy) * p.AW + ix) * p.a_cs + ca];
                const float bv = p.b[((long)(n * p.GH + yc) * p.GW + xc) * p.b_cs + cb];
                acc = fmaf(av, bv, acc);
            }
        }
    float* o = dw + p.taps[t].woff + (long)ca * ldw + cb;
    *o = accumulate ? (*o + acc) : acc;
}

int ss_launch_wgrad_direct(const WGradParams& p, float* dw, int ldw, int accumulate, hipStream_t s) {
    const long total = (long)p.ntaps * p.Ca * p.Cb;
    if (total == 0) return SS_OK;
    hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, dw, ldw, accumulate);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
