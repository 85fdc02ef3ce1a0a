// Micro-benchmark: the matrix pipe's rate and the shader clock under a pure v_mfma_f32_32x32x16_f16 stream (registers only), by
// operand data (zeros / random normal-ish fp16) and waves per SIMD.  Tells what "peak" a real-data kernel can be held against on
// this box: the chip clocks to its power budget (MI355X_MICROARCH.md, DVFS).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock_probe.hip -o tools/mfma_clock_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512, 1) void mfma_loop(const f16x8* __restrict__ src, int iters, float* out, unsigned long long* clk) {
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 8 + i) & 4095]; b[i] = src[(threadIdx.x * 8 + 4 + i) & 4095]; }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
    f16x8* src; float* out; unsigned long long* clk;
    CK(hipMalloc(&src, 4096 * sizeof(f16x8)));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&clk, 64));
    _Float16* h = (_Float16*)malloc(4096 * 16);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (int i = 0; i < 4096 * 8; ++i) {
            float v = 0.f;
            if (mode == 1) { v = 0.f; for (int k = 0; k < 6; ++k) v += (float)rand() / RAND_MAX - 0.5f; v *= 2.f; }
            h[i] = (_Float16)v;
        }
        CK(hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice));
        for (int threads : {256, 512}) {
            const int iters = 20000;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(threads), 0, 0, src, 2000, out, clk);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(threads), 0, 0, src, iters, out, clk);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
            const double flops = 256.0 * (threads / 64) * iters * 24 * 32768.0;
            printf("%s operands, %d waves/SIMD: %8.1f us  %7.1f TF/s (%.3f of 2516.6)   shader cycles %llu, 100 MHz ticks %llu -> %.0f MHz; cycles per MFMA per SIMD %.1f\n",
                   mode ? "random" : "zero  ", threads / 256, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 2516.6, c[0], c[1], c[0] / (c[1] / 100.0),
                   (double)c[0] / (iters * 24.0 * (threads / 256)));
        }
    }
    return 0;
}
