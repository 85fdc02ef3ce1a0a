// Micro-benchmark: what does a read-134-MB / write-134-MB elementwise pass (the trunk's norm apply: 16 x 4096 x 512 fp32) reach on this
// box, by launch geometry and cache policy?  norm_apply_kernel<float,4> runs at 3.8 TB/s (71 us), the copy ceiling of the guide is 6.3.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o tools/stream_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld(const f32x4* p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st(f32x4* p, f32x4 v, int nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0: flat grid-stride, U vectors in flight per thread;  MODE 1: the norm kernels' geometry: block = 128 channel lanes x 2 row lanes,
// rows_per_chunk rows of 2 KB per block, 4 rows in flight per thread
template <int MODE, int U, int NT>
__global__ __launch_bounds__(256) void pass(const f32x4* __restrict__ x, f32x4* __restrict__ y, long nvec, const float* __restrict__ mu, int rpc) {
    if (MODE == 0) {
        const long stride = (long)gridDim.x * 256;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) if (i + u * stride < nvec) v[u] = ld(x + i + u * stride, NT);
#pragma unroll
            for (int u = 0; u < U; ++u) if (i + u * stride < nvec) {
                const int c = (int)((i + u * stride) & 127);
                const f32x4 m = ((const f32x4*)mu)[c];
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float t = __builtin_fmaf(v[u][k] - m[k], 1.25f, 0.1f); o[k] = t > 0.f ? t : 0.f; }
                st(y + i + u * stride, o, NT);
            }
        }
    } else {
        const int ct = threadIdx.x & 127, pt = threadIdx.x >> 7;
        const f32x4 m = ((const f32x4*)mu)[ct];
        const long p0 = (long)blockIdx.x * rpc, p1 = p0 + rpc;
        for (long p = p0 + pt; p < p1; p += 2 * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld(x + (p + 2 * u) * 128 + ct, NT);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float t = __builtin_fmaf(v[u][k] - m[k], 1.25f, 0.1f); o[k] = t > 0.f ? t : 0.f; }
                st(y + (p + 2 * u) * 128 + ct, o, NT);
            }
        }
    }
}

template <int MODE, int U, int NT>
void run(const char* name, f32x4** xs, f32x4** ys, long nvec, const float* mu, int grid, int rpc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((pass<MODE, U, NT>), dim3(grid), dim3(256), 0, 0, xs[i & 7], ys[i & 7], nvec, mu, rpc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 40;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((pass<MODE, U, NT>), dim3(grid), dim3(256), 0, 0, xs[i & 7], ys[i & 7], nvec, mu, rpc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s grid %6d: %6.1f us  %5.2f TB/s (read + write)\n", name, grid, ms * 1e3 / iters, 2.0 * nvec * 16 / (ms / iters) / 1e9);
}

int main() {
    const long nvec = 16L * 4096 * 128;          // 134 MB
    f32x4 *xs[8], *ys[8]; float* mu;
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&xs[i], nvec * 16)); CK(hipMalloc(&ys[i], nvec * 16)); CK(hipMemset(xs[i], 0x3c, nvec * 16)); }
    CK(hipMalloc(&mu, 2048)); CK(hipMemset(mu, 0, 2048));
    run<1, 4, 0>("norm geometry: 32 rows per block, 4 in flight", xs, ys, nvec, mu, 2048, 32);
    run<1, 4, 1>("norm geometry, nontemporal", xs, ys, nvec, mu, 2048, 32);
    run<1, 8, 0>("norm geometry: 64 rows per block, 8 in flight", xs, ys, nvec, mu, 1024, 64);
    run<1, 8, 1>("norm geometry: 64 rows, 8 in flight, nontemporal", xs, ys, nvec, mu, 1024, 64);
    run<1, 8, 0>("norm geometry: 128 rows per block, 8 in flight", xs, ys, nvec, mu, 512, 128);
    run<1, 4, 0>("norm geometry: 256 rows per block, 4 in flight", xs, ys, nvec, mu, 256, 256);
    run<0, 4, 0>("flat grid-stride, 4 in flight", xs, ys, nvec, mu, 2048, 0);
    run<0, 4, 1>("flat grid-stride, 4 in flight, nontemporal", xs, ys, nvec, mu, 2048, 0);
    run<0, 8, 0>("flat grid-stride, 8 in flight", xs, ys, nvec, mu, 2048, 0);
    run<0, 8, 1>("flat grid-stride, 8 in flight, nontemporal", xs, ys, nvec, mu, 2048, 0);
    run<0, 8, 0>("flat grid-stride, 8 in flight", xs, ys, nvec, mu, 4096, 0);
    run<0, 2, 0>("flat grid-stride, 2 in flight", xs, ys, nvec, mu, 8192, 0);
    run<0, 1, 0>("flat, 1 vector per thread per iteration", xs, ys, nvec, mu, 16384, 0);
    run<0, 1, 0>("flat, one vector per thread (no loop)", xs, ys, nvec, mu, (int)(nvec / 256), 0);
    return 0;
}
