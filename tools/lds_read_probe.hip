// Micro-benchmark: LDS cycles of the x3h GEMM's fragment reads (16 ds_read_b128 per wave and K step from a 48 KiB stage of 64-byte rows,
// 16-byte slot XOR-swizzled with bits 2..3 of the row), 8 waves per CU, against a linear (unswizzled) image and a plain
// lane-contiguous pattern.  Reads are inline asm (the compiler must not hoist or merge them).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_read_probe.hip -o tools/lds_read_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 lds_read128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// MODE 0: GEMM pattern (swizzled), 1: same rows, no swizzle, 2: lane-contiguous 1 KiB per instruction, 3: GEMM pattern with 128-byte rows (BK = 64, st_16x32-like swizzle)
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(int steps, float* sink, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int sw = (l31 >> 2) & 3;
    unsigned fa, fb, so0, so1;
    constexpr int ROWB = MODE == 3 ? 128 : 64;
    constexpr int A_PLANE = 256 * ROWB, B_PLANE = 128 * ROWB;
    if (MODE == 0 || MODE == 3) { so0 = ((lh ^ sw) << 4); so1 = so0 ^ 32; }
    else { so0 = lh << 4; so1 = so0 + 32; }
    fa = (wm * 64 + l31) * ROWB;
    fb = 2 * A_PLANE + (wn * 64 + l31) * ROWB;
    if (MODE == 2) { fa = wave * 16384 + lane * 16; fb = fa + 8192; so0 = 0; so1 = 1024; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned so = h ? so1 : so0;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (MODE == 2) {
                        acc += lds_read128(fa + so + (pl * 2 + i) * 2048);
                        acc += lds_read128(fb + so + (pl * 2 + i) * 2048);
                    } else {
                        acc += lds_read128(fa + pl * A_PLANE + i * 32 * ROWB + so);
                        acc += lds_read128(fb + pl * B_PLANE + i * 32 * ROWB + so);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* sink, unsigned long long* clk, int threads) {
    const int steps = 20000;
    CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads), 144 * 1024, 0, 100, sink, clk);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads), 144 * 1024, 0, steps, sink, clk);
    CK(hipDeviceSynchronize());
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    printf("%-44s %d waves: %7.1f cycles per step (16 ds_read_b128 per wave) = %5.2f cycles per wave-instruction per CU\n", name, threads / 64,
           (double)c / steps, (double)c / steps / (16.0 * threads / 64));
}

int main() {
    float* sink; unsigned long long* clk;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clk, 64));
    for (int threads : {512, 256}) {
        run<0>("GEMM fragment pattern (64-B rows, XOR swizzle)", sink, clk, threads);
        run<1>("same rows, no swizzle", sink, clk, threads);
        run<2>("lane-contiguous", sink, clk, threads);
        run<3>("128-B rows, XOR swizzle", sink, clk, threads);
    }
    return 0;
}
