// Micro-benchmark: do LDS-DMA writes (global_load_lds_dwordx4 landing in LDS) and ds_read_b128 fragment reads share the LDS port?
// Waves 0-3 stream L2-resident data into LDS by LDS-DMA; waves 4-7 read LDS with ds_read_b128 (the GEMM's fragment pattern, conflict
// free).  Times: DMA alone, reads alone, both.  both ~ max -> independent;  both ~ sum -> one port.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_contention_probe.hip -o tools/lds_contention_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool DMA, bool RD, bool ST>
__global__ __launch_bounds__(512, 1) void probe(char* __restrict__ src, long region, int steps, float* sink, char* __restrict__ dstg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* base = src + (long)(blockIdx.x & 7) * region;
    const int pieces = (int)(region / 1024);
    int pc = (int)(((long)(blockIdx.x >> 3) * 977) % pieces);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) {
        if (DMA) {
            for (int s = 0; s < steps; ++s) {
                unsigned char* dst = lds + ((s % 3) * 4 * 12 + wave * 12) * 1024;          // 12 pieces per wave and step = 48 KiB per step per CU
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    int q = pc + wave * 12 + j;
                    if (q >= pieces) q -= pieces;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (long)q * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
                }
                pc += 48;
                if (pc >= pieces) pc -= pieces;
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
        }
    } else {
        if (RD) {
            // 32 ds_read_b128 per step per wave (4 waves: 128 KiB per step per CU, the GEMM's fragment volume), lane = row of 64 B, XOR-swizzled slot
            const int row = lane & 31, sl = (lane >> 5) ^ ((row >> 2) & 3);
            const unsigned char* fa = lds + (wave - 4) * 32 * 64 * 8 + row * 64 + sl * 16;
            for (int s = 0; s < steps; ++s) {
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const f32x4 v = *(const f32x4*)(fa + (j & 7) * 2048 + ((j >> 3) & 1) * 32 + (j >> 4) * 65536);
                    acc += v;
                }
            }
        }
        if (ST) {
            char* gb = dstg + ((long)blockIdx.x * 4 + (wave - 4)) * 65536;
            for (int s = 0; s < steps; ++s) {
#pragma unroll
                for (int j = 0; j < 2; ++j)          // 2 x 1 KiB per wave and step = 8 KiB per step per CU (C stores of a 256 x 128 tile over 16 K steps)
                    *(f32x4*)(gb + ((s * 2 + j) & 63) * 1024 + lane * 16) = acc;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <bool DMA, bool RD, bool ST>
void run(const char* name, char* src, long region, float* sink, char* dstg) {
    const int steps = 4000, nwg = 256;
    CK(hipFuncSetAttribute((const void*)probe<DMA, RD, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<DMA, RD, ST>), dim3(nwg), dim3(512), 144 * 1024, 0, src, region, 200, sink, dstg);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<DMA, RD, ST>), dim3(nwg), dim3(512), 144 * 1024, 0, src, region, steps, sink, dstg);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.1f us = %6.3f us per step (48 KiB DMA / 128 KiB reads / 8 KiB stores per step and CU)\n", name, ms * 1e3, ms * 1e3 / steps);
}

int main() {
    char *src, *dstg; float* sink;
    CK(hipMalloc(&src, 1L << 30)); CK(hipMemset(src, 1, 1L << 30));
    CK(hipMalloc(&dstg, 256L * 4 * 65536)); CK(hipMalloc(&sink, 64));
    for (long region : {2L << 20, 128L << 20}) {
        printf("---- DMA source region per XCD: %ld MiB\n", region >> 20);
        run<true, false, false>("LDS-DMA alone (4 waves)", src, region, sink, dstg);
        run<false, true, false>("ds_read_b128 alone (4 waves)", src, region, sink, dstg);
        run<true, true, false>("LDS-DMA + ds_read_b128", src, region, sink, dstg);
        run<false, false, true>("global stores alone (4 waves)", src, region, sink, dstg);
        run<true, false, true>("LDS-DMA + global stores", src, region, sink, dstg);
        run<true, true, true>("LDS-DMA + ds_read_b128 + stores", src, region, sink, dstg);
    }
    return 0;
}
