// Micro-benchmark: what does the path L2 -> LDS (LDS-DMA, global_load_lds_dwordx4) deliver per CU when the source is L2-resident,
// by access shape and by how many pieces a wave keeps in flight?  The x3h Winograd GEMMs move 48 KiB per K step and CU and sit
// at ~20 GB/s per CU whatever their schedule (one-phase or ping-pong): is that the path, the shape (16 rows x 64 B per
// instruction at a 1 KiB pitch) or the depth?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: piece = 16 rows x 64 B, row pitch `pitch` bytes (the GEMM's operand shape); 1: piece = 1 KiB contiguous;
// 2: piece = 8 rows x 128 B (full lines) at row pitch; 3: plain global_load_dwordx4 to registers, contiguous 1 KiB per wave
// Every wave issues NP pieces per step into its own slice of a 3-stage LDS ring and lets NP * (DEPTH) pieces stay in flight.
template <int MODE, int NP, int DEPTH>
__global__ __launch_bounds__(512, 1) void probe(char* __restrict__ src, long region, int pitch, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    char* base = src + (long)(blockIdx.x & 7) * region;          // one region per XCD (block b runs on XCD b % 8)
    // this workgroup walks `region` in units of (nwv * NP) pieces per step, starting at a per-workgroup offset
    const int pieces = (int)(region / 1024);
    int pc = (int)(((long)(blockIdx.x >> 3) * 977) % pieces);
    const int lp = 31 - __builtin_clz(pitch);          // pitch is a power of two: shifts, no divisions in the timed loop
    unsigned lane_off;
    if (MODE == 0) lane_off = (lane >> 2) * pitch + (lane & 3) * 16;
    else if (MODE == 2) lane_off = (lane >> 3) * pitch + (lane & 7) * 16;
    else if (MODE == 4) lane_off = (lane >> 4) * pitch + (lane & 15) * 16;          // 4 rows x 256 B
    else if (MODE == 5) lane_off = (lane >> 5) * pitch + (lane & 31) * 16;          // 2 rows x 512 B
    else if (MODE == 6) lane_off = (lane >> 2) * pitch + (lane & 3) * 16;           // store: 16 rows x 64 B
    else if (MODE == 7) lane_off = (lane >> 4) * pitch + (lane & 15) * 16;          // store: 4 rows x 256 B
    else lane_off = lane * 16;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        unsigned char* dst = lds + ((s % 3) * nwv * NP + wave * NP) * 1024;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int q = pc + wave * NP + j;
            if (q >= pieces) q -= pieces;
            char* g;
            if (MODE == 0) {          // piece q: rows 16 * (q / (pitch/64)) ..., k slot q % (pitch/64)
                g = base + ((long)(q >> (lp - 6)) << (lp + 4)) + ((q & ((1 << (lp - 6)) - 1)) << 6);
            } else if (MODE == 2) {
                g = base + ((long)(q >> (lp - 7)) << (lp + 3)) + ((q & ((1 << (lp - 7)) - 1)) << 7);
            } else if (MODE == 4 || MODE == 7) {
                g = base + ((long)(q >> (lp - 8)) << (lp + 2)) + ((q & ((1 << (lp - 8)) - 1)) << 8);
            } else if (MODE == 5) {
                g = base + ((long)(q >> (lp - 9)) << (lp + 1)) + ((q & ((1 << (lp - 9)) - 1)) << 9);
            } else if (MODE == 6) {
                g = base + ((long)(q >> (lp - 6)) << (lp + 4)) + ((q & ((1 << (lp - 6)) - 1)) << 6);
            } else g = base + (long)q * 1024;
            if (MODE == 3) {
                const float4 v = *(const float4*)(g + lane_off);
                acc += v.x + v.y + v.z + v.w;
            } else if (MODE >= 6) {
                *(float4*)(g + lane_off) = float4{acc, 1.f, 2.f, (float)s};
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane_off),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
            }
        }
        pc += nwv * NP;
        if (pc >= pieces) pc -= pieces;
        if (MODE != 3 && MODE < 6) {
            if (DEPTH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * DEPTH) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 3 && acc == 12345.678f) sink[0] = acc;
    if (MODE != 3 && MODE < 6 && lds[threadIdx.x] == 77 && steps < 0) sink[1] = 1.f;
}

template <int MODE, int NP, int DEPTH>
void run(const char* name, char* src, long region, int pitch, int threads, float* sink) {
    const int steps = 2000;
    const int nwg = 256;
    const size_t ldsb = 3 * (threads / 64) * NP * 1024;
    CK(hipFuncSetAttribute((const void*)probe<MODE, NP, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, NP, DEPTH>), dim3(nwg), dim3(threads), ldsb, 0, src, region, pitch, 200, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, NP, DEPTH>), dim3(nwg), dim3(threads), ldsb, 0, src, region, pitch, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * steps * (threads / 64) * NP * 1024;
    printf("%-34s region/XCD %6.2f MiB pitch %5d waves %2d NP %d depth %d : %7.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", name, region / 1048576.0, pitch,
           threads / 64, NP, DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
}

int main() {
    const long total = 1L << 30;
    char* src;
    float* sink;
    CK(hipMalloc(&src, total));
    CK(hipMemset(src, 1, total));
    CK(hipMalloc(&sink, 64));
    for (long region : {64L << 10, 2L << 20}) {
        printf("---- second set: region per XCD %ld KiB\n", region >> 10);
        run<0, 6, 2>("rows16x64B", src, region, 1024, 512, sink);
        run<2, 6, 2>("rows8x128B", src, region, 1024, 512, sink);
        run<4, 6, 2>("rows4x256B", src, region, 1024, 512, sink);
        run<5, 6, 2>("rows2x512B", src, region, 1024, 512, sink);
        run<1, 6, 2>("contiguous 1 KiB", src, region, 1024, 512, sink);
        run<0, 6, 2>("rows16x64B pitch 2048", src, region, 2048, 512, sink);
        run<0, 6, 2>("rows16x64B pitch 256", src, region, 256, 512, sink);
        run<8, 6, 0>("store contiguous 1 KiB", src, region, 2048, 512, sink);
        run<7, 6, 0>("store rows4x256B pitch 2048", src, region, 2048, 512, sink);
        run<6, 6, 0>("store rows16x64B pitch 1024", src, region, 1024, 512, sink);
    }
    for (long region : {128L << 20}) {
        printf("---- stores to a large region\n");
        run<8, 6, 0>("store contiguous 1 KiB", src, region, 2048, 512, sink);
        run<7, 6, 0>("store rows4x256B pitch 2048", src, region, 2048, 512, sink);
    }
    return 0;
}
