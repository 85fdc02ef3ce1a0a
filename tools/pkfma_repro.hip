// Stand-alone probe for the packed-fp32 hazard of DESIGN.md ("Two concurrent kernel chains ..."): does v_pk_fma_f32 return wrong
// results while ANOTHER stream's bf16-MFMA kernel is resident on the same CU?
//   kernel P: every lane runs chains of v_pk_fma_f32 (plain, op_sel and op_sel_hi forms, as hipcc emits them for the one-channel
//             7x7 kernel) and, on the same operands, the two scalar v_fma_f32 they stand for; any bit difference is counted.
//   kernel M: a register-only loop of v_mfma_f32_32x32x16_bf16 (arg 1) or v_mfma_f32_32x32x2_f32 (arg 2), or nothing (arg 0).
// Both are launched with 2 workgroups of 256 threads per CU on their own streams so that they share SIMDs; P verifies itself.
//   hipcc --offload-arch=gfx950 -O2 -o pkfma_repro pkfma_repro.hip && ./pkfma_repro 1 20
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void pk_kernel(unsigned long long* bad, int iters, float seed) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 a = {1.0f + (t & 15) * 0.125f, 0.5f + (t & 7) * 0.25f}, b = {seed, seed * 0.5f + 0.375f};
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
    float r0x = 0.f, r0y = 0.f, r1x = 0.f, r1y = 0.f, r2x = 0.f, r2y = 0.f;
    unsigned long long nbad = 0;
    for (int i = 0; i < iters; ++i) {
        // packed forms
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));                                   // (ax*bx, ay*by)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc1) : "v"(a), "v"(b));                 // (ax*bx, ax*by)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc2) : "v"(a), "v"(b)); // (ay*bx, ay*by)
        // the scalar instructions they stand for
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0x) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0y) : "v"(a.y), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1x) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1y) : "v"(a.x), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r2x) : "v"(a.y), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r2y) : "v"(a.y), "v"(b.y));
        if ((i & 63) == 63) {
            nbad += (__float_as_uint(acc0.x) != __float_as_uint(r0x)) + (__float_as_uint(acc0.y) != __float_as_uint(r0y)) +
                    (__float_as_uint(acc1.x) != __float_as_uint(r1x)) + (__float_as_uint(acc1.y) != __float_as_uint(r1y)) +
                    (__float_as_uint(acc2.x) != __float_as_uint(r2x)) + (__float_as_uint(acc2.y) != __float_as_uint(r2y));
            acc0 = acc1 = acc2 = f32x2{0.f, 0.f};
            r0x = r0y = r1x = r1y = r2x = r2y = 0.f;
            a.x += 0.0078125f; b.y -= 0.00390625f;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, int kind) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + 0.125f * ((threadIdx.x + j) & 7)); b[j] = (__bf16)(0.5f + 0.0625f * ((threadIdx.x * 3 + j) & 15)); }
    const float fa = 1.0f + (threadIdx.x & 7) * 0.125f, fb = 0.5f + (threadIdx.x & 3) * 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kind == 1) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
            else acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[k], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 10;
    hipStream_t s1, s2;
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2);
    unsigned long long* bad; float* out;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&out, 512 * 256 * 4);
    unsigned long long total = 0;
    for (int r = 0; r < reps; ++r) {
        (void)hipMemset(bad, 0, 8);
        if (kind) hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, s2, out, 60000, kind);
        hipLaunchKernelGGL(pk_kernel, dim3(512), dim3(256), 0, s1, bad, 1 << 18, 0.75f + 0.001f * r);
        (void)hipDeviceSynchronize();
        unsigned long long h = 0;
        (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        total += h;
        printf("rep %d: %llu mismatching packed results\n", r, h);
    }
    printf("kind %d: %llu mismatches in %d repetitions (131072 threads x 3 forms x 2 halves x 4096 checks each)\n", kind, total, reps);
    return total ? 1 : 0;
}
