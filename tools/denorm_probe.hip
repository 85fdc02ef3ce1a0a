// Does gfx950 keep fp16 SUBNORMALS (a) in v_cvt_f16_f32, (b) as inputs of v_mfma_f32_32x32x16_f16 ?
// hipcc --offload-arch=gfx950 -O2 tools/denorm_probe.hip -o gpurun_out/denorm_probe && gpurun_out/denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    // (a) conversion of 2^-20 (fp16 subnormal) and back
    volatile float tiny = 9.5367431640625e-07f;   // 2^-20
    _Float16 hs = (_Float16)tiny;
    if (l == 0) out[0] = (float)hs;
    // (b) MFMA: A row 0 = [2^-20, 0...], B col 0 = [2^10, 0...]  -> C[0][0] = 2^-10 if the subnormal input is honoured, 0 if flushed
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned short sub = 0x0010;    // fp16 subnormal bit pattern = 16 * 2^-24 = 2^-20
    if (l == 0) { a[0] = __builtin_bit_cast(_Float16, sub); b[0] = (_Float16)1024.f; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[1] = c[0];
    // (c) bf16 subnormal-range check is moot (8-bit exponent); (d) fp16 normal control: 2^-14 * 2^10 = 2^-4
    h8 a2 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == 0) a2[0] = (_Float16)6.103515625e-05f;
    f16v c2 = {0};
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c2, 0, 0, 0);
    if (l == 0) out[2] = c2[0];
}
int main() {
    float* d;
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[3];
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("cvt(2^-20) -> fp16 -> f32 = %g (expect 9.53674e-07 if kept)\n", h[0]);
    printf("mfma f16 subnormal input 2^-20 * 2^10 = %g (expect 0.000976562 if kept, 0 if flushed)\n", h[1]);
    printf("mfma f16 normal control 2^-14 * 2^10 = %g (expect 0.0625)\n", h[2]);
    return 0;
}
