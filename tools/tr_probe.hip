#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned short* in, unsigned short* out, int rowstride_bytes) {
    __shared__ __attribute__((aligned(1024))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    // lane supplies: row = (l%16)/4 + 8*(l/32)?  -- here simplest: group g = l/16, lane i = l%16: row = g*4 + i/4, col = (i%4)*4
    const int g = l >> 4, i = l & 15;
    const unsigned addr = (g * 4 + i / 4) * rowstride_bytes + (i % 4) * 8;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[4096], *d_in, *d_out, o[256];
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(o));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, 64);
    hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", o[l * 4 + j]); printf("\n"); }
    return 0;
}
