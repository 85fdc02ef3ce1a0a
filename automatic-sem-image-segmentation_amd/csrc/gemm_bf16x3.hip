// Batched "split-bf16" GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate):
//   C[b][m][n] = sum_k (Ah + Al)[b][m][k] * (Bh + Bl)[b][n][k]      computed as  Ah*Bh + Ah*Bl + Al*Bh   (fp32 accumulate)
// Every fp32 operand v is carried as two bf16 planes  hi = bf16(v), lo = bf16(v - hi)  (16 mantissa bits together); the
// dropped lo*lo term and the representation error are ~2^-16 relative per product.  OPT-IN (SS_ALGO_BF16X3): the default
// path stays exact fp32.  Used for the Winograd GEMMs, whose transforms emit the planes directly (no conversion pass).
//
// Both operands are K-contiguous ([m][k] and [n][k] rows), the easy MFMA case: 16-byte global loads -> padded LDS rows
// (80 B stride: conflict-free ds_read_b128 fragments) -> one b128 fragment read per 32x32x16 MFMA operand.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BM = 128, GB_BN = 128, GB_BK = 32;           // bf16 elements along K per step
constexpr int GB_LD = GB_BK + 8;                               // padded LDS row (bf16 elements): 80 bytes
constexpr int GB_TILE = GB_BM * GB_LD;                         // elements per operand plane per stage

__global__ __launch_bounds__(256) void bgemm_bf16x3_kernel(BGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];    // [stage][Ah|Al|Bh|Bl][128][40]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int gridM = (p.M + GB_BM - 1) / GB_BM, gridN = (p.N + GB_BN - 1) / GB_BN;
    int tile;
    {   // XCD-aware order (speed only): contiguous chunk of the tile space per XCD, N fastest
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int batch = tile / (gridM * gridN);
    tile -= batch * gridM * gridN;
    const int m0 = (tile / gridN) * GB_BM, n0 = (tile % gridN) * GB_BN;

    const unsigned short* Ah = p.ah + (long)batch * p.a_bs;
    const unsigned short* Al = p.al + (long)batch * p.a_bs;
    const unsigned short* Bh = p.bh + (long)batch * p.b_bs;
    const unsigned short* Bl = p.bl + (long)batch * p.b_bs;

    // loader: a plane tile = 128 rows x 64 B = 512 16-B pieces -> 2 per thread: rows (tid>>2) and (tid>>2)+64, piece tid&3
    const int lrow = tid >> 2, lpc = tid & 3;
    long a_off[2], b_off[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ma = m0 + lrow + 64 * j, nb = n0 + lrow + 64 * j;
        a_ok[j] = ma < p.M;
        b_ok[j] = nb < p.N;
        a_off[j] = (long)(a_ok[j] ? ma : 0) * p.lda + lpc * 8;
        b_off[j] = (long)(b_ok[j] ? nb : 0) * p.ldb + lpc * 8;
    }
    u32x4 r[8];
    const u32x4 z = {0u, 0u, 0u, 0u};
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x4 v0 = *(const u32x4*)(Ah + a_off[j] + k0), v1 = *(const u32x4*)(Al + a_off[j] + k0);
            const u32x4 v2 = *(const u32x4*)(Bh + b_off[j] + k0), v3 = *(const u32x4*)(Bl + b_off[j] + k0);
            r[j * 4 + 0] = a_ok[j] ? v0 : z;
            r[j * 4 + 1] = a_ok[j] ? v1 : z;
            r[j * 4 + 2] = b_ok[j] ? v2 : z;
            r[j * 4 + 3] = b_ok[j] ? v3 : z;
        }
    };
    auto store = [&](int stage) {
        unsigned short* base = lds + stage * 4 * GB_TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 4; ++pl)
                *(u32x4*)(base + pl * GB_TILE + (lrow + 64 * j) * GB_LD + lpc * 8) = r[j * 4 + pl];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

    const int nchunks = p.K / GB_BK;
    load(0);
    store(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        if (c + 1 < nchunks) load((c + 1) * GB_BK);
        const unsigned short* sb = lds + st * 4 * GB_TILE;
        const unsigned short* sa_h = sb + (wm * 64 + l31) * GB_LD + 8 * lh;
        const unsigned short* sa_l = sa_h + GB_TILE;
        const unsigned short* sb_h = sb + 2 * GB_TILE + (wn * 64 + l31) * GB_LD + 8 * lh;
        const unsigned short* sb_l = sb_h + GB_TILE;
#pragma unroll
        for (int ks = 0; ks < GB_BK / 16; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *(const bf16x8*)(sa_h + i * 32 * GB_LD + ks * 16);
                al[i] = *(const bf16x8*)(sa_l + i * 32 * GB_LD + ks * 16);
                bh[i] = *(const bf16x8*)(sb_h + i * 32 * GB_LD + ks * 16);
                bl[i] = *(const bf16x8*)(sb_l + i * 32 * GB_LD + ks * 16);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        if (c + 1 < nchunks) store(st ^ 1);
        __syncthreads();
    }

    float* C = p.c + (long)batch * p.c_bs;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                if (m < p.M) C[(long)m * p.ldc + n] = acc[mi][ni][q];
            }
    }
}

}  // namespace

int ss_launch_bgemm_bf16x3(const BGemmParams& p, hipStream_t s) {
    if (p.K % GB_BK != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0) return SS_ERR_UNSUPPORTED;
    const int gridM = (p.M + GB_BM - 1) / GB_BM, gridN = (p.N + GB_BN - 1) / GB_BN;
    const size_t smem = (size_t)2 * 4 * GB_TILE * sizeof(unsigned short);
    // one-time kernel attribute (idempotent; C++11 thread-safe static initialisation, no mutable flag)
    static const bool attr_set = [] {
        (void)hipFuncSetAttribute((const void*)bgemm_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL(bgemm_bf16x3_kernel, dim3((unsigned)(gridM * gridN * (p.nbatch > 1 ? p.nbatch : 1))), dim3(256), smem, s, p);
    SS_LAUNCH_CHECK();
    return SS_OK;
}
