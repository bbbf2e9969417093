// A stand-alone aggressor for tests/diag/diag_hazard_r6.py (BFM_DIAG_AGGR=1): the MFMA tap loop of conv_wino4d without the
// convolution around it -- per step six global_load_dwordx4 into VGPRs (next step's B operands), ds_read_b128 of the A
// operands, 18 v_mfma_f32_32x32x16_f16 on six accumulators; ~250 VGPRs and 76.8 KB of LDS so that two workgroups share a CU
// and two waves a SIMD, as the real kernel.  Built to a code object by tests/diag/hazard_patch.py (variant "aggr").
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
constexpr int BALLAST = 6;

extern "C" __global__ void __launch_bounds__(256, 2) hazard_aggressor(const uint4* __restrict__ w, int nfrag, int steps, int mode,
                                                                      float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = w[(blockIdx.x * 64 + i) % (nfrag * 64)];
    __syncthreads();
    float16v acc[6];
    for (int j = 0; j < 6; ++j) acc[j] = float16v{0};
    float16v ballast[BALLAST];
    for (int j = 0; j < BALLAST; ++j)
        for (int i = 0; i < 16; ++i) ballast[j][i] = (float)(lane + i + j);
    uint4 q[2][6];
    const uint4* wb = w + (size_t)((blockIdx.x * 131 + wave * 17) % (nfrag - 6)) * 64 + lane;
    for (int f = 0; f < 6; ++f) q[0][f] = q[1][f] = wb[f * 64];
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        for (int j = 0; j < BALLAST; ++j) asm volatile("" : "+v"(ballast[j]));
        if (mode & 1) {
            const uint4* nx = w + (size_t)((blockIdx.x * 131 + wave * 17 + (s + 1) * 6) % (nfrag - 6)) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 6; ++f) q[cur ^ 1][f] = nx[f * 64];
        }
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            half8 a0;
            if (mode & 2) a0 = *reinterpret_cast<const half8*>(lds + ((s * 6 + f) & 15) * 1024 + lane * 16);
            else a0 = __builtin_bit_cast(half8, q[cur][(f + 1) % 6]);
            const half8 b = __builtin_bit_cast(half8, q[cur][f]);
            if (mode & 4) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[f], 0, 0, 0);
                acc[(f + 1) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 1) % 6], 0, 0, 0);
                acc[(f + 2) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 2) % 6], 0, 0, 0);
            } else {
                acc[f][0] += (float)a0[0] + (float)b[1];
            }
        }
    }
    float t = 0.f;
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 16; ++i) t += acc[j][i];
    for (int j = 0; j < BALLAST; ++j) {
        asm volatile("" : "+v"(ballast[j]));
        for (int i = 0; i < 16; ++i) t += ballast[j][i];
    }
    if (t == 12345.678f) sink[0] = t;
}
