// How many ds_read_b128 per v_mfma_f32_32x32x16_f16 a wave can take before the LDS, not the matrix pipe, sets the pace:
// the question behind conv_wino2's tile (4 positions x 32 rows x 64 columns per wave = one operand read per MFMA).
// Per iteration 24 MFMAs (8 accumulators x 3) and R reads (lane-linear 1 KiB each, conflict-free), one read behind every
// 24/R-th MFMA; WAVES per SIMD = 1 (256 threads) or 2 (512 threads, one workgroup per CU either way: 100 KB of LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int R, int NT, bool IDLE2>
__global__ void __launch_bounds__(NT, NT / 256) loop(const uint4* __restrict__ seed, int iters, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += NT) reinterpret_cast<uint4*>(lds)[i] = seed[i & 4095];
    __syncthreads();
    floatx16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    half8 f[24];
    for (int k = 0; k < 24; ++k) f[k] = *reinterpret_cast<const half8*>(lds + k * 1024 + lane * 16);
    if (IDLE2 && wave >= 4) {                         // the second wave of every SIMD does nothing but wait
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
        return;
    }
    const unsigned char* base = lds + (wave & 3) * 4096 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned char* b = base + (it & 7) * 2048;
        half8 nf[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) nf[k] = f[k];
#pragma unroll
        for (int k = 0; k < R; ++k) nf[(k * 24) / (R > 0 ? R : 1)] = *reinterpret_cast<const half8*>(b + k * 1024);
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m], f[(m + 7) % 24], acc[m & 7], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (R > 0 && (m * R) / 24 != ((m + 1) * R) / 24) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) f[k] = nf[k];
    }
    float s = 0.f;
    for (int a = 0; a < 8; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * NT + tid] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<_Float16> hs(4096 * 8);
    srand(2);
    for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    uint4* ds; float* dout;
    hipMalloc(&ds, 4096 * 16); hipMalloc(&dout, 256 * 512 * 4);
    hipMemcpy(ds, hs.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t smem = 100 * 1024;
#define RUN(R, NT, IDLE, label)                                                                                          \
    {                                                                                                                    \
        hipFuncSetAttribute((const void*)loop<R, NT, IDLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);      \
        hipLaunchKernelGGL((loop<R, NT, IDLE>), dim3(256), dim3(NT), smem, 0, ds, 200, dout);                            \
        hipEventRecord(e0);                                                                                              \
        hipLaunchKernelGGL((loop<R, NT, IDLE>), dim3(256), dim3(NT), smem, 0, ds, iters, dout);                          \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                                     \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                                      \
        const double waves = 256.0 * ((IDLE) ? 4 : (NT / 64));                                                           \
        const double cyc = ms * 1e-3 * 2.4e9 / iters / 24.0 * ((IDLE || NT == 256) ? 1.0 : 0.5);                         \
        printf("%-44s R=%2d reads / 24 MFMA: %7.2f ms  %6.0f TFLOP/s  (~%.1f cycles @2.4GHz per MFMA per SIMD)\n",      \
               label, R, ms, waves * iters * 24.0 * 32768.0 / ms / 1e9, cyc);                                            \
    }
    RUN(0, 256, false, "1 wave per SIMD");
    RUN(8, 256, false, "1 wave per SIMD");
    RUN(16, 256, false, "1 wave per SIMD");
    RUN(24, 256, false, "1 wave per SIMD");
    RUN(0, 512, false, "2 waves per SIMD, both multiply");
    RUN(8, 512, false, "2 waves per SIMD, both multiply");
    RUN(16, 512, false, "2 waves per SIMD, both multiply");
    RUN(24, 512, false, "2 waves per SIMD, both multiply");
    RUN(24, 512, true, "2 waves per SIMD, second one asleep");
    RUN(16, 512, true, "2 waves per SIMD, second one asleep");
    return 0;
}
