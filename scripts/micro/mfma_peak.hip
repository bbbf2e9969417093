// Sustained v_mfma_f32_32x32x16_f16 rate with register operands only (no memory in the loop): what the matrix pipe holds
// on this chip under its power limit, for dense random operands and for zeros.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(const half8* __restrict__ a_in, const half8* __restrict__ b_in, int iters,
                                                 float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = a_in[(t * 4 + i) & 4095]; b[i] = b_in[(t * 4 + i) & 4095]; }
    floatx16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[n & 3], b[(n >> 1) & 3], acc[n], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[t] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<_Float16> ha(4096 * 8), hz(4096 * 8, (_Float16)0.f);
    srand(1);
    for (auto& v : ha) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    half8 *da, *dz;
    float* dout;
    hipMalloc(&da, 4096 * 16); hipMalloc(&dz, 4096 * 16); hipMalloc(&dout, 256 * 8 * 256 * 4 * 4);
    hipMemcpy(da, ha.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipMemcpy(dz, hz.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        for (int zero = 0; zero < 2; ++zero) {
            const int nblk = 256 * blocks_per_cu;
            const half8* src = zero ? dz : da;
            hipLaunchKernelGGL(mfma_loop<8>, dim3(nblk), dim3(256), 0, 0, src, src, 1000, dout);
            hipDeviceSynchronize();
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop<8>, dim3(nblk), dim3(256), 0, 0, src, src, iters, dout);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double fl = (double)nblk * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
                printf("%d wave(s) per SIMD, %s operands, rep %d: %.2f ms  %.0f TFLOP/s\n", blocks_per_cu,
                       zero ? "zero" : "random", rep, ms, fl / ms / 1e9);
            }
        }
    }
    return 0;
}
