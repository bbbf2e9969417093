// The inner loop of the conv kernels without their staging: per "tap", 8 ds_read_b128 (A hi/lo fragments of 4 row blocks),
// 4 uint4 global loads (B hi/lo of 2 column blocks, L2 resident) and 24 v_mfma_f32_32x32x16_f16 (3 split passes) into 128
// accumulators -- how far operand delivery alone pulls the sustained matrix rate below the register-only loop (mfma_peak).
//   variants: 0 = LDS + global feeds, 1 = LDS feed only (B in registers), 2 = no feeds (register operands, same MFMA pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ void __launch_bounds__(256, 2) feed_loop(const uint4* __restrict__ w, const uint4* __restrict__ seed, int taps,
                                                    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40960 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = seed[i & 4095];
    __syncthreads();
    floatx16 acc[4][2];
    for (int mb = 0; mb < 4; ++mb)
        for (int nb = 0; nb < 2; ++nb)
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
    const uint4* wb = w + (size_t)wave * 9216 * 4 + lane;           // this wave's stream: taps x 4 fragments x 64 lanes
    uint4 bq[4];
    for (int f = 0; f < 4; ++f) bq[f] = wb[f * 64];
    half8 areg[4][2];
    for (int mb = 0; mb < 4; ++mb)
        for (int hl = 0; hl < 2; ++hl)
            areg[mb][hl] = *reinterpret_cast<const half8*>(lds + ((mb * 2 + hl) * 1024 + lane * 16));
    for (int t = 0; t < taps; ++t) {
        const int tt = t % 144;
        if (VAR == 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f) bq[f] = wb[(size_t)tt * 256 + f * 64];
        }
        const int toff = (tt % 18) * 1024;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            half8 a[2];
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                if (VAR <= 1) a[hl] = *reinterpret_cast<const half8*>(lds + toff + (mb * 2 + hl) * 2048 + lane * 16);
                else a[hl] = areg[mb][hl];
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const half8 bhi = __builtin_bit_cast(half8, bq[nb * 2]);
                const half8 blo = __builtin_bit_cast(half8, bq[nb * 2 + 1]);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bhi, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], blo, acc[mb][nb], 0, 0, 0);
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bhi, acc[mb][nb], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int mb = 0; mb < 4; ++mb)
        for (int nb = 0; nb < 2; ++nb)
            for (int i = 0; i < 16; ++i) s += acc[mb][nb][i];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const int taps = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t nw = (size_t)4 * 9216 * 4;                         // uint4 entries: 4 waves x 144 taps x 4 fragments x 64 lanes
    std::vector<_Float16> hw(nw * 8), hs(4096 * 8);
    srand(2);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    uint4 *dw, *ds;
    float* dout;
    hipMalloc(&dw, nw * 16); hipMalloc(&ds, 4096 * 16); hipMalloc(&dout, 512 * 256 * 4);
    hipMemcpy(dw, hw.data(), nw * 16, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int var, int n) {
        if (var == 0) hipLaunchKernelGGL(feed_loop<0>, dim3(512), dim3(256), 40960, 0, dw, ds, n, dout);
        if (var == 1) hipLaunchKernelGGL(feed_loop<1>, dim3(512), dim3(256), 40960, 0, dw, ds, n, dout);
        if (var == 2) hipLaunchKernelGGL(feed_loop<2>, dim3(512), dim3(256), 40960, 0, dw, ds, n, dout);
    };
    const char* names[3] = {"LDS A + L2 B feeds", "LDS A feed, B in registers", "register operands"};
    for (int var = 0; var < 3; ++var) {
        run(var, 500);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            run(var, taps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = 512.0 * 4 * taps * 24 * 2.0 * 32 * 32 * 16;
            printf("%-28s rep %d: %.2f ms  %.0f TFLOP/s (f16 issue)\n", names[var], rep, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
