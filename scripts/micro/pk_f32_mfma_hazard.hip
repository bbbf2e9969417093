// Library-free reproducer attempt for the hazard of HISTORY.md section 3.3 as round 6 cornered it (profiles/r06_hazard_root_cause.txt):
// a wave whose VALU code holds a PACKED-FP32 multiply with bank-conflicting sources,
//        v_cvt_f32_i32_e32 v12, v24
//        v_pk_mul_f32 v[24:25], v[32:33], v[36:37] op_sel:[0,1] op_sel_hi:[0,1]      (v32 / v36 and v33 / v37 share a bank)
// loses the LOW half of the product (v24) in lanes 48..63 when a wave that runs a dense chain of MFMAs fed by global loads
// into VGPRs (the tap loop of conv_wino4 / conv_wino4d) shares its SIMD.  Both halves compute the SAME product here
// (op_sel picks v32 * v37 twice), so the victim checks lo == hi bit for bit in place: no reference needed.
//
//   victim<K>   : K = 0 the two instructions above, in inline assembly with fixed registers, 64 times per thread with
//                 fresh operands; K = 1 the same with three independent full-rate VALU instructions between them and
//                 non-conflicting sources (control).
//   aggressor<M>: M = 0 MFMA chain only; 1 MFMA chain + six global_load_dwordx4 per step into the B operands + ds_read_b128
//                 of the A operands (what the tap loop of conv_wino4d does); 2 loads only; 3 nothing (no co-runner).
// Victims on two streams, the aggressor on a third, like the failing test.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/pk_f32_mfma_hazard scripts/micro/pk_f32_mfma_hazard.hip
//   ./scripts/micro/pk_f32_mfma_hazard [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int K>
__global__ void __launch_bounds__(256) victim(const float* __restrict__ in, unsigned* __restrict__ bad, unsigned* __restrict__ lanebad,
                                              float* __restrict__ out, int reps) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    unsigned wrong = 0;
    for (int r = 0; r < reps; ++r) {
        const float a = in[(tid * 7 + r * 13) & 65535];           // v32: the shared factor
        const float b = in[(tid * 11 + r * 29 + 1) & 65535];       // v37: the other factor
        const int one = (tid + r) >= 0 ? 1 : 0;                    // the integer the convert reads from the product's low register
        float lo, hi, cv;
        if (K == 0) {
            asm volatile(
                "v_mov_b32 v32, %3\n\t"
                "v_mov_b32 v33, %3\n\t"
                "v_mov_b32 v36, %4\n\t"
                "v_mov_b32 v37, %4\n\t"
                "v_mov_b32 v24, %5\n\t"
                "v_mul_lo_u32 v24, v24, %5\n\t"
                "v_mul_lo_u32 v12, v24, %5\n\t"
                "v_cvt_f32_i32_e32 v13, v12\n\t"
                "v_cvt_f32_i32_e32 v12, v24\n\t"
                "v_pk_mul_f32 v[24:25], v[32:33], v[36:37] op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                "v_mov_b32 %0, v24\n\t"
                "v_mov_b32 %1, v25\n\t"
                "v_mov_b32 %2, v12\n\t"
                : "=v"(lo), "=v"(hi), "=v"(cv)
                : "v"(a), "v"(b), "v"(one)
                : "v12", "v13", "v24", "v25", "v32", "v33", "v36", "v37");
        } else {
            asm volatile(
                "v_mov_b32 v32, %3\n\t"
                "v_mov_b32 v33, %3\n\t"
                "v_mov_b32 v38, %4\n\t"
                "v_mov_b32 v39, %4\n\t"
                "v_mov_b32 v24, %5\n\t"
                "v_cvt_f32_i32_e32 v12, v24\n\t"
                "v_add_u32 v13, v12, v12\n\t"
                "v_add_u32 v13, v13, v12\n\t"
                "v_add_u32 v13, v13, v12\n\t"
                "v_pk_mul_f32 v[24:25], v[32:33], v[38:39] op_sel:[0,1] op_sel_hi:[0,1]\n\t"
                "v_mov_b32 %0, v24\n\t"
                "v_mov_b32 %1, v25\n\t"
                "v_mov_b32 %2, v12\n\t"
                : "=v"(lo), "=v"(hi), "=v"(cv)
                : "v"(a), "v"(b), "v"(one)
                : "v12", "v13", "v24", "v25", "v32", "v33", "v38", "v39");
        }
        if (__float_as_uint(lo) != __float_as_uint(hi) || cv != 1.0f) ++wrong;
        acc += lo + hi;
    }
    out[tid] = acc;
    if (wrong) { atomicAdd(bad, wrong); atomicAdd(lanebad + (threadIdx.x & 63), wrong); }
}

template <int M>
__global__ void __launch_bounds__(256, 2) aggressor(const uint4* __restrict__ w, int nfrag, int steps, float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = w[(blockIdx.x * 64 + i) % (nfrag * 64)];
    __syncthreads();
    float16v acc[6];
    for (int j = 0; j < 6; ++j) acc[j] = float16v{0};
    uint4 q[2][6];
    const uint4* wb = w + (size_t)((blockIdx.x * 131 + wave * 17) % (nfrag - 6)) * 64 + lane;
    if (M == 1 || M == 2)
        for (int f = 0; f < 6; ++f) q[0][f] = wb[f * 64];
    else
        for (int f = 0; f < 6; ++f) q[0][f] = q[1][f] = uint4{0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        if (M == 1 || M == 2) {
            const uint4* nx = w + (size_t)((blockIdx.x * 131 + wave * 17 + (s + 1) * 6) % (nfrag - 6)) * 64 + lane;
#pragma unroll
            for (int f = 0; f < 6; ++f) q[cur ^ 1][f] = nx[f * 64];          // six global_load_dwordx4 into VGPRs per step
        }
        if (M == 0 || M == 1) {
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const half8 a0 = *reinterpret_cast<const half8*>(lds + ((s * 6 + f) & 15) * 1024 + lane * 16);   // ds_read_b128
                const half8 b = __builtin_bit_cast(half8, q[cur][f]);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[f], 0, 0, 0);
                acc[(f + 1) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 1) % 6], 0, 0, 0);
                acc[(f + 2) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc[(f + 2) % 6], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int f = 0; f < 6; ++f) acc[f][0] += __uint_as_float(q[cur][f].x & 0x3fffffffu);
        }
    }
    float t = 0.f;
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 16; ++i) t += acc[j][i];
    if (t == 12345.678f) sink[0] = t;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    const int nfrag = 64 * 1024;
    uint4* w;
    float *in, *out[2], *sink;
    unsigned *bad, *lanebad;
    CK(hipMalloc(&w, (size_t)nfrag * 1024));
    CK(hipMemset(w, 0x3c, (size_t)nfrag * 1024));
    CK(hipMalloc(&in, 65536 * 4));
    CK(hipMalloc(&out[0], 1 << 20)); CK(hipMalloc(&out[1], 1 << 20));
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&lanebad, 256));
    {
        float* h = (float*)malloc(65536 * 4);
        unsigned x = 12345u;
        for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; h[i] = 0.01f + (float)(x >> 8) * (1.0f / 16777216.0f); }
        CK(hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice));
        free(h);
    }
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    const char* aname[4] = {"MFMA chain only", "MFMA chain + global_load_dwordx4 -> VGPR + ds_read_b128", "loads only", "no co-runner"};
    const char* vname[2] = {"cvt ; pk_mul with bank-conflicting sources", "control: spaced, no bank conflict"};
    printf("# victim: 15 workgroups x 256 threads x 64 repetitions, 8 launches per round on each of two streams; aggressor: 6 launches\n"
           "# per round on a third stream; %d rounds each; a wrong result = low half != high half of the packed product\n", rounds);
    const int order[4] = {3, 1, 0, 2};
    for (int oi = 0; oi < 4; ++oi) {
        const int am = order[oi];
        for (int vk = 0; vk < 2; ++vk) {
            CK(hipMemset(bad, 0, 4)); CK(hipMemset(lanebad, 0, 256));
            CK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r) {
                for (int k = 0; k < 6; ++k) {
                    if (am == 0) hipLaunchKernelGGL(aggressor<0>, dim3(1024), dim3(256), 16384, s[2], w, nfrag, 400, sink);
                    if (am == 1) hipLaunchKernelGGL(aggressor<1>, dim3(1024), dim3(256), 16384, s[2], w, nfrag, 400, sink);
                    if (am == 2) hipLaunchKernelGGL(aggressor<2>, dim3(1024), dim3(256), 16384, s[2], w, nfrag, 400, sink);
                }
                for (int k = 0; k < 8; ++k)
                    for (int l = 0; l < 2; ++l) {
                        if (vk == 0) hipLaunchKernelGGL(victim<0>, dim3(15), dim3(256), 0, s[l], in, bad, lanebad, out[l], 64);
                        else hipLaunchKernelGGL(victim<1>, dim3(15), dim3(256), 0, s[l], in, bad, lanebad, out[l], 64);
                    }
                CK(hipDeviceSynchronize());
            }
            unsigned h = 0, lb[64];
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(lb, lanebad, 256, hipMemcpyDeviceToHost));
            printf("aggressor: %-56s victim: %-44s wrong: %u of %lld\n", aname[am], vname[vk], h, (long long)rounds * 16 * 15 * 256 * 64);
            if (h) {
                printf("    per lane:");
                for (int l = 0; l < 64; ++l) if (lb[l]) printf(" %d:%u", l, lb[l]);
                printf("\n");
            }
            fflush(stdout);
        }
    }
    return 0;
}
