// Library-independent reproducer attempt for the hazard of DESIGN.md section 3.3: does a gather whose
// ordinary vector loads re-use lines from the per-CU L1 return wrong texels while ANOTHER kernel on another stream fills
// its LDS with global_load_lds_dwordx4 (LDS-DMA)?  Two kernels, no library:
//   gather<SC1>  : every thread samples 8 neighbouring texels of a 256^3 float volume whose content is a known function
//                  of the index, at a smooth (rotated) coordinate, and checks EACH loaded value in place; mismatches are
//                  counted.  SC1 = false: ordinary global_load_dword; true: agent-scope (sc1) loads.
//   corunner<M>  : M = 0: nothing but global_load_lds_dwordx4 of 1-KiB fragments into its own LDS + waits + barriers;
//                  M = 1: the same bytes by ordinary global_load_dwordx4 + ds_write (no LDS-DMA); M = 2: off.
// The gather runs on two streams at once (like two tile lanes), the co-runner on a third, 200 rounds per combination.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/l1_ldsdma_hazard scripts/micro/l1_ldsdma_hazard.hip && ./scripts/micro/l1_ldsdma_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__host__ __device__ inline float texel(uint32_t k) {
    uint32_t h = k * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return (float)(h & 0xFFFFFF) * (1.0f / 16777216.0f) + 1.0f;
}

__global__ void fill(float* v, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = texel(i);
}

template <bool SC1>
__device__ __forceinline__ float ld(const float* p) {
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <bool SC1>
__global__ void gather(const float* __restrict__ X, int N, int ox, int oy, int oz, float* __restrict__ out,
                       unsigned* __restrict__ bad) {
    const int64_t n = (int64_t)ox * oy * oz;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % oz), y = (int)((i / oz) % oy), x = (int)(i / ((int64_t)oy * oz));
        // a rotated, scaled grid inside the volume: neighbouring lanes re-use texel lines
        const float fx = 40.f + 0.98f * x + 0.10f * y - 0.05f * z;
        const float fy = 30.f - 0.10f * x + 0.97f * y + 0.08f * z;
        const float fz = 50.f + 0.05f * x - 0.08f * y + 0.99f * z;
        const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
        float acc = 0.f;
        unsigned wrong = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t k = ((uint32_t)(ix + (c >> 2)) * N + (uint32_t)(iy + ((c >> 1) & 1))) * N + (uint32_t)(iz + (c & 1));
            const float v = ld<SC1>(X + k);
            wrong += v != texel(k);
            acc += v;
        }
        out[i] = acc;
        if (wrong) atomicAdd(bad, wrong);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) corunner(const uint4* __restrict__ w, int nfrag, int rounds, float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                        // 8 fragments of 1 KiB per wave and round -> 32 KiB of LDS
            const int f = (blockIdx.x * 131 + r * 32 + wave * 8 + i) % nfrag;
            const uint4* src = w + (size_t)f * 64 + lane;
            char* dst = lds + (MODE == 5 ? 65536 : 0) + (wave * 8 + i) * 1024;   // M = 5: slots ABOVE 64 KiB of LDS
            if (MODE == 0 || MODE == 4 || MODE == 5) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else {
                const uint4 v = *src;
                reinterpret_cast<uint4*>(dst)[lane] = v;
            }
        }
        if (MODE == 4 && r == rounds - 1) break;             // M = 4: the wave ENDS with its last eight LDS-DMA loads in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s += reinterpret_cast<const float*>(lds + (MODE == 5 ? 65536 : 0))[(threadIdx.x * 37 + r) & 8191];
        __syncthreads();
    }
    if (s == 12345.678f) sink[0] = s;                        // keeps the LDS reads alive
}

// M = 3: what a conv kernel of the library does around its LDS-DMA -- a two-slot ring with COUNTED waits (the next slot's
// DMA in flight while this one is consumed), ds_read_b128 of the fragments and a chain of MFMAs on them, 64 KiB of LDS
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) corunner_mfma(const uint4* __restrict__ w, int nfrag, int rounds, float* __restrict__ sink) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float16v acc = {0};
    auto issue = [&](int r, int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = (blockIdx.x * 131 + r * 32 + wave * 8 + i) % nfrag;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + (size_t)f * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(lds + slot * 32768 + (wave * 8 + i) * 1024),
                                             16, 0, 0);
        }
    };
    issue(0, 0);
    for (int r = 0; r < rounds; ++r) {
        const int slot = r & 1;
        if (r + 1 < rounds) {
            issue(r + 1, slot ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // this slot's eight DMAs have landed, the next eight fly
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const half8 a = *reinterpret_cast<const half8*>(lds + slot * 32768 + (((wave + i) & 3) * 8 + i) * 1024 + lane * 16);
            const half8 b = *reinterpret_cast<const half8*>(lds + slot * 32768 + (wave * 8 + ((i + 3) & 7)) * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    const int N = 256;
    const uint32_t nvol = (uint32_t)N * N * N;
    const int ox = 160, oy = 160, oz = 80;
    const int64_t nout = (int64_t)ox * oy * oz;
    float *X, *out[2], *sink;
    unsigned* bad;
    uint4* w;
    const int nfrag = 64 * 1024;                               // 64 MiB of "weights"
    CK(hipMalloc(&X, (size_t)nvol * 4));
    CK(hipMalloc(&out[0], nout * 4)); CK(hipMalloc(&out[1], nout * 4));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&w, (size_t)nfrag * 1024));
    CK(hipMemset(w, 1, (size_t)nfrag * 1024));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, X, nvol);
    CK(hipDeviceSynchronize());
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    const char* cname[6] = {"LDS-DMA (global_load_lds_dwordx4) only", "ordinary loads + ds_write, same bytes", "no co-runner",
                            "LDS-DMA ring, counted waits, ds_read_b128, MFMA", "LDS-DMA, waves END with DMA in flight",
                            "LDS-DMA into LDS above 64 KiB (96 KiB allocated)"};
    const char* gname[2] = {"ordinary global_load_dword", "agent scope (sc1)"};
    const int rounds = 400;
    printf("# gather 160x160x80 from a 256^3 volume on two streams, co-runner on a third; %d rounds each; wrong texels\n", rounds);
    CK(hipFuncSetAttribute((const void*)corunner<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    for (int cm = 0; cm < 6; ++cm) {
        for (int gm = 0; gm < 2; ++gm) {
            CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r) {
                if (cm == 0) hipLaunchKernelGGL(corunner<0>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 1) hipLaunchKernelGGL(corunner<1>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 5) hipLaunchKernelGGL(corunner<5>, dim3(512), dim3(256), 98304, s[2], w, nfrag, 40, sink);
                if (cm == 4) hipLaunchKernelGGL(corunner<4>, dim3(2048), dim3(256), 32768, s[2], w, nfrag, 10, sink);
                if (cm == 3) hipLaunchKernelGGL(corunner_mfma, dim3(512), dim3(256), 65536, s[2], w, nfrag, 60, sink);
                for (int l = 0; l < 2; ++l) {
                    if (gm == 0) hipLaunchKernelGGL(gather<false>, dim3(2048), dim3(256), 0, s[l], X, N, ox, oy, oz, out[l], bad);
                    else hipLaunchKernelGGL(gather<true>, dim3(2048), dim3(256), 0, s[l], X, N, ox, oy, oz, out[l], bad);
                }
            }
            CK(hipDeviceSynchronize());
            unsigned h = 0;
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("co-runner: %-48s gather loads: %-28s wrong texels: %u of %lld\n", cname[cm], gname[gm], h,
                   (long long)rounds * 2 * nout * 8);
        }
    }
    return 0;
}
