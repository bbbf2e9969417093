// Library-independent reproducer attempt for the hazard of HISTORY.md section 3.3: does a gather whose
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
#include <cstdlib>

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

// A victim shaped like the failing grid_pull3d case (tests/golden/synth_grid_pull.npz): ONE workgroup, 240 active lanes, all of
// them sampling a volume of a few hundred floats (every lane hits the same handful of lines), 8 corner loads in flight, each
// loaded value checked in place.  lanebad[l] counts wrong values per lane of the wave, cornerbad[c] per corner.
template <bool SC1>
__global__ void tiny_gather(const float* __restrict__ X, int nx, int ny, int nz, unsigned seed, unsigned* __restrict__ bad,
                            unsigned* __restrict__ lanebad, unsigned* __restrict__ cornerbad, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i >= 240) return;
    unsigned h = (unsigned)i * 2654435761u + seed * 40503u;
    h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    const int ix = (int)(h % (unsigned)(nx - 1)), iy = (int)((h >> 8) % (unsigned)(ny - 1)), iz = (int)((h >> 16) % (unsigned)(nz - 1));
    float v[8];
    uint32_t k[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        k[c] = ((uint32_t)(ix + (c >> 2)) * ny + (uint32_t)(iy + ((c >> 1) & 1))) * nz + (uint32_t)(iz + (c & 1));
        v[c] = ld<SC1>(X + k[c]);
    }
    float acc = 0.f;
    unsigned wrong = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        acc += v[c];
        if (v[c] != texel(k[c])) { ++wrong; atomicAdd(cornerbad + c, 1u); }
    }
    out[i] = acc;
    if (wrong) { atomicAdd(bad, wrong); atomicAdd(lanebad + (i & 63), wrong); }
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
            } else if (MODE == 6) {                          // round 6: the SAME DMA with lanes 48..63 switched off (EXEC partial),
                if (lane < 48)                               // as conv_wino4d issues it (two halo rows = 48 lanes per instruction)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            } else if (MODE == 7 || MODE == 8) {             // conv_wino4d's inline-assembly form (M0 by hand, SGPR base +
                const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)dst));   // 32-bit VGPR offset)
                const unsigned vo = (unsigned)lane * 16u;
                const uint4* base = w + (size_t)__builtin_amdgcn_readfirstlane(f) * 64;
                if (MODE == 8 || lane < 48)                  // 7: lanes 48..63 off; 8: all lanes
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(base), "s"(la) : "memory");
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

int main(int argc, char** argv) {
    const int N = 256;
    const uint32_t nvol = (uint32_t)N * N * N;
    const int ox = 160, oy = 160, oz = 80;
    const int64_t nout = (int64_t)ox * oy * oz;
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const int only = argc > 2 ? atoi(argv[2]) : -1;            // one co-runner mode only
    float *X, *out[2], *sink, *Xs, *outs[2];
    unsigned *bad, *lanebad, *cornerbad;
    uint4* w;
    const int nfrag = 64 * 1024;                               // 64 MiB of "weights"
    const int tx = 5, ty = 6, tz = 14;                         // the tiny victim's volume: 420 floats
    CK(hipMalloc(&X, (size_t)nvol * 4));
    CK(hipMalloc(&Xs, (size_t)tx * ty * tz * 4));
    CK(hipMalloc(&out[0], nout * 4)); CK(hipMalloc(&out[1], nout * 4));
    CK(hipMalloc(&outs[0], 1024)); CK(hipMalloc(&outs[1], 1024));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&lanebad, 64 * 4)); CK(hipMalloc(&cornerbad, 8 * 4));
    CK(hipMalloc(&w, (size_t)nfrag * 1024));
    CK(hipMemset(w, 1, (size_t)nfrag * 1024));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, X, nvol);
    hipLaunchKernelGGL(fill, dim3(1), dim3(256), 0, 0, Xs, (uint32_t)(tx * ty * tz));
    CK(hipDeviceSynchronize());
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    const int NCM = 9;
    const char* cname[NCM] = {"LDS-DMA (global_load_lds_dwordx4) only", "ordinary loads + ds_write, same bytes", "no co-runner",
                              "LDS-DMA ring, counted waits, ds_read_b128, MFMA", "LDS-DMA, waves END with DMA in flight",
                              "LDS-DMA into LDS above 64 KiB (96 KiB allocated)",
                              "LDS-DMA, lanes 48..63 OFF (builtin)", "LDS-DMA, lanes 48..63 OFF (asm, M0 by hand)",
                              "LDS-DMA, all lanes (asm, M0 by hand)"};
    const char* gname[2] = {"ordinary global_load_dword", "agent scope (sc1)"};
    printf("# victims: gather 160x160x80 from a 256^3 volume, and tiny_gather (one workgroup, 240 lanes, 420-float volume, 20 launches\n"
           "# per round), each on two streams, co-runner on a third; %d rounds each; wrong texels\n", rounds);
    CK(hipFuncSetAttribute((const void*)corunner<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    const int order[NCM] = {2, 6, 7, 8, 0, 3, 1, 4, 5};
    for (int oi = 0; oi < NCM; ++oi) {
        const int cm = order[oi];
        if (only >= 0 && cm != only) continue;
        for (int vict = 0; vict < 2; ++vict)
        for (int gm = 0; gm < 2; ++gm) {
            CK(hipMemset(bad, 0, 4)); CK(hipMemset(lanebad, 0, 256)); CK(hipMemset(cornerbad, 0, 32));
            CK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r) {
                if (cm == 0) hipLaunchKernelGGL(corunner<0>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 1) hipLaunchKernelGGL(corunner<1>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 5) hipLaunchKernelGGL(corunner<5>, dim3(512), dim3(256), 98304, s[2], w, nfrag, 40, sink);
                if (cm == 4) hipLaunchKernelGGL(corunner<4>, dim3(2048), dim3(256), 32768, s[2], w, nfrag, 10, sink);
                if (cm == 3) hipLaunchKernelGGL(corunner_mfma, dim3(512), dim3(256), 65536, s[2], w, nfrag, 60, sink);
                if (cm == 6) hipLaunchKernelGGL(corunner<6>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 7) hipLaunchKernelGGL(corunner<7>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                if (cm == 8) hipLaunchKernelGGL(corunner<8>, dim3(512), dim3(256), 32768, s[2], w, nfrag, 40, sink);
                for (int l = 0; l < 2; ++l) {
                    if (vict == 0) {
                        if (gm == 0) hipLaunchKernelGGL(gather<false>, dim3(2048), dim3(256), 0, s[l], X, N, ox, oy, oz, out[l], bad);
                        else hipLaunchKernelGGL(gather<true>, dim3(2048), dim3(256), 0, s[l], X, N, ox, oy, oz, out[l], bad);
                    } else {
                        for (int k = 0; k < 20; ++k) {
                            if (gm == 0) hipLaunchKernelGGL(tiny_gather<false>, dim3(1), dim3(256), 0, s[l], Xs, tx, ty, tz, (unsigned)(r * 20 + k), bad, lanebad, cornerbad, outs[l]);
                            else hipLaunchKernelGGL(tiny_gather<true>, dim3(1), dim3(256), 0, s[l], Xs, tx, ty, tz, (unsigned)(r * 20 + k), bad, lanebad, cornerbad, outs[l]);
                        }
                    }
                }
                if ((r & 15) == 15) CK(hipDeviceSynchronize());        // bound the queue depth
            }
            CK(hipDeviceSynchronize());
            unsigned h = 0, lb[64], cb[8];
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(lb, lanebad, 256, hipMemcpyDeviceToHost));
            CK(hipMemcpy(cb, cornerbad, 32, hipMemcpyDeviceToHost));
            const long long tot = vict == 0 ? (long long)rounds * 2 * nout * 8 : (long long)rounds * 2 * 20 * 240 * 8;
            printf("co-runner: %-48s victim: %-11s loads: %-28s wrong texels: %u of %lld\n", cname[cm], vict ? "tiny_gather" : "gather",
                   gname[gm], h, tot);
            if (vict == 1 && h) {
                printf("    per lane:");
                for (int l = 0; l < 64; ++l) if (lb[l]) printf(" %d:%u", l, lb[l]);
                printf("\n    per corner:");
                for (int c = 0; c < 8; ++c) printf(" %u", cb[c]);
                printf("\n");
            }
            fflush(stdout);
        }
    }
    return 0;
}
