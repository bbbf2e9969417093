// Shared helpers for the brainfm_hip kernels (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/brainfm_hip.h"

#define BFM_WAVE 64

static inline int bfm_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? BFM_OK : BFM_E_LAUNCH;
}

static inline hipStream_t bfm_s(bfm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int bfm_cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int64_t bfm_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Device view of bfm_upsample_t (passed by value to kernels).
struct UpView {
    int d, h, w;
    const int32_t *mapD, *mapH, *mapW, *repD, *repH, *repW;
};

static inline UpView make_upview(const bfm_upsample_t* up) {
    UpView u{};
    if (up) {
        u.d = up->d; u.h = up->h; u.w = up->w;
        u.mapD = up->mapD; u.mapH = up->mapH; u.mapW = up->mapW;
        u.repD = up->repD; u.repH = up->repH; u.repW = up->repW;
    }
    return u;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
