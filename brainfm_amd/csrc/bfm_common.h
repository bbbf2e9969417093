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

// Stage ROWS rows of LEN contiguous floats (row r at src + r * src_pitch) into LDS rows of LDS_PITCH floats (any pitch,
// e.g. odd against bank conflicts), each value multiplied by `scale`, with a block of NT threads.  vec16: rows start on
// 16-byte boundaries -- every thread then issues ALL its 16-byte loads before the first LDS store (the weight-packing
// kernels of the training step walked these rows with one dependent 4-byte load per loop trip: latency-bound at a
// fraction of the memory rate).
template <int ROWS, int LEN, int LDS_PITCH, int NT>
__device__ __forceinline__ void bfm_stage_rows(const float* __restrict__ src, int64_t src_pitch, float* __restrict__ lds,
                                               float scale, bool vec16) {
    static_assert(LEN % 4 == 0, "row length in floats must be a multiple of 4");
    if (vec16) {
        constexpr int Q = LEN / 4, N4 = ROWS * Q, PER = (N4 + NT - 1) / NT;
        float4 v[PER];
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int i = (int)threadIdx.x + m * NT;
            if (i < N4) {
                const int r = i / Q, c = i - r * Q;
                v[m] = *reinterpret_cast<const float4*>(src + (int64_t)r * src_pitch + 4 * c);
            }
        }
#pragma unroll
        for (int m = 0; m < PER; ++m) {
            const int i = (int)threadIdx.x + m * NT;
            if (i < N4) {
                const int r = i / Q, c = i - r * Q;
                float* d = lds + r * LDS_PITCH + 4 * c;
                d[0] = v[m].x * scale; d[1] = v[m].y * scale; d[2] = v[m].z * scale; d[3] = v[m].w * scale;
            }
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * LEN; i += NT) {
            const int r = i / LEN, c = i - r * LEN;
            lds[r * LDS_PITCH + c] = src[(int64_t)r * src_pitch + c] * scale;
        }
    }
}
