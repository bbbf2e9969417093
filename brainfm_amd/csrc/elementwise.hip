// Small HBM-bound per-voxel kernels shared by the stand-alone processors /
// post-processor (joiner.py:69-77,149-157; Trainer/models/__init__.py:272-354)
// and by the synthesis augmentations (Generator/utils.py:568-638,
// Generator/datasets.py:306-372).  Strided so that channel slices of a
// channels-last buffer can be read in place.
#include "bfm_common.h"

namespace {

inline int grid_for(int64_t n, int tpb = 256, int cap = 4096) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__device__ __forceinline__ float ew_u(int op, float x, float a, float b) {
    switch (op) {
        case BFM_EW_EXP: return expf(x);
        case BFM_EW_AFFINE: return x * a + b;
        case BFM_EW_CLAMP: return fminf(fmaxf(x, a), b);
        case BFM_EW_CLAMP_MIN: return x < a ? a : x;
        // a * (x/a)^b through the hardware log2 / exp2 (1 ulp each): within 2e-6 of max|result| of powf over the
        // augmentation's range (x/a in [0, 1], b = exp(N(0, 0.1))), at a fifth of its instructions; (0)^b = 0, negative
        // bases give NaN like pow with a non-integer exponent
        case BFM_EW_GAMMA: return a * __builtin_amdgcn_exp2f(b * __builtin_amdgcn_logf(x / a));
        case BFM_EW_SIGMOID: return 1.f / (1.f + expf(-x));
        case BFM_EW_DIV: return x / a;
        case BFM_EW_NONZERO: return x != 0.f ? 1.f : 0.f;
        case BFM_EW_SUB_DIV: return (x - a) / b;
        case BFM_EW_GE: return x >= a ? 1.f : 0.f;
        case BFM_EW_NAN_TO_NUM: return x != x ? 0.f : (x == INFINITY ? 3.402823466e+38f : (x == -INFINITY ? -3.402823466e+38f : x));
        default: return x;
    }
}

__device__ __forceinline__ float ew_b(int op, float p, float q, float a) {
    float r;
    switch (op) {
        case BFM_EW_ADD: return p + q;
        case BFM_EW_MUL: return p * q;
        case BFM_EW_MUL_EXP: return p * expf(q);
        case BFM_EW_AXPY_CLAMP0: r = p + a * q; return r < 0.f ? 0.f : r;   // add_noise, utils.py:633-638
        case BFM_EW_AXPY: return p + a * q;
        case BFM_EW_DIV2: return p / q;
        case BFM_EW_ZERO_WHERE_ZERO: return q == 0.f ? 0.f : p;
        default: return p;
    }
}

__global__ void ew_unary(int op, const float* __restrict__ in, int64_t is, float* __restrict__ out, int64_t os,
                         int64_t n, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i * os] = ew_u(op, in[i * is], a, b);
}

// contiguous, 16-byte aligned operands: four elements per lane and access
__global__ void ew_unary4(int op, const float4* __restrict__ in, float4* __restrict__ out, int64_t n4, float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        out[i] = make_float4(ew_u(op, v.x, a, b), ew_u(op, v.y, a, b), ew_u(op, v.z, a, b), ew_u(op, v.w, a, b));
    }
}

__global__ void ew_binary(int op, const float* __restrict__ x, int64_t xs, const float* __restrict__ y, int64_t ys,
                          float* __restrict__ out, int64_t os, int64_t n, float a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i * os] = ew_b(op, x[i * xs], y[i * ys], a);
}

__global__ void ew_binary4(int op, const float4* __restrict__ x, const float4* __restrict__ y, float4* __restrict__ out,
                           int64_t n4, float a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 p = x[i], q = y[i];
        out[i] = make_float4(ew_b(op, p.x, q.x, a), ew_b(op, p.y, q.y, a), ew_b(op, p.z, q.z, a), ew_b(op, p.w, q.w, a));
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// softmax over the last (channel) axis of a channels-last tensor, one thread per voxel
__global__ void softmax_cl(const float* __restrict__ x, int64_t xrs, int C, float* __restrict__ y, int64_t yrs,
                           int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = x + i * xrs;
        float* o = y + i * yrs;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, r[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(r[c] - m);
        for (int c = 0; c < C; ++c) o[c] = expf(r[c] - m) / s;
    }
}

__global__ void argmax_lut_cl(const float* __restrict__ p, int64_t prs, int C, const int32_t* __restrict__ lut,
                              int64_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = p + i * prs;
        int best = 0;
        float bv = r[0];
        for (int c = 1; c < C; ++c) if (r[c] > bv) { bv = r[c]; best = c; }
        out[i] = lut ? (int64_t)lut[best] : (int64_t)best;
    }
}

__device__ __forceinline__ float fake_term(float v, float add, float gain) {
    return gain * (1.f - (tanhf(2.f * (v + add)) + 1.f) / 2.f);
}

// fake_cortical, Trainer/models/__init__.py:327-338 (a = 2)
__global__ void fake_cortical(const float* __restrict__ d, int64_t rs, int nd, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* r = d + i * rs;
        float f = fake_term(r[1], 0.3f, 70.f) + fake_term(r[0], 0.f, 40.f);
        if (nd == 4) f = f + (fake_term(r[3], 0.3f, 70.f) + fake_term(r[2], 0.f, 40.f));
        out[i] = f;
    }
}

// encode_pathology (Generator/datasets.py:496-518): I + Pprob * (mu[round(P)] + sigma[round(P)] * randn), clamp >= 0
__global__ void pathology_encode(const float* __restrict__ I, const float* __restrict__ P,
                                 const float* __restrict__ Pprob, const float* __restrict__ rn, float mu0, float mu1,
                                 float s0, float s1, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool one = rintf(P[i]) >= 1.f;
        const float v = I[i] + Pprob[i] * ((one ? mu1 : mu0) + (one ? s1 : s0) * rn[i]);
        out[i] = v < 0.f ? 0.f : v;
    }
}

// max |x| over `rows` runs of `len` floats, `row_stride` apart (a contiguous tensor: one row), folded into *out with an
// integer atomicMax on the bit pattern (non-negative floats order like unsigned integers; *out starts at +0): one launch,
// any order.  NaNs do not take part (fmaxf), as max |w| of finite weights is what the callers want.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t rows, int64_t len, int64_t row_stride,
                                                     unsigned* __restrict__ out) {
    float m = 0.f;
    if ((len & 3) == 0 && (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // 16-byte loads, rows dealt to blockIdx.y (no 64-bit division per element: the scalar loop below ran at a quarter
        // of the memory rate over the 264 M weights of a training step)
        const int64_t q = len >> 2;
        for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
            const float4* p = reinterpret_cast<const float4*>(x + r * row_stride);
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (int64_t)gridDim.x * blockDim.x) {
                const float4 v = p[i];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
        }
    } else {
        const int64_t n = rows * len;
        const int64_t nthr = (int64_t)gridDim.x * gridDim.y * blockDim.x;
        for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += nthr) {
            const int64_t r = i / len;
            m = fmaxf(m, fabsf(x[r * row_stride + (i - r * len)]));
        }
    }
    m = wave_reduce_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

}  // namespace

extern "C" int bfm_absmax_f32(const float* x, int64_t rows, int64_t len, int64_t row_stride, float* out_zeroed,
                              bfm_stream_t stream) {
    if (!x || !out_zeroed || rows <= 0 || len <= 0 || row_stride < len) return BFM_E_ARG;
    // a contiguous tensor is one long row: blocks along x; many short rows (a column slice of the weights): rows along y
    const int gx = grid_for(bfm_cdiv64(len, 4), 256, 1024);
    const int gy = (int)std::min<int64_t>(rows, std::max<int64_t>(1, 2048 / gx));
    hipLaunchKernelGGL(absmax_kernel, dim3(gx, gy), dim3(256), 0, bfm_s(stream), x, rows, len, row_stride,
                       reinterpret_cast<unsigned*>(out_zeroed));
    return bfm_launch_status();
}

extern "C" int bfm_pathology_encode(const float* I, const float* P, const float* Pprob, const float* randn, float mu0,
                                    float mu1, float s0, float s1, int64_t n, float* out, bfm_stream_t stream) {
    if (!I || !P || !Pprob || !randn || !out || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(pathology_encode, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), I, P, Pprob, randn, mu0, mu1, s0,
                       s1, n, out);
    return bfm_launch_status();
}

extern "C" int bfm_ew_unary(int op, const float* in, int64_t in_stride, float* out, int64_t out_stride, int64_t n,
                            float a, float b, bfm_stream_t stream) {
    if (!in || !out || n <= 0 || in_stride <= 0 || out_stride <= 0) return BFM_E_ARG;
    if (in_stride == 1 && out_stride == 1 && aligned16(in) && aligned16(out) && n >= 1024) {
        const int64_t n4 = n >> 2;
        hipLaunchKernelGGL(ew_unary4, dim3(grid_for(n4)), dim3(256), 0, bfm_s(stream), op, (const float4*)in, (float4*)out,
                           n4, a, b);
        if (n & 3)
            hipLaunchKernelGGL(ew_unary, dim3(1), dim3(64), 0, bfm_s(stream), op, in + (n4 << 2), (int64_t)1,
                               out + (n4 << 2), (int64_t)1, n & 3, a, b);
        return bfm_launch_status();
    }
    hipLaunchKernelGGL(ew_unary, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), op, in, in_stride, out, out_stride, n,
                       a, b);
    return bfm_launch_status();
}

extern "C" int bfm_ew_binary(int op, const float* x, int64_t xs, const float* y, int64_t ys, float* out, int64_t os,
                             int64_t n, float a, bfm_stream_t stream) {
    if (!x || !y || !out || n <= 0 || xs <= 0 || ys < 0 || os <= 0) return BFM_E_ARG;
    if (xs == 1 && ys == 1 && os == 1 && aligned16(x) && aligned16(y) && aligned16(out) && n >= 1024) {
        const int64_t n4 = n >> 2;
        hipLaunchKernelGGL(ew_binary4, dim3(grid_for(n4)), dim3(256), 0, bfm_s(stream), op, (const float4*)x,
                           (const float4*)y, (float4*)out, n4, a);
        if (n & 3)
            hipLaunchKernelGGL(ew_binary, dim3(1), dim3(64), 0, bfm_s(stream), op, x + (n4 << 2), (int64_t)1,
                               y + (n4 << 2), (int64_t)1, out + (n4 << 2), (int64_t)1, n & 3, a);
        return bfm_launch_status();
    }
    hipLaunchKernelGGL(ew_binary, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), op, x, xs, y, ys, out, os, n, a);
    return bfm_launch_status();
}

extern "C" int bfm_softmax_cl(const float* x, int64_t x_row_stride, int C, float* y, int64_t y_row_stride, int64_t n,
                              bfm_stream_t stream) {
    if (!x || !y || n <= 0 || C <= 0 || x_row_stride < C || y_row_stride < C) return BFM_E_ARG;
    hipLaunchKernelGGL(softmax_cl, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), x, x_row_stride, C, y, y_row_stride,
                       n);
    return bfm_launch_status();
}

extern "C" int bfm_argmax_lut_cl(const float* p, int64_t row_stride, int C, const int32_t* lut, int64_t* out,
                                 int64_t n, bfm_stream_t stream) {
    if (!p || !out || n <= 0 || C <= 0 || row_stride < C) return BFM_E_ARG;
    hipLaunchKernelGGL(argmax_lut_cl, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), p, row_stride, C, lut, out, n);
    return bfm_launch_status();
}

extern "C" int bfm_fake_cortical(const float* dist, int64_t row_stride, int n_dist, float* out, int64_t n,
                                 bfm_stream_t stream) {
    if (!dist || !out || n <= 0 || (n_dist != 2 && n_dist != 4) || row_stride < n_dist) return BFM_E_ARG;
    hipLaunchKernelGGL(fake_cortical, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), dist, row_stride, n_dist, out, n);
    return bfm_launch_status();
}
