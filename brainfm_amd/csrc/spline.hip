// Cubic B-spline resize (utils/interpol/resize.py:13-119 -> grid_pull(interpolation=3, prefilter=True)), the path
// Generator/datasets.py:337-338 takes when `bspline_zooming` is on:
//   1. interpolating-spline prefilter along each axis (utils/interpol/coeff.py:254-344: gain, causal + anticausal
//      recursion with pole sqrt(3)-2, DCT-II ("reflect") boundary conditions ported from scipy's ni_splines.c);
//   2. evaluation of the cubic B-spline at the output positions (utils/interpol/nd.py:36-142, splines.py:40-43,
//      bounds.py:30-38).  A resize samples a separable grid (one coordinate list per axis), so the 64-tap gather of
//      the generic code factors into three 4-tap passes; rounding differs from the reference at the fp32 epsilon level.
// HBM-bound, volumes of ~160^3.
#include "bfm_common.h"

namespace {

struct Dims3 { int n[3]; };

// one thread per line along `axis`; the recursion is sequential in the line index
__global__ void prefilter_dct2_kernel(float* __restrict__ vol, Dims3 d, int axis, float pole, float gain,
                                      const float* __restrict__ init_w /*[n-2]: z^i + z^(2n-1-i), i=1..n-2*/,
                                      float pole_last, float init_scale /* z / (1 - z^2n) */,
                                      float final_scale /* z / (z - 1) */) {
    const int n = d.n[axis];
    const int64_t total = (int64_t)d.n[0] * d.n[1] * d.n[2];
    const int64_t lines = total / n;
    int64_t stride = 1;
    for (int a = 2; a > axis; --a) stride *= d.n[a];
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += (int64_t)gridDim.x * blockDim.x) {
        // line l -> base offset: split l into (outer, inner) around the axis
        const int64_t inner = l % stride, outer = l / stride;
        float* p = vol + outer * stride * n + inner;
        // gain
        for (int i = 0; i < n; ++i) p[i * stride] *= gain;
        // initial condition (coeff.py:141-175)
        const float x00 = p[0];
        float acc = 0.f;
        for (int i = 1; i < n - 1; ++i) acc += p[i * stride] * init_w[i - 1];
        float c0 = acc + (x00 + pole_last * p[(int64_t)(n - 1) * stride]);
        c0 = c0 * init_scale;
        c0 = c0 + x00;
        p[0] = c0;
        // causal
        float prev = c0;
        for (int i = 1; i < n; ++i) {
            prev = fmaf(pole, prev, p[i * stride]);        // inp[i].add_(inp[i-1], alpha=pole): fused on the CPU path
            p[i * stride] = prev;
        }
        // final condition + anticausal (coeff.py:218-226, 262-271)
        float nxt = prev * final_scale;
        p[(int64_t)(n - 1) * stride] = nxt;
        for (int i = n - 2; i >= 0; --i) {
            nxt = (nxt - p[i * stride]) * pole;
            p[i * stride] = nxt;
        }
    }
}

// Contiguous axis (axis == 2): a thread-per-line walk reads 64 different cache lines per instruction.  Here a block
// copies 64 lines into an LDS tile with coalesced loads, each thread filters its line in LDS (odd row stride:
// conflict-free), and the tile goes back coalesced.
__global__ void __launch_bounds__(64) prefilter_dct2_z_kernel(float* __restrict__ vol, int64_t lines, int n, float pole,
                                                              float gain, const float* __restrict__ init_w,
                                                              float pole_last, float init_scale, float final_scale) {
    extern __shared__ float tile[];                  // [64][n | 1]
    const int ld = n | 1;
    const int t = threadIdx.x;
    for (int64_t l0 = (int64_t)blockIdx.x * 64; l0 < lines; l0 += (int64_t)gridDim.x * 64) {
        const int nl = (int)min<int64_t>(64, lines - l0);
        float* base = vol + l0 * n;
        for (int i = t; i < nl * n; i += 64) tile[(i / n) * ld + (i % n)] = base[i] * gain;
        __syncthreads();
        if (t < nl) {
            float* p = tile + t * ld;
            const float x00 = p[0];
            float acc = 0.f;
            for (int i = 1; i < n - 1; ++i) acc += p[i] * init_w[i - 1];
            float c0 = acc + (x00 + pole_last * p[n - 1]);
            c0 = c0 * init_scale;
            c0 = c0 + x00;
            p[0] = c0;
            float prev = c0;
            for (int i = 1; i < n; ++i) {
                prev = fmaf(pole, prev, p[i]);
                p[i] = prev;
            }
            float nxt = prev * final_scale;
            p[n - 1] = nxt;
            for (int i = n - 2; i >= 0; --i) {
                nxt = (nxt - p[i]) * pole;
                p[i] = nxt;
            }
        }
        __syncthreads();
        for (int i = t; i < nl * n; i += 64) base[i] = tile[(i / n) * ld + (i % n)];
        __syncthreads();
    }
}

__device__ __forceinline__ float bspline3(float x) {               // splines.py:40-43 on |x|
    x = fabsf(x);
    const float lo = (x * x * (x - 2.f) * 3.f + 4.f) / 6.f;
    const float t = 2.f - x;
    const float up = t * t * t / 6.f;
    return x < 1.f ? lo : up;
}

__device__ __forceinline__ int reflect_dct2(int i, int n) {         // bounds.py:33-38
    const int n2 = 2 * n;
    int r;
    if (i < 0) {
        int m = (-i - 1) % n2;
        r = n2 - 1 - m;
    } else {
        r = i % n2;
    }
    return r >= n ? n2 - 1 - r : r;
}

__device__ __forceinline__ int clamp_idx(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// out dims = in dims with n[axis] -> n_out; out[.., o, ..] = sum_k w_k(coord[o]) * in[.., idx_k, ..]
__global__ void cubic_axis_kernel(const float* __restrict__ in, Dims3 d, int axis, const float* __restrict__ coord,
                                  int n_out, int bound /*1 replicate, 3 dct2*/, float* __restrict__ out) {
    Dims3 od = d;
    od.n[axis] = n_out;
    const int n = d.n[axis];
    const int64_t total = (int64_t)od.n[0] * od.n[1] * od.n[2];
    int64_t istride = 1;
    for (int a = 2; a > axis; --a) istride *= d.n[a];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int idx[3];
        idx[2] = (int)(i % od.n[2]);
        const int64_t t = i / od.n[2];
        idx[1] = (int)(t % od.n[1]);
        idx[0] = (int)(t / od.n[1]);
        const float g = coord[idx[axis]];
        const float g0f = floorf(g - 1.f);
        const float dist0 = g - g0f;
        const int g0 = (int)g0f;
        idx[axis] = 0;
        const int64_t base = ((int64_t)idx[0] * d.n[1] + idx[1]) * d.n[2] + idx[2];
        float acc = 0.f;
#pragma unroll
        for (int node = 0; node < 4; ++node) {
            const int j = bound == 3 ? reflect_dct2(g0 + node, n) : clamp_idx(g0 + node, n);
            acc = acc + in[base + (int64_t)j * istride] * bspline3(dist0 - (float)node);
        }
        out[i] = acc;
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256)); }

}  // namespace

extern "C" int bfm_bspline3_prefilter_axis(float* vol, int nx, int ny, int nz, int axis, int bound, float pole,
                                           float gain, const float* init_w, float pole_last, float init_scale,
                                           float final_scale, bfm_stream_t stream) {
    if (!vol || nx <= 0 || ny <= 0 || nz <= 0 || axis < 0 || axis > 2) return BFM_E_ARG;
    if (bound != 1 && bound != 3) return BFM_E_SHAPE;             // 'nearest' / 'dct2' share the DCT-II conditions (coeff.py:236-239)
    Dims3 d{{nx, ny, nz}};
    const int n = d.n[axis];
    if (n == 1) return BFM_OK;
    if (n > 2 && !init_w) return BFM_E_ARG;
    const int64_t lines = (int64_t)nx * ny * nz / n;
    const size_t tile_bytes = (size_t)64 * (n | 1) * sizeof(float);
    if (axis == 2 && tile_bytes <= 64 * 1024) {
        const int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(lines, 64));
        hipLaunchKernelGGL(prefilter_dct2_z_kernel, dim3(nb), dim3(64), tile_bytes, bfm_s(stream), vol, lines, n, pole, gain,
                           init_w, pole_last, init_scale, final_scale);
        return bfm_launch_status();
    }
    hipLaunchKernelGGL(prefilter_dct2_kernel, dim3(grid_for(lines)), dim3(256), 0, bfm_s(stream), vol, d, axis, pole, gain,
                       init_w, pole_last, init_scale, final_scale);
    return bfm_launch_status();
}

extern "C" int bfm_bspline3_resample_axis(const float* in, int nx, int ny, int nz, int axis, const float* coord,
                                          int n_out, int bound, float* out, bfm_stream_t stream) {
    if (!in || !coord || !out || nx <= 0 || ny <= 0 || nz <= 0 || axis < 0 || axis > 2 || n_out <= 0) return BFM_E_ARG;
    if (bound != 1 && bound != 3) return BFM_E_SHAPE;
    Dims3 d{{nx, ny, nz}};
    Dims3 od = d;
    od.n[axis] = n_out;
    const int64_t total = (int64_t)od.n[0] * od.n[1] * od.n[2];
    hipLaunchKernelGGL(cubic_axis_kernel, dim3(grid_for(total)), dim3(256), 0, bfm_s(stream), in, d, axis, coord, n_out, bound,
                       out);
    return bfm_launch_status();
}
