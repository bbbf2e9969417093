// Small device kernels of the pre-processing chain around the inference path (utils/test_utils.py:235-284
// prepare_image): axis permutation + flips of align_volume_to_ref (utils/misc.py:1207-1247), the
// bounding box of zero_crop (utils/test_utils.py:60-72) and the channel mean of multi-frame inputs.  All HBM-bound.
#include "bfm_common.h"
#include <climits>

namespace {

struct PF {
    int n[3];        // input dims
    int o[3];        // output dims
    int perm[3];
    int flip[3];
};

__global__ void permute_flip_kernel(const float* __restrict__ in, PF p, float* __restrict__ out) {
    const int64_t total = (int64_t)p.o[0] * p.o[1] * p.o[2];
    const int64_t sx = (int64_t)p.n[1] * p.n[2], sy = p.n[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int idx[3];
        idx[2] = (int)(i % p.o[2]);
        const int64_t t = i / p.o[2];
        idx[1] = (int)(t % p.o[1]);
        idx[0] = (int)(t / p.o[1]);
        int j[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a) j[p.perm[a]] = p.flip[a] ? p.o[a] - 1 - idx[a] : idx[a];
        out[i] = in[j[0] * sx + j[1] * sy + j[2]];
    }
}

__global__ void bbox_init_kernel(int32_t* box) {
    if (threadIdx.x < 3) box[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) box[threadIdx.x] = 0;
}

// integer min/max atomics: the result does not depend on the order of arrival
__global__ void bbox_kernel(const float* __restrict__ in, int nx, int ny, int nz, float tol, int32_t* box) {
    const int64_t total = (int64_t)nx * ny * nz;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (in[i] > tol) {
            const int z = (int)(i % nz);
            const int64_t t = i / nz;
            const int y = (int)(t % ny);
            const int x = (int)(t / ny);
            lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
            hi[0] = max(hi[0], x + 1); hi[1] = max(hi[1], y + 1); hi[2] = max(hi[2], z + 1);
        }
    }
    // wave fold, then block fold through LDS: one atomic pair per block and axis instead of one per wave (65k waves
    // hammering six addresses serialise)
    __shared__ int s_lo[3][4], s_hi[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int l = lo[a], h = hi[a];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            l = min(l, __shfl_xor(l, m));
            h = max(h, __shfl_xor(h, m));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[a][threadIdx.x >> 6] = l; s_hi[a][threadIdx.x >> 6] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        int l = INT_MAX, h = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { l = min(l, s_lo[a][w]); h = max(h, s_hi[a][w]); }
        if (l != INT_MAX) atomicMin(&box[a], l);
        if (h != 0) atomicMax(&box[3 + a], h);
    }
}

__global__ void mean_lastdim_kernel(const float* __restrict__ in, int64_t n, int c, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int j = 0; j < c; ++j) s += in[i * c + j];
        out[i] = s / (float)c;
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256)); }

// a tile's window of the volume, copied into the contiguous input of the tile's graph (one wave per x-run)
__global__ void __launch_bounds__(256) crop3d_kernel(const float* __restrict__ vol, int H, int W, int z0, int y0, int x0,
                                                     int d, int h, int w, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)d * h;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int z = (int)(r / h), y = (int)(r - (int64_t)z * h);
        const float* src = vol + ((int64_t)(z0 + z) * H + (y0 + y)) * W + x0;
        float* dst = out + r * w;
        for (int x = lane; x < w; x += 64) dst[x] = src[x];
    }
}

}  // namespace

extern "C" int bfm_crop3d(const float* vol, int D, int H, int W, int z0, int y0, int x0, int d, int h, int w, float* out,
                          bfm_stream_t stream) {
    if (!vol || !out || D <= 0 || H <= 0 || W <= 0 || d <= 0 || h <= 0 || w <= 0) return BFM_E_ARG;
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + d > D || y0 + h > H || x0 + w > W) return BFM_E_SHAPE;
    const int64_t rows = (int64_t)d * h;
    hipLaunchKernelGGL(crop3d_kernel, dim3((unsigned)std::min<int64_t>(4096, bfm_cdiv64(rows, 4))), dim3(256), 0, bfm_s(stream),
                       vol, H, W, z0, y0, x0, d, h, w, out);
    return bfm_launch_status();
}

extern "C" int bfm_permute_flip3d(const float* in, int nx, int ny, int nz, const int* perm, const int* flip, float* out,
                                  bfm_stream_t stream) {
    if (!in || !out || !perm || !flip || nx <= 0 || ny <= 0 || nz <= 0) return BFM_E_ARG;
    int seen[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) {
        if (perm[a] < 0 || perm[a] > 2 || seen[perm[a]]) return BFM_E_ARG;
        seen[perm[a]] = 1;
    }
    PF p;
    p.n[0] = nx; p.n[1] = ny; p.n[2] = nz;
    for (int a = 0; a < 3; ++a) { p.perm[a] = perm[a]; p.flip[a] = flip[a] ? 1 : 0; p.o[a] = p.n[perm[a]]; }
    const int64_t n = (int64_t)nx * ny * nz;
    hipLaunchKernelGGL(permute_flip_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), in, p, out);
    return bfm_launch_status();
}

extern "C" int bfm_bbox_nonzero(const float* in, int nx, int ny, int nz, float tol, int32_t* box, bfm_stream_t stream) {
    if (!in || !box || nx <= 0 || ny <= 0 || nz <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, bfm_s(stream), box);
    const int64_t n = (int64_t)nx * ny * nz;
    const int nb = std::min(grid_for(n), 512);       // same-address atomics serialise: few blocks, long grid-stride loops
    hipLaunchKernelGGL(bbox_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), in, nx, ny, nz, tol, box);
    return bfm_launch_status();
}

extern "C" int bfm_mean_lastdim(const float* in, int64_t n, int c, float* out, bfm_stream_t stream) {
    if (!in || !out || n <= 0 || c <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(mean_lastdim_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), in, n, c, out);
    return bfm_launch_status();
}
