// HBM-bound helpers of the inference path: MaxPool3d(2) and tile stitching.
#include "bfm_common.h"

namespace {

__device__ __forceinline__ float nan_max(float m, float q) { return (m != m) ? m : ((q != q) ? q : fmaxf(m, q)); }

// nn.MaxPool3d(kernel_size=2): window 2, stride 2, floor, no padding
// (Trainer/models/unet3d/buildingblocks.py:185-186).  Channels-last, float4 per lane.
// Optional output-moment rows (see conv3d_mfma.hip / gn_stats.hip): one row per block, so that the GroupNorm of the
// next SingleConv does not re-read the pooled tensor.  Needs VEC == 4 and a grid whose thread count is a multiple of
// C/4 (then a thread keeps the same 4 channels across its grid-stride loop).
template <int VEC>
__global__ void maxpool2_kernel(const float* __restrict__ in, int C, int D, int H, int W, int d, int h, int w,
                                float* __restrict__ out, double* __restrict__ rsum, double* __restrict__ rsq,
                                float* __restrict__ rmn, float* __restrict__ rmx) {
    // batch (bfm_maxpool2_batch): blockIdx.y = sample; a sample's blocks do exactly what they do in a launch of it alone
    in += (int64_t)blockIdx.y * D * H * W * C;
    out += (int64_t)blockIdx.y * d * h * w * C;
    const size_t row0 = (size_t)blockIdx.y * gridDim.x;
    const int CV = C / VEC;
    float fs[VEC], fq[VEC], fmn[VEC], fmx[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { fs[k] = 0.f; fq[k] = 0.f; fmn[k] = INFINITY; fmx[k] = -INFINITY; }
    const int64_t n = (int64_t)d * h * w * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int cv = (int)(i % CV);
        int64_t v = i / CV;
        int x = (int)(v % w);
        int64_t t = v / w;
        int y = (int)(t % h);
        int z = (int)(t / h);
        float m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float* p = in + ((((int64_t)(2 * z + dz)) * H + (2 * y + dy)) * W + (2 * x + dx)) * C +
                                     (int64_t)cv * VEC;
                    if constexpr (VEC == 4) {
                        float4 q = *reinterpret_cast<const float4*>(p);
                        // torch's max propagates NaN from ANY corner; fmaxf does not -- sticky in both operands
                        m[0] = nan_max(m[0], q.x);
                        m[1] = nan_max(m[1], q.y);
                        m[2] = nan_max(m[2], q.z);
                        m[3] = nan_max(m[3], q.w);
                    } else {
                        float q = *p;
                        m[0] = nan_max(m[0], q);
                    }
                }
        float* o = out + v * C + (int64_t)cv * VEC;
        if constexpr (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(m[0], m[1], m[2], m[3]);
        else *o = m[0];
        if (rsum) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                fs[k] += m[k]; fq[k] = fmaf(m[k], m[k], fq[k]);
                fmn[k] = fminf(fmn[k], m[k]); fmx[k] = fmaxf(fmx[k], m[k]);
            }
        }
    }
    if (rsum) {                                                  // uniform; VEC == 4 guaranteed by the host
        extern __shared__ double pool_smem[];
        double* ls = pool_smem;                                  // [256][VEC]
        double* lq = ls + 256 * VEC;
        float* lmn = reinterpret_cast<float*>(lq + 256 * VEC);
        float* lmx = lmn + 256 * VEC;
        const int t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            ls[t * VEC + k] = (double)fs[k]; lq[t * VEC + k] = (double)fq[k];
            lmn[t * VEC + k] = fmn[k]; lmx[t * VEC + k] = fmx[k];
        }
        __syncthreads();
        const int per = 256 / CV;                                // threads of this block that share a channel group
        for (int c = t; c < C; c += 256) {
            const int cv = c / VEC, k = c - cv * VEC;
            // thread t' = cv0 + j*CV holds channel group cv0 = (first global thread id + t') % CV; the block's first
            // thread id is a multiple of 256, itself a multiple of CV, so cv0 == t' % CV
            double S = 0.0, Q = 0.0;
            float MN = INFINITY, MX = -INFINITY;
            for (int j = 0; j < per; ++j) {
                const int tt = cv + j * CV;
                S += ls[tt * VEC + k]; Q += lq[tt * VEC + k];
                MN = fminf(MN, lmn[tt * VEC + k]); MX = fmaxf(MX, lmx[tt * VEC + k]);
            }
            const size_t o = (row0 + blockIdx.x) * C + c;
            rsum[o] = S; rsq[o] = Q; rmn[o] = MN; rmx[o] = MX;
        }
    }
}

// scripts/demo_test.py:88-89,113-118: full[range] += tile * (tile_input != 0)
__global__ void stitch_kernel(const float* __restrict__ tile, const int64_t* __restrict__ tile_label,
                              const float* __restrict__ tin, int td, int th, int tw, float* __restrict__ full, int H,
                              int W, int z0, int y0, int x0) {
    const int64_t n = (int64_t)td * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(i % tw);
        int64_t t = i / tw;
        int y = (int)(t % th);
        int z = (int)(t / th);
        const bool m = tin == nullptr || tin[i] != 0.f;
        // v * mask with mask in {0, 1}, as a select: what lies outside the mask may never have been computed (the tile
        // loop's last layers skip it) and must not leak a NaN through 0 * NaN; +0 where the reference has +-0 -- both
        // vanish in `full (+0) += v`.  labels: (int64 * float mask) -> float, saved, re-read as int, summed as float
        float v = !m ? 0.f : (tile_label ? (float)(int)(float)tile_label[i] : tile[i]);
        full[((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x)] += v;
    }
}

// all stitched keys of one tile in a single launch: maps [n_maps][tvox] from the fused tail, sel[k] = map row of
// key k (-1: the int64 label), full [K][D*H*W]
__global__ void stitch_multi_kernel(const float* __restrict__ maps, int64_t map_stride, const int32_t* __restrict__ sel,
                                    int K, const int64_t* __restrict__ label, const float* __restrict__ tin, int td,
                                    int th, int tw, float* __restrict__ full, int D, int H, int W, int z0, int y0,
                                    int x0) {
    const int64_t n = (int64_t)td * th * tw;
    const int64_t vol = (int64_t)D * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(i % tw);
        int64_t t = i / tw;
        int y = (int)(t % th);
        int z = (int)(t / th);
        const int64_t o = ((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
        if (tin == nullptr) {                       // rows already masked and float typed (shipped by a peer rank)
            for (int k = 0; k < K; ++k) full[(int64_t)k * vol + o] += maps[(int64_t)sel[k] * map_stride + i];
            continue;
        }
        const bool m = tin[i] != 0.f;
        for (int k = 0; k < K; ++k) {
            const int r = sel[k];
            const float v = !m ? 0.f : (r < 0 ? (float)(int)(float)label[i] : maps[(int64_t)r * map_stride + i]);
            full[(int64_t)k * vol + o] += v;
        }
    }
}

// all K stitched keys of one tile, masked and float typed, packed [K][n]: what a rank ships to rank 0
__global__ void pack_multi_kernel(const float* __restrict__ maps, int64_t map_stride, const int32_t* __restrict__ sel,
                                  int K, const int64_t* __restrict__ label, const float* __restrict__ tin, int64_t n,
                                  float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool m = tin[i] != 0.f;                  // select, not multiply: see stitch_kernel
        for (int k = 0; k < K; ++k) {
            const int r = sel[k];
            out[(int64_t)k * n + i] = !m ? 0.f : (r < 0 ? (float)(int)(float)label[i] : maps[(int64_t)r * map_stride + i]);
        }
    }
}

__global__ void divide_multi_kernel(float* __restrict__ full, const float* __restrict__ cnt, int64_t vol, int K) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vol; i += (int64_t)gridDim.x * blockDim.x) {
        const float c = cnt[i];
        for (int k = 0; k < K; ++k) full[(int64_t)k * vol + i] = full[(int64_t)k * vol + i] / c;
    }
}

// tile output * (tile_input != 0), flattened (what a rank ships to rank 0)
__global__ void mask_kernel(const float* __restrict__ tile, const int64_t* __restrict__ tile_label,
                            const float* __restrict__ tin, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool m = tin[i] != 0.f;
        out[i] = !m ? 0.f : (tile_label ? (float)(int)(float)tile_label[i] : tile[i]);
    }
}

__global__ void count_add_kernel(float* __restrict__ cnt, int H, int W, int z0, int z1, int y0, int y1, int x0,
                                 int x1) {
    const int tw = x1 - x0, th = y1 - y0, td = z1 - z0;
    const int64_t n = (int64_t)td * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x = (int)(i % tw);
        int64_t t = i / tw;
        int y = (int)(t % th);
        int z = (int)(t / th);
        cnt[((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x)] += 1.f;
    }
}

__global__ void divide_kernel(float* __restrict__ full, const float* __restrict__ cnt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        full[i] = full[i] / cnt[i];
}

// The whole stitch of the multi-GPU path in one launch (scripts/demo_test.py:108-119): every tile's K masked rows are
// resident on rank 0 (packed [K][td*th*tw] by pack_multi_kernel, own or received), so each output voxel sums the tiles
// that cover it IN TABLE ORDER (the reference's tile order: 0 + t_a + t_b + ... is what the sequential `full[range] +=`
// computes), divides by their number (what the reference's cnt volume holds) and is written once -- no zero fill, no
// read-modify-write per tile, no separate divide pass.  tiles [T][8] int64: {row pointer, z0, y0, x0, td, th, tw, 0}.
// One block per (z, y) line: wave 0 compacts the tiles covering the line into LDS (ballot, order kept), then every
// thread walks that short list.
struct LineTile { const float* p; int64_t n, off; int x0, x1; };

__global__ void stitch_gather_kernel(const int64_t* __restrict__ tiles, int T, int K, float* __restrict__ full, int D,
                                     int H, int W) {
    extern __shared__ unsigned char smem_raw[];
    LineTile* cand = reinterpret_cast<LineTile*>(smem_raw);
    __shared__ int ncand_s, vec_s;
    const int64_t vol = (int64_t)D * H * W;
    const int lane = threadIdx.x & 63;
    const bool out_vec = !(W & 3) && !(reinterpret_cast<uintptr_t>(full) & 15);
    for (int line = blockIdx.x; line < D * H; line += gridDim.x) {
        const int z = line / H, y = line % H;
        __syncthreads();
        if (threadIdx.x < 64) {
            int n = 0, bad = 0;
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                bool ok = false, al = true;
                LineTile c{nullptr, 0, 0, 0, 0};
                if (t < T) {
                    const int64_t* d = tiles + (int64_t)t * 8;
                    const int z0 = (int)d[1], y0 = (int)d[2], x0 = (int)d[3];
                    const int td = (int)d[4], th = (int)d[5], tw = (int)d[6];
                    ok = z >= z0 && z < z0 + td && y >= y0 && y < y0 + th;
                    c.n = (int64_t)td * th * tw;
                    c.p = reinterpret_cast<const float*>(d[0]);
                    c.off = ((int64_t)(z - z0) * th + (y - y0)) * tw - x0;
                    c.x0 = x0;
                    c.x1 = x0 + tw;
                    al = !((x0 | tw) & 3) && !(c.n & 3) && !(d[0] & 15);   // whole 4-voxel groups in or out, 16-B rows
                }
                const unsigned long long m = __ballot(ok);
                if (ok) cand[n + __popcll(m & ((1ull << lane) - 1ull))] = c;
                n += __popcll(m);
                bad += __popcll(__ballot(ok && !al));
            }
            if (lane == 0) { ncand_s = n; vec_s = (bad == 0 && out_vec) ? 1 : 0; }
        }
        __syncthreads();
        const int nc = ncand_s;
        if (vec_s) {
            // items = (4-voxel group, key): a thread's items of one line share the group when W/4 divides the block
            const int nq = W >> 2;
            for (int idx = threadIdx.x; idx < nq * K; idx += blockDim.x) {
                const int x = (idx % nq) << 2, k = idx / nq;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                int cover = 0;
                for (int j = 0; j < nc; j += 4) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {           // loads first; an uncovered slot adds +0 (acc is never -0)
                        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j + u < nc && x >= cand[j + u].x0 && x < cand[j + u].x1) {
                            v[u] = *reinterpret_cast<const float4*>(cand[j + u].p + (int64_t)k * cand[j + u].n +
                                                                    cand[j + u].off + x);
                            ++cover;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                const float c = (float)cover;
                acc.x = acc.x / c; acc.y = acc.y / c; acc.z = acc.z / c; acc.w = acc.w / c;
                *reinterpret_cast<float4*>(full + (int64_t)k * vol + (int64_t)line * W + x) = acc;
            }
            continue;
        }
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            const int64_t o = (int64_t)line * W + x;
            int cover = 0;
            for (int j = 0; j < nc; ++j) cover += (x >= cand[j].x0 && x < cand[j].x1) ? 1 : 0;
            const float c = (float)cover;
            for (int k = 0; k < K; ++k) {
                float acc = 0.f;
                for (int j = 0; j < nc; ++j)
                    if (x >= cand[j].x0 && x < cand[j].x1) acc += cand[j].p[(int64_t)k * cand[j].n + cand[j].off + x];
                full[(int64_t)k * vol + o] = acc / c;
            }
        }
    }
}

inline int grid_for(int64_t n, int tpb = 256, int cap = 4096) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// rows of the output-moment table bfm_maxpool2_ex writes (0: this shape cannot emit them)
// ---------------------------------------------------------------------------------------------------------------
// The compact shipping form.  The tile loop keeps v * (tile input != 0) (scripts/demo_test.py:88-100) and the tile input is
// a window of the volume, so which voxels of a tile survive is known from the volume alone -- on every rank, before any
// tile is computed.  A tile's K rows then hold only its surviving voxels, in the tile's raster order: [K][row_stride]
// with voxel i at column pos[i] = number of non-zero input voxels before i in the tile.  On the bench volume (a head in a
// 256^3 box) that is 1/5 of the bytes a rank packs, ships over xGMI and rank 0 reads back.
// tile_mask_*: pos[] for every tile of the volume in three launches.  tiles [T][8] int64 = {offset of the tile's first
// entry in pos[], z0, y0, x0, td, th, tw, index of the tile's first block}; a block owns IDX_CHUNK consecutive voxels.
constexpr int IDX_CHUNK = 2048;                           // 256 threads x 8

__device__ __forceinline__ int idx_tile_of(const int64_t* __restrict__ tiles, int T, int b) {
    int lo = 0, hi = T - 1;                               // last tile whose first block is <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)tiles[(int64_t)mid * 8 + 7] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ bool idx_flag(const float* __restrict__ full, int H, int W, const int64_t* __restrict__ d,
                                         int64_t i, int64_t n) {
    if (i >= n) return false;
    const int th = (int)d[5], tw = (int)d[6];
    const int x = (int)(i % tw);
    const int64_t r = i / tw;
    const int y = (int)(r % th), z = (int)(r / th);
    return full[((int64_t)((int)d[1] + z) * H + ((int)d[2] + y)) * W + ((int)d[3] + x)] != 0.f;
}

__global__ void __launch_bounds__(256) tile_mask_count_kernel(const float* __restrict__ full, int H, int W,
                                                              const int64_t* __restrict__ tiles, int T,
                                                              int32_t* __restrict__ blockcount) {
    __shared__ int wsum[4];
    const int b = blockIdx.x, t = idx_tile_of(tiles, T, b);
    const int64_t* d = tiles + (int64_t)t * 8;
    const int64_t n = d[4] * d[5] * d[6], i0 = (int64_t)(b - (int)d[7]) * IDX_CHUNK;
    int c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) c += idx_flag(full, H, W, d, i0 + j * 256 + threadIdx.x, n) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockcount[b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one block per tile: exclusive scan of its blocks' counts (in place), the total to nnz[t]
__global__ void __launch_bounds__(256) tile_mask_scan_kernel(const int64_t* __restrict__ tiles, int T, int nblk_total,
                                                             int32_t* __restrict__ blockcount, int32_t* __restrict__ nnz) {
    __shared__ int sc[256];
    __shared__ int carry_s;
    const int t = blockIdx.x;
    const int b0 = (int)tiles[(int64_t)t * 8 + 7];
    const int b1 = t + 1 < T ? (int)tiles[(int64_t)(t + 1) * 8 + 7] : nblk_total;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = b0; base < b1; base += 256) {
        const int b = base + threadIdx.x;
        const int v = b < b1 ? blockcount[b] : 0;
        sc[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {                // Hillis-Steele inclusive scan
            const int a = threadIdx.x >= o ? sc[threadIdx.x - o] : 0;
            __syncthreads();
            sc[threadIdx.x] += a;
            __syncthreads();
        }
        const int carry = carry_s;
        if (b < b1) blockcount[b] = carry + sc[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + sc[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) nnz[t] = carry_s;
}

__global__ void __launch_bounds__(256) tile_mask_pos_kernel(const float* __restrict__ full, int H, int W,
                                                            const int64_t* __restrict__ tiles, int T,
                                                            const int32_t* __restrict__ blockoff, int32_t* __restrict__ pos) {
    __shared__ int wcnt[32];                              // [j][wave] counts, then their exclusive scan
    const int b = blockIdx.x, t = idx_tile_of(tiles, T, b);
    const int64_t* d = tiles + (int64_t)t * 8;
    const int64_t n = d[4] * d[5] * d[6], i0 = (int64_t)(b - (int)d[7]) * IDX_CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int pre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned long long m = __ballot(idx_flag(full, H, W, d, i0 + j * 256 + threadIdx.x, n));
        pre[j] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[j * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = blockoff[b];
        for (int q = 0; q < 32; ++q) { const int v = wcnt[q]; wcnt[q] = run; run += v; }
    }
    __syncthreads();
    int32_t* pt = pos + d[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t i = i0 + j * 256 + threadIdx.x;
        if (i < n) pt[i] = wcnt[j * 4 + wave] + pre[j];
    }
}

// the K keys of one tile where its input is non-zero: out[k * row_stride + pos[i]]
__global__ void pack_compact_kernel(const float* __restrict__ maps, int64_t map_stride, const int32_t* __restrict__ sel,
                                    int K, const int64_t* __restrict__ label, const float* __restrict__ tin, int64_t n,
                                    const int32_t* __restrict__ pos, int64_t row_stride, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (tin[i] == 0.f) continue;                       // NaN != 0: kept, as `im != 0` keeps it
        const int64_t p = pos[i];
        for (int k = 0; k < K; ++k) {
            const int r = sel[k];
            out[(int64_t)k * row_stride + p] = r < 0 ? (float)(int)(float)label[i] : maps[(int64_t)r * map_stride + i];
        }
    }
}

// stitch_gather_kernel on compact rows.  tiles [T][10] int64 = {rows, pos (the tile's), z0, y0, x0, td, th, tw, row_stride, 0}.
// A voxel whose volume input is zero is masked out of every tile: 0 + 0 + ... / cover = +0, written without reading.
struct LineTileC { const float* p; const int32_t* pos; int64_t rs, off; int x0, x1; };

__global__ void stitch_gather_compact_kernel(const int64_t* __restrict__ tiles, int T, int K,
                                             const float* __restrict__ vol_in, float* __restrict__ full, int D, int H, int W) {
    extern __shared__ unsigned char smem_raw[];
    LineTileC* cand = reinterpret_cast<LineTileC*>(smem_raw);
    __shared__ int ncand_s;
    const int64_t vol = (int64_t)D * H * W;
    const int lane = threadIdx.x & 63;
    for (int line = blockIdx.x; line < D * H; line += gridDim.x) {
        const int z = line / H, y = line % H;
        __syncthreads();
        if (threadIdx.x < 64) {
            int n = 0;
            for (int base = 0; base < T; base += 64) {
                const int t = base + lane;
                bool ok = false;
                LineTileC c{nullptr, nullptr, 0, 0, 0, 0};
                if (t < T) {
                    const int64_t* d = tiles + (int64_t)t * 10;
                    const int z0 = (int)d[2], y0 = (int)d[3], x0 = (int)d[4];
                    const int td = (int)d[5], th = (int)d[6], tw = (int)d[7];
                    ok = z >= z0 && z < z0 + td && y >= y0 && y < y0 + th;
                    c.p = reinterpret_cast<const float*>(d[0]);
                    c.pos = reinterpret_cast<const int32_t*>(d[1]);
                    c.rs = d[8];
                    c.off = ((int64_t)(z - z0) * th + (y - y0)) * tw - x0;
                    c.x0 = x0;
                    c.x1 = x0 + tw;
                }
                const unsigned long long m = __ballot(ok);
                if (ok) cand[n + __popcll(m & ((1ull << lane) - 1ull))] = c;
                n += __popcll(m);
            }
            if (lane == 0) ncand_s = n;
        }
        __syncthreads();
        const int nc = ncand_s;
        // background first: most of a head volume is zero input; whole groups of 4 such voxels are written as float4
        // zeros, one (group, key) item per thread
        const bool vec = !(W & 3) && !((reinterpret_cast<uintptr_t>(full) | reinterpret_cast<uintptr_t>(vol_in)) & 15);
        if (vec) {
            const int nq = W >> 2;
            const float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int it = threadIdx.x; it < nq * K; it += blockDim.x) {
                const int g = it % nq, k = it / nq;
                const float4 iv = *reinterpret_cast<const float4*>(vol_in + (int64_t)line * W + 4 * g);
                if (iv.x == 0.f && iv.y == 0.f && iv.z == 0.f && iv.w == 0.f)
                    *reinterpret_cast<float4*>(full + (int64_t)k * vol + (int64_t)line * W + 4 * g) = zz;
            }
        }
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            const int64_t o = (int64_t)line * W + x;
            if (vol_in[o] == 0.f) {
                bool done = false;
                if (vec) {
                    const float4 iv = *reinterpret_cast<const float4*>(vol_in + (o & ~(int64_t)3));
                    done = iv.x == 0.f && iv.y == 0.f && iv.z == 0.f && iv.w == 0.f;
                }
                if (!done)
                    for (int k = 0; k < K; ++k) full[(int64_t)k * vol + o] = 0.f;
                continue;
            }
            // the covering tiles' columns, in table order; a line of the reference tilings has at most 12 candidate tiles,
            // kept in registers through fully unrolled loops (no dynamically indexed private arrays)
            constexpr int NCMAX = 12;
            if (nc <= NCMAX) {
                const float* colp[NCMAX];
                int64_t crs[NCMAX];
                int cover = 0;
#pragma unroll
                for (int j = 0; j < NCMAX; ++j) {
                    const bool in = j < nc && x >= cand[j].x0 && x < cand[j].x1;
                    colp[j] = in ? cand[j].p + cand[j].pos[cand[j].off + x] : nullptr;
                    crs[j] = in ? cand[j].rs : 0;
                    cover += in ? 1 : 0;
                }
                const float c = (float)cover;
                for (int k = 0; k < K; ++k) {
                    float v[NCMAX];
#pragma unroll
                    for (int j = 0; j < NCMAX; ++j) v[j] = colp[j] ? colp[j][(int64_t)k * crs[j]] : 0.f;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NCMAX; ++j) acc += v[j];          // an absent tile adds +0: acc is never -0
                    full[(int64_t)k * vol + o] = acc / c;
                }
            } else {
                int cover = 0;
                for (int j = 0; j < nc; ++j) cover += (x >= cand[j].x0 && x < cand[j].x1) ? 1 : 0;
                const float c = (float)cover;
                for (int k = 0; k < K; ++k) {
                    float acc = 0.f;
                    for (int j = 0; j < nc; ++j)
                        if (x >= cand[j].x0 && x < cand[j].x1)
                            acc += cand[j].p[(int64_t)k * cand[j].rs + cand[j].pos[cand[j].off + x]];
                    full[(int64_t)k * vol + o] = acc / c;
                }
            }
        }
    }
}

extern "C" int bfm_maxpool2_rows(int C, int D, int H, int W) {
    if (C <= 0 || C % 4 || D < 2 || H < 2 || W < 2) return 0;
    const int CV = C / 4;
    if (CV > 256 || 256 % CV) return 0;
    return grid_for((int64_t)(D / 2) * (H / 2) * (W / 2) * CV);
}

extern "C" int bfm_maxpool2_ex(const float* in, int C, int D, int H, int W, float* out, void* moment_rows,
                               bfm_stream_t stream) {
    if (!in || !out || C <= 0 || D < 2 || H < 2 || W < 2) return BFM_E_ARG;
    int d = D / 2, h = H / 2, w = W / 2;
    bool v4 = (C % 4 == 0) && !((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15);
    int64_t n = (int64_t)d * h * w * (v4 ? C / 4 : C);
    const int nb = grid_for(n);
    double *rsum = nullptr, *rsq = nullptr;
    float *rmn = nullptr, *rmx = nullptr;
    size_t smem = 0;
    if (moment_rows) {
        if (!v4 || bfm_maxpool2_rows(C, D, H, W) != nb) return BFM_E_SHAPE;
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t k = (size_t)nb * C;
        rsum = reinterpret_cast<double*>(rb);
        rsq = reinterpret_cast<double*>(rb + k * 8);
        rmn = reinterpret_cast<float*>(rb + k * 16);
        rmx = reinterpret_cast<float*>(rb + k * 20);
        smem = (size_t)256 * 4 * 24;
    }
    if (v4) hipLaunchKernelGGL(maxpool2_kernel<4>, dim3(nb), dim3(256), smem, bfm_s(stream), in, C, D, H, W, d, h, w, out,
                               rsum, rsq, rmn, rmx);
    else hipLaunchKernelGGL(maxpool2_kernel<1>, dim3(nb), dim3(256), 0, bfm_s(stream), in, C, D, H, W, d, h, w, out, rsum,
                            rsq, rmn, rmx);
    return bfm_launch_status();
}

// S same-shape samples (S,D,H,W,C) -> (S,D/2,H/2,W/2,C) in one launch (the batched levels, engine.deep_region); moment rows
// [S * bfm_maxpool2_batch_rows()][C], at most 128 per sample so that bfm_gn_stats_rows_batch reads them directly.  A sample's
// output and rows do not depend on S.
extern "C" int bfm_maxpool2_batch_rows(int C, int D, int H, int W) {
    const int n = bfm_maxpool2_rows(C, D, H, W);
    return n > 128 ? 128 : n;
}

extern "C" int bfm_maxpool2_batch(const float* in, int C, int S, int D, int H, int W, float* out, void* moment_rows,
                                  bfm_stream_t stream) {
    if (!in || !out || C <= 0 || S < 1 || S > 65535 || D < 2 || H < 2 || W < 2) return BFM_E_ARG;
    if ((C % 4) || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15)) return BFM_E_SHAPE;
    const int d = D / 2, h = H / 2, w = W / 2;
    const int nb = std::min(128, grid_for((int64_t)d * h * w * (C / 4)));
    double *rsum = nullptr, *rsq = nullptr;
    float *rmn = nullptr, *rmx = nullptr;
    size_t smem = 0;
    if (moment_rows) {
        if (bfm_maxpool2_batch_rows(C, D, H, W) != nb) return BFM_E_SHAPE;
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t k = (size_t)S * nb * C;
        rsum = reinterpret_cast<double*>(rb);
        rsq = reinterpret_cast<double*>(rb + k * 8);
        rmn = reinterpret_cast<float*>(rb + k * 16);
        rmx = reinterpret_cast<float*>(rb + k * 20);
        smem = (size_t)256 * 4 * 24;
    }
    hipLaunchKernelGGL(maxpool2_kernel<4>, dim3(nb, S), dim3(256), smem, bfm_s(stream), in, C, D, H, W, d, h, w, out, rsum,
                       rsq, rmn, rmx);
    return bfm_launch_status();
}

extern "C" int bfm_maxpool2(const float* in, int C, int D, int H, int W, float* out, bfm_stream_t stream) {
    return bfm_maxpool2_ex(in, C, D, H, W, out, nullptr, stream);
}

extern "C" int bfm_stitch_accumulate(const float* tile, const int64_t* tile_label, const float* tile_input, int td,
                                     int th, int tw, float* full, int D, int H, int W, int z0, int y0, int x0,
                                     bfm_stream_t stream) {
    if ((!tile && !tile_label) || !full || td <= 0 || th <= 0 || tw <= 0) return BFM_E_ARG;
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + td > D || y0 + th > H || x0 + tw > W) return BFM_E_SHAPE;
    int64_t n = (int64_t)td * th * tw;
    hipLaunchKernelGGL(stitch_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), tile, tile_label, tile_input, td,
                       th, tw, full, H, W, z0, y0, x0);
    return bfm_launch_status();
}

extern "C" int bfm_stitch_accumulate_multi(const float* maps, int64_t map_stride, const int32_t* sel, int K,
                                           const int64_t* tile_label, const float* tile_input, int td, int th, int tw,
                                           float* full, int D, int H, int W, int z0, int y0, int x0,
                                           bfm_stream_t stream) {
    if (!maps || !sel || K <= 0 || !full || td <= 0 || th <= 0 || tw <= 0) return BFM_E_ARG;
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + td > D || y0 + th > H || x0 + tw > W) return BFM_E_SHAPE;
    int64_t n = (int64_t)td * th * tw;
    hipLaunchKernelGGL(stitch_multi_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), maps, map_stride, sel, K,
                       tile_label, tile_input, td, th, tw, full, D, H, W, z0, y0, x0);
    return bfm_launch_status();
}

extern "C" int bfm_mask_tile(const float* tile, const int64_t* tile_label, const float* tile_input, int64_t n,
                             float* out, bfm_stream_t stream) {
    if ((!tile && !tile_label) || !tile_input || !out || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(mask_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), tile, tile_label, tile_input, n, out);
    return bfm_launch_status();
}

extern "C" int bfm_tile_count_add(float* cnt, int D, int H, int W, int z0, int z1, int y0, int y1, int x0, int x1,
                                  bfm_stream_t stream) {
    if (!cnt) return BFM_E_ARG;
    if (z0 < 0 || y0 < 0 || x0 < 0 || z1 > D || y1 > H || x1 > W || z1 <= z0 || y1 <= y0 || x1 <= x0)
        return BFM_E_SHAPE;
    int64_t n = (int64_t)(z1 - z0) * (y1 - y0) * (x1 - x0);
    hipLaunchKernelGGL(count_add_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), cnt, H, W, z0, z1, y0, y1, x0,
                       x1);
    return bfm_launch_status();
}

extern "C" int bfm_tile_mask_blocks(int64_t tile_voxels) {
    return tile_voxels > 0 ? (int)((tile_voxels + IDX_CHUNK - 1) / IDX_CHUNK) : 0;
}

extern "C" int bfm_tile_mask_index(const float* volume, int D, int H, int W, const int64_t* tiles, int T, int total_blocks,
                                   int32_t* pos, int32_t* nnz, int32_t* block_ws, bfm_stream_t stream) {
    if (!volume || !tiles || !pos || !nnz || !block_ws || D <= 0 || H <= 0 || W <= 0 || T <= 0 || total_blocks <= 0)
        return BFM_E_ARG;
    hipLaunchKernelGGL(tile_mask_count_kernel, dim3(total_blocks), dim3(256), 0, bfm_s(stream), volume, H, W, tiles, T,
                       block_ws);
    hipLaunchKernelGGL(tile_mask_scan_kernel, dim3(T), dim3(256), 0, bfm_s(stream), tiles, T, total_blocks, block_ws, nnz);
    hipLaunchKernelGGL(tile_mask_pos_kernel, dim3(total_blocks), dim3(256), 0, bfm_s(stream), volume, H, W, tiles, T,
                       block_ws, pos);
    return bfm_launch_status();
}

extern "C" int bfm_pack_tile_compact(const float* maps, int64_t map_stride, const int32_t* sel, int K,
                                     const int64_t* tile_label, const float* tile_input, int64_t n, const int32_t* pos,
                                     int64_t row_stride, float* out, bfm_stream_t stream) {
    if (!maps || !sel || K <= 0 || !tile_input || !pos || !out || n <= 0 || row_stride < 0) return BFM_E_ARG;
    hipLaunchKernelGGL(pack_compact_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), maps, map_stride, sel, K,
                       tile_label, tile_input, n, pos, row_stride, out);
    return bfm_launch_status();
}

extern "C" int bfm_stitch_gather_compact(const int64_t* tiles, int T, int K, const float* volume, float* full, int D, int H,
                                         int W, bfm_stream_t stream) {
    if (!tiles || !volume || !full || T <= 0 || K <= 0 || D <= 0 || H <= 0 || W <= 0) return BFM_E_ARG;
    if (T > 1024) return BFM_E_SHAPE;                          // candidate list lives in LDS (48 B per tile)
    const int64_t lines = (int64_t)D * H;
    const int nb = (int)(lines > 65536 ? 65536 : lines);
    hipLaunchKernelGGL(stitch_gather_compact_kernel, dim3(nb), dim3(256), (size_t)T * sizeof(LineTileC), bfm_s(stream), tiles,
                       T, K, volume, full, D, H, W);
    return bfm_launch_status();
}

extern "C" int bfm_divide_by_count(float* full, const float* cnt, int64_t n, bfm_stream_t stream) {
    if (!full || !cnt || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(divide_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), full, cnt, n);
    return bfm_launch_status();
}

extern "C" int bfm_pack_tile_multi(const float* maps, int64_t map_stride, const int32_t* sel, int K,
                                   const int64_t* tile_label, const float* tile_input, int64_t n, float* out,
                                   bfm_stream_t stream) {
    if (!maps || !sel || K <= 0 || !tile_input || !out || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(pack_multi_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), maps, map_stride, sel, K,
                       tile_label, tile_input, n, out);
    return bfm_launch_status();
}

extern "C" int bfm_divide_by_count_multi(float* full, const float* cnt, int64_t vol, int K, bfm_stream_t stream) {
    if (!full || !cnt || vol <= 0 || K <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(divide_multi_kernel, dim3(grid_for(vol)), dim3(256), 0, bfm_s(stream), full, cnt, vol, K);
    return bfm_launch_status();
}

extern "C" int bfm_stitch_gather_multi(const int64_t* tiles, int T, int K, float* full, int D, int H, int W,
                                       bfm_stream_t stream) {
    if (!tiles || !full || T <= 0 || K <= 0 || D <= 0 || H <= 0 || W <= 0) return BFM_E_ARG;
    if (T > 2048) return BFM_E_SHAPE;                          // candidate list lives in LDS (32 B per tile)
    const int64_t lines = (int64_t)D * H;
    const int nb = (int)(lines > 65536 ? 65536 : lines);
    hipLaunchKernelGGL(stitch_gather_kernel, dim3(nb), dim3(256), (size_t)T * sizeof(LineTile), bfm_s(stream), tiles, T, K,
                       full, D, H, W);
    return bfm_launch_status();
}
