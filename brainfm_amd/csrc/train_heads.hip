// Training-side kernels around the backbone: the losses of Trainer/models/criterion.py with their
// gradients w.r.t. the raw head outputs, the backward of the 1x1x1 task heads + F.normalize, and the optimiser step.
//
//   losses (criterion.py:111-124, 178-186, 215-273, 282-294; losses.py:10-11, 27-74):
//     l1        mean(|o - t| * w)                                   T1/T2/FLAIR/CT/SR/distance/registration/bias_field_log
//     grad_l1   mean|d_x o - d_x t| + mean|d_y ..| + mean|d_z ..|    forward differences, last slice zero (GradientLoss)
//     seg       CE = mean_v( -sum_c log(max(p,1e-5)) * w_c * t_c ),  Dice = sum_c w_c (1 - 2 sum_v p t / max(sum_v (p+t), 1e-5))
//               with p = softmax(logits)  (SegProcessor, joiner.py:69-77)
//   Every loss kernel ADDS  coef * dL/d(raw)  into its columns of dRaw ([nvox][n_out], channels-last) and returns the
//   loss value in fp64; reductions are two-stage with a fixed order.
//   heads:   raw = Fn W^T + b,  Fn = F.normalize(feat)  (head.py:52-59, model.py:207)
//   AdamW:   torch.optim.AdamW semantics (decoupled weight decay), one fused elementwise kernel.
#include "bfm_common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int RB = 1024;                         // partial blocks of the reductions

__device__ __forceinline__ double block_sum(double v, double* red) {        // 256 threads, fixed tree
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ void final_sum_kernel(const double* __restrict__ part, int nb, double scale, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// ----------------------------------------------------------------------------- L1
// out column `co` of raw vs target [nvox] (optional weight [nvox]); clampv > 0: the prediction is clamp(o, +-clampv)
// first (DistProcessor), whose gradient is zero where it clamps
__global__ void __launch_bounds__(256) l1_kernel(const float* __restrict__ raw, int n_out, int co,
                                                 const float* __restrict__ target, const float* __restrict__ weight,
                                                 const float* __restrict__ mask_mul, int64_t nvox, float clampv,
                                                 int l2, float coef, float* __restrict__ dRaw,
                                                 double* __restrict__ part) {
    __shared__ double red[256];
    double acc = 0.0;
    GRID_STRIDE(v, nvox) {
        float o = raw[v * n_out + co];
        bool live = true;
        if (clampv > 0.f) {
            if (o > clampv) { o = clampv; live = false; }
            else if (o < -clampv) { o = -clampv; live = false; }
        }
        const float m = mask_mul ? mask_mul[v] : 1.f;        // bias field: both sides are multiplied by the soft mask
        const float w = weight ? weight[v] : 1.f;
        const float d = o * m - target[v] * m;
        acc += (double)((l2 ? d * d : fabsf(d)) * w);
        const float sg = l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        if (live && dRaw) dRaw[v * n_out + co] += coef * sg * w * m;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// All l1 / l2 entries of one sample in ONE pass over the head outputs: a block stages rows of raw (and of dRaw) for a
// run of voxels through LDS with coalesced row segments, each thread handles one voxel and walks the entry table
// (column, target, weight, mask, clamp, l2, coefficient); per-entry sums are folded per block in fp64 in a fixed order.
// The single-entry kernel above re-reads a 276-byte-strided column of the whole tensor per launch (22 launches, 3.4 ms at
// 128^3).
constexpr int L1M_MAX = 32;
struct L1Entry {
    int col;
    int l2;
    float clampv;
    float coef;                       // already divided by nvox
    const float* target;
    const float* weight;
    const float* mask;
};
struct L1Table {
    L1Entry e[L1M_MAX];
    int n;
};

__global__ void __launch_bounds__(256) l1_multi_kernel(const float* __restrict__ raw, int n_out, int64_t nvox, int tile_vox,
                                                       const L1Table tab, float* __restrict__ dRaw,
                                                       double* __restrict__ part /*[nb][L1M_MAX]*/) {
    extern __shared__ float tile[];                           // [tile_vox][ld] raw, then the same for the gradient
    __shared__ double red[256];
    const int ld = n_out | 1;
    float* traw = tile;
    float* tgrd = tile + (size_t)tile_vox * ld;
    const int t = threadIdx.x;
    double acc[L1M_MAX];
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) acc[k] = 0.0;
    const int64_t ntile = bfm_cdiv64(nvox, tile_vox);
    for (int64_t tb = blockIdx.x; tb < ntile; tb += gridDim.x) {
        const int64_t vb = tb * tile_vox;
        const int nv = (int)min<int64_t>(tile_vox, nvox - vb);
        __syncthreads();
        for (int i = t; i < nv * n_out; i += 256) {
            const int vl = i / n_out, c = i - vl * n_out;
            traw[vl * ld + c] = raw[vb * n_out + i];
            tgrd[vl * ld + c] = 0.f;
        }
        __syncthreads();
        if (t < nv) {
            const int64_t v = vb + t;
#pragma unroll
            for (int k = 0; k < L1M_MAX; ++k) {
                if (k >= tab.n) continue;                          // (a break keeps the loop rolled and acc[] in scratch)
                const L1Entry& e = tab.e[k];
                float o = traw[t * ld + e.col];
                bool live = true;
                if (e.clampv > 0.f) {
                    if (o > e.clampv) { o = e.clampv; live = false; }
                    else if (o < -e.clampv) { o = -e.clampv; live = false; }
                }
                const float m = e.mask ? e.mask[v] : 1.f;
                const float w = e.weight ? e.weight[v] : 1.f;
                const float d = o * m - e.target[v] * m;
                acc[k] += (double)((e.l2 ? d * d : fabsf(d)) * w);
                const float sg = e.l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                if (live) tgrd[t * ld + e.col] += e.coef * sg * w * m;
            }
        }
        __syncthreads();
        if (dRaw)
            for (int i = t; i < nv * n_out; i += 256) {
                const int vl = i / n_out, c = i - vl * n_out;
                const float g = tgrd[vl * ld + c];
                if (g != 0.f) dRaw[vb * n_out + i] += g;
            }
    }
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) {
        if (k >= tab.n) continue;                                  // wave-uniform
        const double sk = block_sum(acc[k], red);
        if (t == 0) part[(int64_t)blockIdx.x * L1M_MAX + k] = sk;
    }
}

// one wave per quantity: lane l adds its strided partials in order, then a fixed xor tree (deterministic; one thread
// walking 1 024 partials of every quantity took 350 us per call)
__global__ void l1_multi_fold_kernel(const double* __restrict__ part, int nb, int n, double scale, double* __restrict__ out) {
    const int k = blockIdx.x, l = threadIdx.x;
    if (k >= n) return;
    double s_ = 0.0;
    for (int b = l; b < nb; b += 64) s_ += part[(int64_t)b * L1M_MAX + k];
    s_ = wave_reduce_sum(s_);
    if (l == 0) out[k] = s_ * scale;
}

// ----------------------------------------------------------------------------- gradient L1 (GradientLoss 'l1')
__global__ void __launch_bounds__(256) grad_l1_kernel(const float* __restrict__ raw, int n_out, int co,
                                                      const float* __restrict__ target, const float* __restrict__ weight,
                                                      int D, int H, int W, float coef, float* __restrict__ dRaw,
                                                      double* __restrict__ part) {
    __shared__ double red[256];
    const int64_t nvox = (int64_t)D * H * W;
    double acc = 0.0;
    GRID_STRIDE(v, nvox) {
        const int x = (int)(v % W);
        const int64_t t2 = v / W;
        const int y = (int)(t2 % H), z = (int)(t2 / H);
        const float o = raw[v * n_out + co], t = target[v];
        const float w = weight ? weight[v] : 1.f;
        float g = 0.f;                                        // d(sum of the three terms)/d(o at v), own contributions
        // forward differences along the fastest (x), middle (y) and slowest (z) axis; zero on the last slice
        if (x + 1 < W) {
            const float d = (raw[(v + 1) * n_out + co] - o) - (target[v + 1] - t);
            acc += (double)(fabsf(d) * w);
            g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
        }
        if (y + 1 < H) {
            const float d = (raw[(v + W) * n_out + co] - o) - (target[v + W] - t);
            acc += (double)(fabsf(d) * w);
            g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
        }
        if (z + 1 < D) {
            const int64_t s = (int64_t)H * W;
            const float d = (raw[(v + s) * n_out + co] - o) - (target[v + s] - t);
            acc += (double)(fabsf(d) * w);
            g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
        }
        // contributions of the differences anchored at the previous voxel along each axis (weight of THAT voxel)
        if (x > 0) {
            const float d = (o - raw[(v - 1) * n_out + co]) - (t - target[v - 1]);
            g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - 1] : 1.f);
        }
        if (y > 0) {
            const float d = (o - raw[(v - W) * n_out + co]) - (t - target[v - W]);
            g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - W] : 1.f);
        }
        if (z > 0) {
            const int64_t s = (int64_t)H * W;
            const float d = (o - raw[(v - s) * n_out + co]) - (t - target[v - s]);
            g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - s] : 1.f);
        }
        if (dRaw) dRaw[v * n_out + co] += coef * g;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// Every gradient-L1 entry of a sample in one launch (round 4): the seven rows of raw a voxel's stencil touches are fetched
// once for all entries instead of once per entry and launch (8 launches x 218 us at 128^3: each read one 4-byte column
// out of 276-byte rows).  Same expressions per entry as grad_l1_kernel.
struct GL1Entry { int col; float coef; const float* target; const float* weight; };
struct GL1Table { GL1Entry e[L1M_MAX]; int n; };

__global__ void __launch_bounds__(256) grad_l1_multi_kernel(const float* __restrict__ raw, int n_out, const GL1Table tab, int D,
                                                            int H, int W, float* __restrict__ dRaw,
                                                            double* __restrict__ part /*[nb][L1M_MAX]*/) {
    __shared__ double red[256];
    const int64_t nvox = (int64_t)D * H * W;
    const int64_t s = (int64_t)H * W;
    double acc[L1M_MAX];
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) acc[k] = 0.0;
    GRID_STRIDE(v, nvox) {
        const int x = (int)(v % W);
        const int64_t t2 = v / W;
        const int y = (int)(t2 % H), z = (int)(t2 / H);
        const float* r0 = raw + v * n_out;
#pragma unroll
        for (int k = 0; k < L1M_MAX; ++k) {
            if (k >= tab.n) continue;
            const GL1Entry& e = tab.e[k];
            const float* target = e.target;
            const float* weight = e.weight;
            const int co = e.col;
            const float o = r0[co], t = target[v];
            const float w = weight ? weight[v] : 1.f;
            float g = 0.f;
            double a = 0.0;
            if (x + 1 < W) {
                const float d = (r0[n_out + co] - o) - (target[v + 1] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (y + 1 < H) {
                const float d = (r0[(int64_t)W * n_out + co] - o) - (target[v + W] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (z + 1 < D) {
                const float d = (r0[s * n_out + co] - o) - (target[v + s] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (x > 0) {
                const float d = (o - r0[-(int64_t)n_out + co]) - (t - target[v - 1]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - 1] : 1.f);
            }
            if (y > 0) {
                const float d = (o - r0[-(int64_t)W * n_out + co]) - (t - target[v - W]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - W] : 1.f);
            }
            if (z > 0) {
                const float d = (o - r0[-s * n_out + co]) - (t - target[v - s]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - s] : 1.f);
            }
            acc[k] += a;
            if (dRaw) dRaw[v * n_out + co] += e.coef * g;
        }
    }
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) {
        if (k >= tab.n) continue;                                  // wave-uniform
        const double sk = block_sum(acc[k], red);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * L1M_MAX + k] = sk;
    }
}

// ----------------------------------------------------------------------------- segmentation: CE + Dice on softmax
// pass 1: probabilities P [nvox][ns] (written out) and per-block partials of CE      part: [nb][1 + 2*ns], slot 0
__global__ void __launch_bounds__(256) seg_fwd_kernel(const float* __restrict__ raw, int n_out, int c0, int ns,
                                                      const float* __restrict__ target /*[ns][nvox] one-hot / soft (NCDHW)*/,
                                                      const float* __restrict__ wce /*[ns]*/, int64_t nvox,
                                                      float* __restrict__ P, double* __restrict__ part) {
    __shared__ double red[256];
    double ce = 0.0;
    GRID_STRIDE(v, nvox) {
        const float* r = raw + v * n_out + c0;
        float m = -INFINITY;
        for (int c = 0; c < ns; ++c) m = fmaxf(m, r[c]);
        float sum = 0.f;
        for (int c = 0; c < ns; ++c) sum += expf(r[c] - m);
        const float inv = 1.f / sum;
        for (int c = 0; c < ns; ++c) {
            const float p = expf(r[c] - m) * inv;
            P[v * ns + c] = p;
            ce -= (double)(logf(fmaxf(p, 1e-5f)) * wce[c] * target[(int64_t)c * nvox + v]);
        }
    }
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * (1 + 2 * ns)] = ce;
}

// pass 1b: per-class sums  sum_v p*t  and  sum_v (p+t)  of one voxel chunk; thread = (voxel lane j, class c), fixed order
__global__ void __launch_bounds__(256) seg_class_sums_kernel(const float* __restrict__ P, const float* __restrict__ target,
                                                             int ns, int64_t nvox, int64_t vox_per_block,
                                                             double* __restrict__ part) {
    __shared__ double s_pt[256], s_ps[256];
    const int nj = 256 / ns;
    const int t = threadIdx.x, j = t / ns, c = t - j * ns;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    double a = 0.0, b = 0.0;
    if (j < nj)
        for (int64_t v = v0 + j; v < v1; v += nj) {
            const float p = P[v * ns + c], tc = target[(int64_t)c * nvox + v];
            a += (double)(p * tc);
            b += (double)(p + tc);
        }
    s_pt[t] = a;
    s_ps[t] = b;
    __syncthreads();
    if (t < ns) {
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < nj; ++k) { sa += s_pt[k * ns + t]; sb += s_ps[k * ns + t]; }
        part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + t] = sa;
        part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + ns + t] = sb;
    }
}

// fold the per-block partials: out[0] = CE sum, out[1..ns] = sum p*t, out[1+ns..] = sum (p+t)
__global__ void seg_fold_kernel(const double* __restrict__ part, int nb, int ns, double* __restrict__ out) {
    const int k = blockIdx.x, l = threadIdx.x;                    // one wave per quantity
    if (k >= 1 + 2 * ns) return;
    double s_ = 0.0;
    for (int b = l; b < nb; b += 64) s_ += part[(int64_t)b * (1 + 2 * ns) + k];
    s_ = wave_reduce_sum(s_);
    if (l == 0) out[k] = s_;
}

// pass 2: d(w_ce*CE_mean + w_dice*Dice)/d(logits) through the softmax, added into dRaw
__global__ void seg_bwd_kernel(const float* __restrict__ P, const float* __restrict__ target, int ns,
                               const float* __restrict__ wce, const float* __restrict__ wdice,
                               const double* __restrict__ sums, int64_t nvox, float coef_ce /* weight/(nvox) */,
                               float coef_dice, int n_out, int c0, float* __restrict__ dRaw) {
    GRID_STRIDE(v, nvox) {
        const float* p = P + v * ns;
        const float* t = target + v;                          // class c at t[c * nvox]
        // dL/dp_c
        float dot = 0.f;
        for (int c = 0; c < ns; ++c) {
            const float pc = p[c];
            float g = 0.f;
            if (pc > 1e-5f) g -= coef_ce * wce[c] * t[(int64_t)c * nvox] / pc;               // clamp(min=1e-5) kills the gradient below it
            const double num = sums[1 + c], den = sums[1 + ns + c];
            if (den > 1e-5) g += coef_dice * wdice[c] * (float)(-2.0 * ((double)t[(int64_t)c * nvox] * den - num) / (den * den));
            dot += g * pc;
        }
        for (int c = 0; c < ns; ++c) {
            const float pc = p[c];
            float g = 0.f;
            if (pc > 1e-5f) g -= coef_ce * wce[c] * t[(int64_t)c * nvox] / pc;
            const double num = sums[1 + c], den = sums[1 + ns + c];
            if (den > 1e-5) g += coef_dice * wdice[c] * (float)(-2.0 * ((double)t[(int64_t)c * nvox] * den - num) / (den * den));
            dRaw[v * n_out + c0 + c] += pc * (g - dot);                         // softmax Jacobian
        }
    }
}

// LDS-tiled versions for ns <= 61 (every label set of the reference): a block stages the logits / probabilities of 256
// voxels with coalesced row segments, each thread then works on its own voxel's row in LDS (odd stride: no bank
// conflicts), and the results leave as coalesced rows again.  The per-thread versions above touch memory with a stride of
// n_out (69) floats per lane: 5.4 + 7.3 ms at 128^3.
// SUMS: the per-class sums of seg_class_sums_kernel ride along -- thread (j, c) adds p * t and p + t of the tile's voxels
// j, j + nj, ... from the probabilities it has in LDS (round 4: the separate pass re-read P and the target, 0.95 ms at 128^3)
template <bool SUMS>
__global__ void __launch_bounds__(256) seg_fwd_tile_kernel(const float* __restrict__ raw, int n_out, int c0, int ns,
                                                           const float* __restrict__ target, const float* __restrict__ wce,
                                                           int64_t nvox, float* __restrict__ P, double* __restrict__ part) {
    extern __shared__ float tile[];                          // [256][ld]
    __shared__ double red[256];
    __shared__ double s_pt[SUMS ? 256 : 1], s_ps[SUMS ? 256 : 1];
    const int ld = ns | 1;
    const int t = threadIdx.x;
    const int nj = 256 / ns, cj = t / ns, cc = t - cj * ns;   // class-sum mapping
    double sa = 0.0, sb = 0.0;
    double ce = 0.0;
    const int64_t ntile = bfm_cdiv64(nvox, 256);
    for (int64_t tb = blockIdx.x; tb < ntile; tb += gridDim.x) {
        const int64_t vb = tb * 256;
        const int nv = (int)min<int64_t>(256, nvox - vb);
        __syncthreads();
        for (int i = t; i < nv * ns; i += 256) {
            const int vl = i / ns, c = i - vl * ns;
            tile[vl * ld + c] = raw[(vb + vl) * n_out + c0 + c];
        }
        __syncthreads();
        if (t < nv) {
            float* r = tile + t * ld;
            float m = -INFINITY;
            for (int c = 0; c < ns; ++c) m = fmaxf(m, r[c]);
            float sum = 0.f;
            for (int c = 0; c < ns; ++c) sum += expf(r[c] - m);
            const float inv = 1.f / sum;
            for (int c = 0; c < ns; ++c) {
                const float pr = expf(r[c] - m) * inv;
                r[c] = pr;
                ce -= (double)(logf(fmaxf(pr, 1e-5f)) * wce[c] * target[(int64_t)c * nvox + vb + t]);
            }
        }
        __syncthreads();
        for (int i = t; i < nv * ns; i += 256) {
            const int vl = i / ns, c = i - vl * ns;
            P[vb * ns + i] = tile[vl * ld + c];
        }
        if (SUMS && cj < nj)
            for (int vl = cj; vl < nv; vl += nj) {
                const float p = tile[vl * ld + cc], tc = target[(int64_t)cc * nvox + vb + vl];
                sa += (double)(p * tc);
                sb += (double)(p + tc);
            }
    }
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * (1 + 2 * ns)] = ce;
    if (SUMS) {
        s_pt[t] = sa;
        s_ps[t] = sb;
        __syncthreads();
        if (t < ns) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < nj; ++k) { a += s_pt[k * ns + t]; b += s_ps[k * ns + t]; }
            part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + t] = a;
            part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + ns + t] = b;
        }
    }
}

__global__ void __launch_bounds__(256) seg_bwd_tile_kernel(const float* __restrict__ P, const float* __restrict__ target, int ns,
                                                           const float* __restrict__ wce, const float* __restrict__ wdice,
                                                           const double* __restrict__ sums, int64_t nvox, float coef_ce,
                                                           float coef_dice, int n_out, int c0, float* __restrict__ dRaw) {
    extern __shared__ float tile[];                          // [256][ld] + 3 * ns class constants
    const int ld = ns | 1;
    const int t = threadIdx.x;
    float* ka = tile + 256 * ld;                             // coef_ce * wce
    float* k1 = ka + ns;                                     // dDice/dp = t * k1 + k0
    float* k0 = k1 + ns;
    for (int c = t; c < ns; c += 256) {
        ka[c] = coef_ce * wce[c];
        const double num = sums[1 + c], den = sums[1 + ns + c];
        const double k = (double)coef_dice * (double)wdice[c];
        k1[c] = den > 1e-5 ? (float)(-2.0 * k / den) : 0.f;
        k0[c] = den > 1e-5 ? (float)(2.0 * k * num / (den * den)) : 0.f;
    }
    const int64_t ntile = bfm_cdiv64(nvox, 256);
    for (int64_t tb = blockIdx.x; tb < ntile; tb += gridDim.x) {
        const int64_t vb = tb * 256;
        const int nv = (int)min<int64_t>(256, nvox - vb);
        __syncthreads();
        for (int i = t; i < nv * ns; i += 256) {
            const int vl = i / ns, c = i - vl * ns;
            tile[vl * ld + c] = P[vb * ns + i];
        }
        __syncthreads();
        if (t < nv) {
            float* r = tile + t * ld;
            float dot = 0.f;
            for (int c = 0; c < ns; ++c) {
                const float pc = r[c], tc = target[(int64_t)c * nvox + vb + t];
                float g = tc * k1[c] + k0[c];
                if (pc > 1e-5f) g -= ka[c] * tc / pc;         // clamp(min=1e-5) kills the gradient below it
                dot += g * pc;
            }
            for (int c = 0; c < ns; ++c) {
                const float pc = r[c], tc = target[(int64_t)c * nvox + vb + t];
                float g = tc * k1[c] + k0[c];
                if (pc > 1e-5f) g -= ka[c] * tc / pc;
                r[c] = pc * (g - dot);                        // softmax Jacobian
            }
        }
        __syncthreads();
        for (int i = t; i < nv * ns; i += 256) {
            const int vl = i / ns, c = i - vl * ns;
            dRaw[(vb + vl) * n_out + c0 + c] += tile[vl * ld + c];
        }
    }
}

// ----------------------------------------------------------------------------- task heads backward
// dFn[v][c] = sum_o dRaw[v][o] * W[o][c];   weights in LDS
__global__ void __launch_bounds__(256) head_dfeat_kernel(const float* __restrict__ dRaw, const float* __restrict__ Wt,
                                                         int n_out, int C, int64_t nvox, float* __restrict__ dFn) {
    extern __shared__ float wl[];                             // [n_out][C]
    for (int i = threadIdx.x; i < n_out * C; i += 256) wl[i] = Wt[i];
    __syncthreads();
    const int64_t n = nvox * C;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        const int64_t v = i / C;
        float s = 0.f;
        for (int o = 0; o < n_out; ++o) s = fmaf(dRaw[v * n_out + o], wl[o * C + c], s);
        dFn[i] = s;
    }
}

// dW[o][c] = sum_v dRaw[v][o] * Fn[v][c]  on the exact-fp32 matrix core; one wave per (split, o-block, c-block)
__global__ void __launch_bounds__(64) head_wgrad_kernel(const float* __restrict__ dRaw, const float* __restrict__ Fn,
                                                        int n_out, int C, int64_t nvox, int64_t vox_per_split,
                                                        float* __restrict__ part /*[S][n_out][C]*/) {
    const int lane = threadIdx.x, l32 = lane & 31, lh = lane >> 5;
    const int ob = blockIdx.y, cb = blockIdx.z;
    const int o = ob * 32 + l32, c = cb * 32 + l32;
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_split, v1 = min(nvox, v0 + vox_per_split);
    for (int64_t v = v0; v < v1; v += 2) {
        const int64_t vv = v + lh;
        const float a = (vv < v1 && o < n_out) ? dRaw[vv * n_out + o] : 0.f;
        const float b = (vv < v1 && c < C) ? Fn[vv * C + c] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    float* out = part + (int64_t)blockIdx.x * n_out * C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = ob * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (row < n_out && c < C) out[(int64_t)row * C + c] = acc[i];
    }
}

__global__ void fold_splits_kernel(const float* __restrict__ part, int S, int64_t n, float* __restrict__ out) {
    GRID_STRIDE(i, n) {
        float s = part[i];
        for (int k = 1; k < S; ++k) s += part[i + (int64_t)k * n];
        out[i] = s;
    }
}

// db[o] = sum_v dRaw[v][o]: per-block partial column sums in fp64, then folded
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ X, int n_out, int64_t nvox,
                                                     int64_t vox_per_block, double* __restrict__ part) {
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    for (int o = threadIdx.x; o < n_out; o += 256) {
        double s = 0.0;
        for (int64_t v = v0; v < v1; ++v) s += (double)X[v * n_out + o];
        part[(int64_t)blockIdx.x * n_out + o] = s;
    }
}
__global__ void colsum_fold_kernel(const double* __restrict__ part, int nb, int n_out, float* __restrict__ out) {
    const int o = blockIdx.x, l = threadIdx.x;                    // one wave per column
    if (o >= n_out) return;
    double s_ = 0.0;
    for (int b = l; b < nb; b += 64) s_ += part[(int64_t)b * n_out + o];
    s_ = wave_reduce_sum(s_);
    if (l == 0) out[o] = (float)s_;
}

// F.normalize backward: f^ = f / max(|f|, eps);  df = (dF^ - f^ <dF^, f^>) / max(|f|, eps)   (|f| > eps branch)
__global__ void normalize_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ dFn, int C, int64_t nvox,
                                     float eps, float* __restrict__ dfeat) {
    GRID_STRIDE(v, nvox) {
        const float* f = feat + v * C;
        const float* g = dFn + v * C;
        float ss = 0.f;
        for (int c = 0; c < C; ++c) ss = fmaf(f[c], f[c], ss);
        const float nrm = sqrtf(ss);
        if (nrm > eps) {
            const float inv = 1.f / nrm;
            float dot = 0.f;
            for (int c = 0; c < C; ++c) dot = fmaf(g[c], f[c] * inv, dot);
            for (int c = 0; c < C; ++c) dfeat[v * C + c] = (g[c] - f[c] * inv * dot) * inv;
        } else {
            for (int c = 0; c < C; ++c) dfeat[v * C + c] = g[c] / eps;
        }
    }
}

// the same with C/4 lanes per voxel (C/4 a power of two <= 64): float4 per lane, shuffle reductions -- coalesced rows
template <int LPV>
__global__ void __launch_bounds__(256) normalize_bwd_rows_kernel(const float* __restrict__ feat, const float* __restrict__ dFn,
                                                                 int64_t nvox, float eps, float* __restrict__ dfeat) {
    constexpr int C = LPV * 4;
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x / LPV;
    const int q = (int)(gt % LPV);
    for (int64_t v = gt / LPV; v < nvox + (stride - nvox % stride) % stride; v += stride) {   // whole waves stay in step
        const bool ok = v < nvox;
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f), g = f;
        if (ok) {
            f = *reinterpret_cast<const float4*>(feat + v * C + q * 4);
            g = *reinterpret_cast<const float4*>(dFn + v * C + q * 4);
        }
        float ss = f.x * f.x + f.y * f.y + f.z * f.z + f.w * f.w;
#pragma unroll
        for (int o = LPV / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float nrm = sqrtf(ss);
        float4 r;
        if (nrm > eps) {
            const float inv = 1.f / nrm;
            float dot = (g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w) * inv;
#pragma unroll
            for (int o = LPV / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
            r = make_float4((g.x - f.x * inv * dot) * inv, (g.y - f.y * inv * dot) * inv, (g.z - f.z * inv * dot) * inv,
                            (g.w - f.w * inv * dot) * inv);
        } else {
            float dot = 0.f;
#pragma unroll
            for (int o = LPV / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);   // keep the shuffles convergent
            r = make_float4(g.x / eps, g.y / eps, g.z / eps, g.w / eps);
        }
        if (ok) *reinterpret_cast<float4*>(dfeat + v * C + q * 4) = r;
    }
}

// ----------------------------------------------------------------------------- optimiser / scaler helpers
// torch.optim.AdamW (amsgrad off, maximize off): p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bias1, float bias2_sqrt, float grad_scale) {
    GRID_STRIDE(i, n) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bias2_sqrt + eps;
        pi = pi - (lr / bias1) * (mi / denom);
        p[i] = pi;
    }
}

// sum of squares + non-finite flag of a gradient tensor (clip_gradients / GradScaler.unscale_)
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part,
                                                    int* __restrict__ nonfinite) {
    __shared__ double red[256];
    double acc = 0.0;
    bool bad = false;
    GRID_STRIDE(i, n) {
        const float x = g[i];
        if (!(fabsf(x) <= 3.402823466e+38f)) bad = true;
        acc += (double)x * (double)x;
    }
    if (bad) atomicOr(nonfinite, 1);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

int grid_for(int64_t n, int cap = 4096) { return (int)std::min<int64_t>(cap, std::max<int64_t>(1, bfm_cdiv64(n, 256))); }

// ---- all parameters of the step in one launch each (round 4): the per-tensor kernels above cost 252 x 3 launches per
// iteration, most of them over a few hundred floats (GroupNorm gamma / beta).  A block owns one chunk of one tensor.
__global__ void __launch_bounds__(256) sumsq_multi_kernel(const bfm_adam_tensor_t* __restrict__ T, const int32_t* __restrict__ chunk_tensor,
                                                          int64_t chunk_elems, double* __restrict__ part, int* __restrict__ nonfinite) {
    __shared__ double red[256];
    const int t = chunk_tensor[blockIdx.x];
    const bfm_adam_tensor_t d = T[t];
    const int64_t i0 = (int64_t)(blockIdx.x - d.first_chunk) * chunk_elems, i1 = min(d.n, i0 + chunk_elems);
    const float* g = d.g;
    double acc = 0.0;
    bool bad = false;
    if (((reinterpret_cast<uintptr_t>(g) & 15) == 0) && ((i1 - i0) & 3) == 0) {
        const float4* g4 = reinterpret_cast<const float4*>(g + i0);
        for (int64_t i = threadIdx.x; i < (i1 - i0) >> 2; i += 256) {
            const float4 x = g4[i];
            if (!(fabsf(x.x) <= 3.402823466e+38f) || !(fabsf(x.y) <= 3.402823466e+38f) || !(fabsf(x.z) <= 3.402823466e+38f) ||
                !(fabsf(x.w) <= 3.402823466e+38f))
                bad = true;
            acc += (double)x.x * (double)x.x + (double)x.y * (double)x.y + (double)x.z * (double)x.z + (double)x.w * (double)x.w;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float x = g[i];
            if (!(fabsf(x) <= 3.402823466e+38f)) bad = true;
            acc += (double)x * (double)x;
        }
    }
    if (bad) atomicOr(nonfinite, 1);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void sumsq_multi_fold_kernel(const bfm_adam_tensor_t* __restrict__ T, int ntensors, int64_t chunk_elems,
                                        const double* __restrict__ part, double* __restrict__ out) {
    const int t = blockIdx.x, l = threadIdx.x;                    // one wave per tensor, fixed order
    if (t >= ntensors) return;
    const int c0 = T[t].first_chunk, nc = (int)((T[t].n + chunk_elems - 1) / chunk_elems);
    double s_ = 0.0;
    for (int c = l; c < nc; c += 64) s_ += part[c0 + c];
    s_ = wave_reduce_sum(s_);
    if (l == 0) out[t] = s_;
}

__global__ void __launch_bounds__(256) adamw_multi_kernel(const bfm_adam_tensor_t* __restrict__ T, const int32_t* __restrict__ chunk_tensor,
                                                          int64_t chunk_elems, float lr, float b1, float b2, float eps, float wd) {
    const int t = chunk_tensor[blockIdx.x];
    const bfm_adam_tensor_t d = T[t];
    const int64_t i0 = (int64_t)(blockIdx.x - d.first_chunk) * chunk_elems, i1 = min(d.n, i0 + chunk_elems);
    const float gs = d.grad_scale, bias1 = d.bias1, bias2_sqrt = d.bias2_sqrt;
    auto one = [&](float& pi, float gi, float& mi, float& vi) __attribute__((always_inline)) {      // adamw_kernel's expressions
        gi = gi * gs;
        pi = pi * (1.f - lr * wd);
        mi = b1 * mi + (1.f - b1) * gi;
        vi = b2 * vi + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bias2_sqrt + eps;
        pi = pi - (lr / bias1) * (mi / denom);
    };
    const uintptr_t al = reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g) | reinterpret_cast<uintptr_t>(d.m) |
                         reinterpret_cast<uintptr_t>(d.v);
    if ((al & 15) == 0 && ((i1 - i0) & 3) == 0) {
        float4* p4 = reinterpret_cast<float4*>(d.p + i0);
        const float4* g4 = reinterpret_cast<const float4*>(d.g + i0);
        float4* m4 = reinterpret_cast<float4*>(d.m + i0);
        float4* v4 = reinterpret_cast<float4*>(d.v + i0);
        for (int64_t i = threadIdx.x; i < (i1 - i0) >> 2; i += 256) {
            float4 pp = p4[i], mm = m4[i], vv = v4[i];
            const float4 gg = g4[i];
            one(pp.x, gg.x, mm.x, vv.x);
            one(pp.y, gg.y, mm.y, vv.y);
            one(pp.z, gg.z, mm.z, vv.z);
            one(pp.w, gg.w, mm.w, vv.w);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            float pi = d.p[i], mi = d.m[i], vi = d.v[i];
            one(pi, d.g[i], mi, vi);
            d.p[i] = pi; d.m[i] = mi; d.v[i] = vi;
        }
    }
}


// ============================================================================ "rows" layout of the head outputs (round 4)
// raw / dRaw as [n_out] rows of nvox values (row pitch rs >= nvox) instead of [nvox][n_out]: what the loss kernels walk is
// a handful of columns per entry, which in the channels-last form costs one 4-byte element out of every 276-byte voxel row
// (the LDS-tiled kernels above recover coalescing by staging whole rows, at 1.4-1.6 ms per pass at 128^3).  In rows every
// access of a wave is one 256-byte segment, nothing goes through LDS and each kernel is one grid-stride loop with the SAME
// per-voxel expressions as its channels-last twin (tests compare the two layouts bit for bit where the order allows).
__global__ void __launch_bounds__(256) l1_multi_rows_kernel(const float* __restrict__ raw, int64_t rs, int64_t nvox,
                                                            const L1Table tab, float* dRaw,
                                                            double* __restrict__ part /*[nb][L1M_MAX]*/) {
    __shared__ double red[256];
    double acc[L1M_MAX];
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) acc[k] = 0.0;
    GRID_STRIDE(v, nvox) {
#pragma unroll
        for (int k = 0; k < L1M_MAX; ++k) {
            if (k >= tab.n) continue;
            const L1Entry& e = tab.e[k];
            float o = raw[(int64_t)e.col * rs + v];
            bool live = true;
            if (e.clampv > 0.f) {
                if (o > e.clampv) { o = e.clampv; live = false; }
                else if (o < -e.clampv) { o = -e.clampv; live = false; }
            }
            const float m = e.mask ? e.mask[v] : 1.f;
            const float w = e.weight ? e.weight[v] : 1.f;
            const float d = o * m - e.target[v] * m;
            acc[k] += (double)((e.l2 ? d * d : fabsf(d)) * w);
            const float sg = e.l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            if (live && dRaw) dRaw[(int64_t)e.col * rs + v] += e.coef * sg * w * m;
        }
    }
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) {
        if (k >= tab.n) continue;
        const double sk = block_sum(acc[k], red);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * L1M_MAX + k] = sk;
    }
}

__global__ void __launch_bounds__(256) grad_l1_multi_rows_kernel(const float* __restrict__ raw, int64_t rs, const GL1Table tab,
                                                                 int D, int H, int W, float* __restrict__ dRaw,
                                                                 double* __restrict__ part /*[nb][L1M_MAX]*/) {
    __shared__ double red[256];
    const int64_t nvox = (int64_t)D * H * W;
    const int64_t s = (int64_t)H * W;
    double acc[L1M_MAX];
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) acc[k] = 0.0;
    GRID_STRIDE(v, nvox) {
        const int x = (int)(v % W);
        const int64_t t2 = v / W;
        const int y = (int)(t2 % H), z = (int)(t2 / H);
#pragma unroll
        for (int k = 0; k < L1M_MAX; ++k) {
            if (k >= tab.n) continue;
            const GL1Entry& e = tab.e[k];
            const float* target = e.target;
            const float* weight = e.weight;
            const float* r0 = raw + (int64_t)e.col * rs + v;
            const float o = r0[0], t = target[v];
            const float w = weight ? weight[v] : 1.f;
            float g = 0.f;
            double a = 0.0;
            if (x + 1 < W) {
                const float d = (r0[1] - o) - (target[v + 1] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (y + 1 < H) {
                const float d = (r0[W] - o) - (target[v + W] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (z + 1 < D) {
                const float d = (r0[s] - o) - (target[v + s] - t);
                a += (double)(fabsf(d) * w);
                g -= (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
            }
            if (x > 0) {
                const float d = (o - r0[-1]) - (t - target[v - 1]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - 1] : 1.f);
            }
            if (y > 0) {
                const float d = (o - r0[-W]) - (t - target[v - W]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - W] : 1.f);
            }
            if (z > 0) {
                const float d = (o - r0[-s]) - (t - target[v - s]);
                g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * (weight ? weight[v - s] : 1.f);
            }
            acc[k] += a;
            if (dRaw) dRaw[(int64_t)e.col * rs + v] += e.coef * g;
        }
    }
#pragma unroll
    for (int k = 0; k < L1M_MAX; ++k) {
        if (k >= tab.n) continue;
        const double sk = block_sum(acc[k], red);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * L1M_MAX + k] = sk;
    }
}

// softmax + CE of one voxel per thread, the logits of its ns <= LIM classes in registers; P leaves as [ns][nvox] rows
template <int LIM>
__global__ void __launch_bounds__(256) seg_fwd_rows_kernel(const float* __restrict__ raw, int64_t rs, int c0, int ns,
                                                           const float* __restrict__ target, const float* __restrict__ wce,
                                                           int64_t nvox_, float* __restrict__ P, double* __restrict__ part) {
    __shared__ double red[256];
    double ce = 0.0;
    int64_t nvox = nvox_;
    GRID_STRIDE(v, nvox_) {
        asm volatile("" : "+s"(nvox));
        // (opaque per iteration: hoisted out of the voxel loop, the LIM row addresses take 2 LIM scalar registers and spill)
        asm volatile("" : "+s"(rs));
        const float* r = raw + (int64_t)c0 * rs + v;
        float sv[LIM];
#pragma unroll
        for (int c = 0; c < LIM; ++c) sv[c] = c < ns ? r[(int64_t)c * rs] : -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < LIM; ++c) if (c < ns) m = fmaxf(m, sv[c]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < LIM; ++c) if (c < ns) sum += expf(sv[c] - m);
        const float inv = 1.f / sum;
#pragma unroll
        for (int c = 0; c < LIM; ++c) {
            if (c < ns) {
                const float pr = expf(sv[c] - m) * inv;
                P[(int64_t)c * nvox + v] = pr;
                ce -= (double)(logf(fmaxf(pr, 1e-5f)) * wce[c] * target[(int64_t)c * nvox + v]);
            }
        }
    }
    ce = block_sum(ce, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * (1 + 2 * ns)] = ce;
}

// class sums of one (voxel chunk, class): both operands are rows
__global__ void __launch_bounds__(256) seg_class_sums_rows_kernel(const float* __restrict__ P, const float* __restrict__ target,
                                                                  int ns, int64_t nvox, int64_t vox_per_block,
                                                                  double* __restrict__ part) {
    __shared__ double red[256];
    const int c = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    const float* p = P + (int64_t)c * nvox;
    const float* t = target + (int64_t)c * nvox;
    double a = 0.0, b = 0.0;
    for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const float pv = p[v], tc = t[v];
        a += (double)(pv * tc);
        b += (double)(pv + tc);
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    if (threadIdx.x == 0) {
        part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + c] = a;
        part[(int64_t)blockIdx.x * (1 + 2 * ns) + 1 + ns + c] = b;
    }
}

template <int LIM>
__global__ void __launch_bounds__(256) seg_bwd_rows_kernel(const float* __restrict__ P, const float* __restrict__ target, int ns,
                                                           const float* __restrict__ wce, const float* __restrict__ wdice,
                                                           const double* __restrict__ sums, int64_t nvox_, float coef_ce,
                                                           float coef_dice, int64_t rs, int c0, float* __restrict__ dRaw) {
    __shared__ float ka[LIM], k1[LIM], k0[LIM];
    for (int c = threadIdx.x; c < ns; c += 256) {
        ka[c] = coef_ce * wce[c];
        const double num = sums[1 + c], den = sums[1 + ns + c];
        const double k = (double)coef_dice * (double)wdice[c];
        k1[c] = den > 1e-5 ? (float)(-2.0 * k / den) : 0.f;
        k0[c] = den > 1e-5 ? (float)(2.0 * k * num / (den * den)) : 0.f;
    }
    __syncthreads();
    int64_t nvox = nvox_;
    GRID_STRIDE(v, nvox_) {
        asm volatile("" : "+s"(nvox), "+s"(rs));              // see seg_fwd_rows_kernel
        float pv[LIM], gv[LIM];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < LIM; ++c) {
            if (c < ns) {
                const float pc = P[(int64_t)c * nvox + v], tc = target[(int64_t)c * nvox + v];
                float g = tc * k1[c] + k0[c];
                if (pc > 1e-5f) g -= ka[c] * tc / pc;         // clamp(min=1e-5) kills the gradient below it
                dot += g * pc;
                pv[c] = pc;
                gv[c] = g;
            }
        }
        float* d = dRaw + (int64_t)c0 * rs + v;
#pragma unroll
        for (int c = 0; c < LIM; ++c)
            if (c < ns) d[(int64_t)c * rs] += pv[c] * (gv[c] - dot);                     // softmax Jacobian
    }
}

// ----------------------------------------------------------------------------- pathology head (round 6)
// criterion.py:193-212 loss_pathol_ce / loss_pathol_dice on p = sigmoid(raw) (PatholProcessor, joiner.py:79-87), one channel:
//   ce = mean_v( -log(max(p, 1e-5)) * t ),   dice = 1 - 2 sum(p t) / max(sum(p + t), 1e-5)
// The head output of voxel v is raw[col_off + v * vstride] (channels-last: col_off = column, vstride = n_out; rows: col_off =
// column * row stride, vstride = 1).  part: [3][RB] block partials (fp64) of { -log(max(p,1e-5)) t, p t, p + t }.
__global__ void __launch_bounds__(256) pathol_fwd_kernel(const float* __restrict__ raw, int64_t col_off, int64_t vstride,
                                                         const float* __restrict__ target, int64_t nvox,
                                                         double* __restrict__ part) {
    __shared__ double red[256];
    double a = 0.0, b = 0.0, c = 0.0;
    GRID_STRIDE(v, nvox) {
        const float x = raw[col_off + v * vstride];
        const float p = 1.0f / (1.0f + expf(-x));
        const float t = target[v];
        a += (double)(-logf(fmaxf(p, 1e-5f)) * t);
        b += (double)(p * t);
        c += (double)(p + t);
    }
    a = block_sum(a, red);
    b = block_sum(b, red);
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = a;
        part[RB + blockIdx.x] = b;
        part[2 * RB + blockIdx.x] = c;
    }
}

// sums[0..2] = the three totals; loss_out[0] = ce, loss_out[1] = dice
__global__ void pathol_fold_kernel(const double* __restrict__ part, int nb, double inv_nvox, double* __restrict__ sums,
                                   double* __restrict__ loss_ce, double* __restrict__ loss_dice) {
    __shared__ double red[256];
    double t[3];
    for (int k = 0; k < 3; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nb; i += 256) s += part[k * RB + i];
        t[k] = block_sum(s, red);
    }
    if (threadIdx.x == 0) {
        sums[0] = t[0]; sums[1] = t[1]; sums[2] = t[2];
        if (loss_ce) loss_ce[0] = t[0] * inv_nvox;
        if (loss_dice) loss_dice[0] = 1.0 - 2.0 * t[1] / fmax(t[2], 1e-5);
    }
}

__global__ void __launch_bounds__(256) pathol_bwd_kernel(const float* __restrict__ raw, int64_t col_off, int64_t vstride,
                                                         const float* __restrict__ target, int64_t nvox,
                                                         const double* __restrict__ sums, float coef_ce, float coef_dice,
                                                         float* __restrict__ dRaw) {
    const double num = sums[1], den = sums[2];
    const bool clamped = den < 1e-5;                       // torch.clamp(min=1e-5): no gradient through the denominator then
    const double dc = clamped ? 1e-5 : den;
    const float g_t = (float)(-2.0 / dc);                  // d dice / d p = -2 t / dc + 2 num / dc^2 (the latter only unclamped)
    const float g_1 = clamped ? 0.f : (float)(2.0 * num / (dc * dc));
    GRID_STRIDE(v, nvox) {
        const int64_t o = col_off + v * vstride;
        const float x = raw[o];
        const float p = 1.0f / (1.0f + expf(-x));
        const float t = target[v];
        float g = coef_dice * (g_t * t + g_1);
        if (p > 1e-5f) g += coef_ce * (-t / p);            // clamp(p, min=1e-5): zero gradient where it clamps
        dRaw[o] += g * p * (1.0f - p);
    }
}

}  // namespace

extern "C" size_t bfm_loss_workspace(int ns) { return (size_t)RB * (1 + 2 * (ns > 0 ? ns : 0)) * sizeof(double) + 4096; }

extern "C" int bfm_loss_l1(const float* raw, int n_out, int co, const float* target, const float* weight,
                           const float* mask_mul, int64_t nvox, float clampv, int l2, float coef, float* dRaw,
                           double* loss_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw || !target || !loss_out || !workspace || nvox <= 0 || co < 0 || co >= n_out) return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_workspace(0)) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipLaunchKernelGGL(l1_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), raw, n_out, co, target, weight, mask_mul, nvox,
                       clampv, l2, coef / (float)nvox, dRaw, part);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, bfm_s(stream), part, nb, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

extern "C" size_t bfm_loss_l1_multi_workspace(void) { return (size_t)RB * L1M_MAX * sizeof(double) + 256; }

extern "C" int bfm_loss_l1_multi(const float* raw, int n_out, int64_t nvox, int n, const int32_t* cols, const int32_t* l2,
                                 const float* clampv, const float* coef, const float* const* targets,
                                 const float* const* weights, const float* const* masks, float* dRaw, double* loss_out,
                                 void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw || nvox <= 0 || n_out <= 0 || n <= 0 || n > L1M_MAX || !cols || !l2 || !clampv || !coef || !targets ||
        !loss_out || !workspace)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_l1_multi_workspace()) return BFM_E_WORKSPACE;
    L1Table tab{};
    tab.n = n;
    for (int k = 0; k < n; ++k) {
        if (cols[k] < 0 || cols[k] >= n_out || !targets[k]) return BFM_E_ARG;
        tab.e[k] = L1Entry{cols[k], l2[k], clampv[k], coef[k] / (float)nvox, targets[k], weights ? weights[k] : nullptr,
                           masks ? masks[k] : nullptr};
    }
    const int ld = n_out | 1;
    int tile_vox = (int)((60 * 1024) / (2 * sizeof(float) * ld));
    if (tile_vox > 256) tile_vox = 256;
    tile_vox &= ~31;
    if (tile_vox < 32) return BFM_E_SHAPE;
    double* part = static_cast<double*>(workspace);
    const int nb = (int)std::min<int64_t>(RB, bfm_cdiv64(nvox, tile_vox));
    hipStream_t st = bfm_s(stream);
    hipLaunchKernelGGL(l1_multi_kernel, dim3(nb), dim3(256), (size_t)2 * tile_vox * ld * sizeof(float), st, raw, n_out, nvox,
                       tile_vox, tab, dRaw, part);
    hipLaunchKernelGGL(l1_multi_fold_kernel, dim3(n), dim3(64), 0, st, part, nb, n, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

extern "C" int bfm_loss_grad_l1(const float* raw, int n_out, int co, const float* target, const float* weight, int D,
                                int H, int W, float coef, float* dRaw, double* loss_out, void* workspace,
                                size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw || !target || !loss_out || !workspace || D <= 0 || H <= 0 || W <= 0 || co < 0 || co >= n_out)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_workspace(0)) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int64_t nvox = (int64_t)D * H * W;
    const int nb = grid_for(nvox, RB);
    hipLaunchKernelGGL(grad_l1_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), raw, n_out, co, target, weight, D, H, W,
                       coef / (float)nvox, dRaw, part);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, bfm_s(stream), part, nb, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

extern "C" int bfm_loss_grad_l1_multi(const float* raw, int n_out, int n, const int32_t* cols, const float* coef,
                                      const float* const* targets, const float* const* weights, int D, int H, int W,
                                      float* dRaw, double* loss_out, void* workspace, size_t workspace_bytes,
                                      bfm_stream_t stream) {
    if (!raw || !cols || !coef || !targets || !loss_out || !workspace || n <= 0 || n > L1M_MAX || D <= 0 || H <= 0 || W <= 0)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_l1_multi_workspace()) return BFM_E_WORKSPACE;
    const int64_t nvox = (int64_t)D * H * W;
    GL1Table tab{};
    tab.n = n;
    for (int k = 0; k < n; ++k) {
        if (cols[k] < 0 || cols[k] >= n_out || !targets[k]) return BFM_E_ARG;
        for (int j = 0; j < k; ++j)
            if (cols[j] == cols[k]) return BFM_E_ARG;               // two entries on one column would race on dRaw
        tab.e[k] = GL1Entry{cols[k], coef[k] / (float)nvox, targets[k], weights ? weights[k] : nullptr};
    }
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipLaunchKernelGGL(grad_l1_multi_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), raw, n_out, tab, D, H, W, dRaw, part);
    hipLaunchKernelGGL(l1_multi_fold_kernel, dim3(n), dim3(64), 0, bfm_s(stream), part, nb, n, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

// loss_out[0] = CE (mean over voxels), loss_out[1] = Dice; P = softmax probabilities [nvox][ns] (scratch / output)
extern "C" int bfm_loss_seg(const float* raw, int n_out, int c0, int ns, const float* target, const float* wce,
                            const float* wdice, int64_t nvox, float coef_ce, float coef_dice, float* P, float* dRaw,
                            double* loss_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw || !target || !wce || !wdice || !P || !loss_out || !workspace || nvox <= 0 || ns <= 0 || c0 < 0 ||
        c0 + ns > n_out)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_workspace(ns)) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipStream_t st = bfm_s(stream);
    const int64_t vpb = bfm_cdiv64(nvox, nb);
    if (ns > 256 || bfm_cdiv64(nvox, vpb) > nb) return BFM_E_SHAPE;
    const bool tiled = ns <= 61;                                 // 256 rows of (ns | 1) floats + constants within 64 KB
    const size_t tile_bytes = (size_t)(256 * (ns | 1) + 3 * ns) * sizeof(float);
    if (tiled) {
        hipLaunchKernelGGL(seg_fwd_tile_kernel<true>, dim3(nb), dim3(256), tile_bytes, st, raw, n_out, c0, ns, target, wce, nvox, P,
                           part);
    } else {
        hipLaunchKernelGGL(seg_fwd_kernel, dim3(nb), dim3(256), 0, st, raw, n_out, c0, ns, target, wce, nvox, P, part);
        // blocks past the last voxel chunk still write zeros: the fold below reads all nb rows
        hipLaunchKernelGGL(seg_class_sums_kernel, dim3(nb), dim3(256), 0, st, P, target, ns, nvox, vpb, part);
    }
    double* sums = part + (size_t)RB * (1 + 2 * ns);             // [1 + 2 ns]
    hipLaunchKernelGGL(seg_fold_kernel, dim3(1 + 2 * ns), dim3(64), 0, st, part, nb, ns, sums);
    if (dRaw) {
        if (tiled)
            hipLaunchKernelGGL(seg_bwd_tile_kernel, dim3(grid_for(nvox, 2048)), dim3(256), tile_bytes, st, P, target, ns, wce, wdice, sums,
                               nvox, coef_ce / (float)nvox, coef_dice, n_out, c0, dRaw);
        else
            hipLaunchKernelGGL(seg_bwd_kernel, dim3(grid_for(nvox)), dim3(256), 0, st, P, target, ns, wce, wdice, sums, nvox,
                               coef_ce / (float)nvox, coef_dice, n_out, c0, dRaw);
    }
    // loss values are finished on the host from `sums` (CE mean and the Dice sum need wdice): copy them out
    (void)hipMemcpyAsync(loss_out, sums, (size_t)(1 + 2 * ns) * sizeof(double), hipMemcpyDeviceToDevice, st);
    return bfm_launch_status();
}

// The three products of the heads' backward in ONE pass over dRaw [nvox][n_out] and Fn [nvox][64] (round 4; the separate
// kernels read dRaw three times with 276-byte strides: 4.6 ms per 128^3 sample).  A workgroup walks tiles of 64 voxels:
// the dRaw and Fn tiles and the head weights sit in LDS; on the exact-fp32 matrix core (v_mfma_f32_32x32x2_f32)
//   dFn tile [64 x 64] = dRaw tile [64 x n_out] . W [n_out x 64]          (4 waves x one 32 x 32 block, written out)
//   dW       [n_out x 64] += dRaw tile^T [n_out x 64 vox] . Fn tile       (<= 6 blocks of 32 x 32, kept in registers)
// and the column sums of dRaw (fp64) for db ride along; a workgroup leaves one dW / db partial, folded in block order.
constexpr int HB_TV = 64, HB_C = 64, HB_BLOCKS = 768, HB_MAXO = 96;
__global__ void __launch_bounds__(256) head_bwd_fused_kernel(const float* __restrict__ dRaw, const float* __restrict__ Fn,
                                                             const float* __restrict__ Wt, int n_out, int64_t nvox,
                                                             int64_t rs /* > 0: dRaw as [n_out] rows of pitch rs */,
                                                             float* __restrict__ dFn, float* __restrict__ wpart,
                                                             double* __restrict__ bpart) {
    extern __shared__ float hsm[];
    const int ldr = (n_out + 2) | 1;                              // dRaw tile row: n_out values, zero padded to an even K, odd stride
    const int K2 = (n_out + 1) & ~1;
    float* sR = hsm;                                              // [HB_TV][ldr]
    float* sF = sR + HB_TV * ldr;                                 // [HB_TV][HB_C]
    float* sW = sF + HB_TV * HB_C;                                // [K2][HB_C]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, l32 = lane & 31, lh = lane >> 5;
    for (int i = t; i < K2 * HB_C; i += 256) sW[i] = i < n_out * HB_C ? Wt[i] : 0.f;
    const int nob = (n_out + 31) >> 5;                            // 32-row blocks of dW: <= 3
    // dW blocks (ob, cb) dealt round-robin to the four waves: wave w owns blocks w and w + 4 of the 2 * nob
    floatx16 accW[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) accW[u][i] = 0.f;
    double colsum = 0.0;
    const int64_t ntile = (nvox + HB_TV - 1) / HB_TV;
    for (int64_t tb = blockIdx.x; tb < ntile; tb += gridDim.x) {
        const int64_t v0 = tb * HB_TV;
        const int nv = (int)min<int64_t>(HB_TV, nvox - v0);
        __syncthreads();
        if (rs > 0) {
            for (int i = t; i < HB_TV * n_out; i += 256) {        // one 256-byte segment of a row per wave
                const int o = i >> 6, vl = i & (HB_TV - 1);
                sR[vl * ldr + o] = vl < nv ? dRaw[(int64_t)o * rs + v0 + vl] : 0.f;
            }
        } else {
            for (int i = t; i < HB_TV * n_out; i += 256) {        // the tile's rows are one contiguous run of dRaw
                const int vl = i / n_out, o = i - vl * n_out;
                sR[vl * ldr + o] = vl < nv ? dRaw[v0 * n_out + i] : 0.f;
            }
        }
        for (int i = t; i < HB_TV * (ldr - n_out); i += 256) {    // the padding columns
            const int vl = i / (ldr - n_out), o = n_out + (i - vl * (ldr - n_out));
            sR[vl * ldr + o] = 0.f;
        }
        for (int i = t; i < HB_TV * HB_C / 4; i += 256) {
            const int vl = (i * 4) / HB_C;
            reinterpret_cast<float4*>(sF)[i] = vl < nv ? reinterpret_cast<const float4*>(Fn + v0 * HB_C)[i]
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // ---- dFn block of wave w: rows vb * 32.., columns cb * 32..
        {
            const int vb = w >> 1, cb = w & 1;
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const float* ap = sR + (vb * 32 + l32) * ldr + lh;
            const float* bp = sW + lh * HB_C + cb * 32 + l32;
            for (int k = 0; k < K2; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k * HB_C], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int vl = vb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                if (vl < nv) dFn[(v0 + vl) * HB_C + cb * 32 + l32] = acc[i];
            }
        }
        // ---- dW blocks of wave w
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int blk = w + 4 * u;
            if (blk < 2 * nob) {                                   // wave-uniform
                const int ob = blk >> 1, cb = blk & 1;
                const int o = ob * 32 + l32;
                const bool ok = o < n_out;
                const float* ap = sR + lh * ldr + (ok ? o : 0);
                const float* bp = sF + lh * HB_C + cb * 32 + l32;
                for (int v = 0; v < HB_TV; v += 2) {
                    const float a = ok ? ap[v * ldr] : 0.f;
                    accW[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[v * HB_C], accW[u], 0, 0, 0);
                }
            }
        }
        if (t < n_out) {
            double s_ = 0.0;
            for (int v = 0; v < HB_TV; ++v) s_ += (double)sR[v * ldr + t];
            colsum += s_;
        }
    }
    float* wout = wpart + (int64_t)blockIdx.x * n_out * HB_C;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int blk = w + 4 * u;
        if (blk < 2 * nob) {
            const int ob = blk >> 1, cb = blk & 1;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = ob * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                if (row < n_out) wout[(int64_t)row * HB_C + cb * 32 + l32] = accW[u][i];
            }
        }
    }
    if (t < n_out) bpart[(int64_t)blockIdx.x * n_out + t] = colsum;
}

extern "C" size_t bfm_head_bwd_workspace(int n_out, int C, int64_t nvox) {
    const int S = (int)std::min<int64_t>(256, std::max<int64_t>(1, nvox / 4096));
    const size_t unfused = (size_t)S * n_out * C * sizeof(float) + (size_t)RB * n_out * sizeof(double) + 256;
    const size_t fused = (size_t)HB_BLOCKS * n_out * C * sizeof(float) + 256 + (size_t)HB_BLOCKS * n_out * sizeof(double);
    return std::max(unfused, fused);
}

static int head_bwd_launch(const float* dRaw, int64_t row_stride, const float* Fn, const float* head_w, int n_out, int C,
                           int64_t nvox, float* dW, float* db, float* dFn, void* workspace, size_t workspace_bytes,
                           bfm_stream_t stream) {
    if (!dRaw || !Fn || !head_w || !dW || !db || !dFn || !workspace || n_out <= 0 || C <= 0 || nvox <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_head_bwd_workspace(n_out, C, nvox)) return BFM_E_WORKSPACE;
    if ((size_t)n_out * C * sizeof(float) > 64 * 1024) return BFM_E_SHAPE;
    hipStream_t st = bfm_s(stream);
    bool fused = C == HB_C && n_out <= HB_MAXO && (reinterpret_cast<uintptr_t>(Fn) & 15) == 0;
    const int ldr = (n_out + 2) | 1, K2 = (n_out + 1) & ~1;
    const size_t smem = ((size_t)HB_TV * ldr + (size_t)HB_TV * HB_C + (size_t)K2 * HB_C) * sizeof(float);
    if (fused && smem > 64 * 1024) {
        // more than 64 KB of dynamic LDS (n_out 95, 96): allowed on this part once the attribute is set; where the device
        // has less, the three-pass kernels below take over -- except in the rows layout, which only this
        // kernel reads
        static int big_ok = -1;                                   // -1 unknown, 0 refused, 1 set
        if (big_ok < 0) {
            int dev = 0, lim = 0;
            big_ok = hipGetDevice(&dev) == hipSuccess &&
                     hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
                     (size_t)lim >= ((size_t)HB_TV * (HB_MAXO + 3) + (size_t)HB_TV * HB_C + (size_t)HB_MAXO * HB_C) * sizeof(float) &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&head_bwd_fused_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
        }
        if (!big_ok) fused = false;
    }
    if (row_stride > 0 && !fused) return BFM_E_SHAPE;             // the rows layout exists for the one-pass kernel only
    if (fused) {
        const int64_t ntile = (nvox + HB_TV - 1) / HB_TV;
        const int nb = (int)std::min<int64_t>(HB_BLOCKS, ntile);
        float* wpart = static_cast<float*>(workspace);
        double* bpart = reinterpret_cast<double*>(static_cast<char*>(workspace) +
                                                  (((size_t)HB_BLOCKS * n_out * C * sizeof(float) + 255) & ~(size_t)255));
        hipLaunchKernelGGL(head_bwd_fused_kernel, dim3(nb), dim3(256), smem, st, dRaw, Fn, head_w, n_out, nvox, row_stride, dFn,
                           wpart, bpart);
        hipLaunchKernelGGL(fold_splits_kernel, dim3(grid_for((int64_t)n_out * C)), dim3(256), 0, st, wpart, nb,
                           (int64_t)n_out * C, dW);
        hipLaunchKernelGGL(colsum_fold_kernel, dim3(n_out), dim3(64), 0, st, bpart, nb, n_out, db);
        return bfm_launch_status();
    }
    hipLaunchKernelGGL(head_dfeat_kernel, dim3(grid_for(nvox * C)), dim3(256), (size_t)n_out * C * sizeof(float), st, dRaw,
                       head_w, n_out, C, nvox, dFn);
    const int S = (int)std::min<int64_t>(256, std::max<int64_t>(1, nvox / 4096));
    int64_t vps = bfm_cdiv64(nvox, S);
    vps += vps & 1;                                              // even: the K pair never straddles two splits
    const int S2 = (int)bfm_cdiv64(nvox, vps);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(S2, bfm_cdiv(n_out, 32), bfm_cdiv(C, 32)), dim3(64), 0, st, dRaw, Fn, n_out, C,
                       nvox, vps, part);
    hipLaunchKernelGGL(fold_splits_kernel, dim3(grid_for((int64_t)n_out * C)), dim3(256), 0, st, part, S2,
                       (int64_t)n_out * C, dW);
    double* cpart = reinterpret_cast<double*>(static_cast<char*>(workspace) + (((size_t)S * n_out * C * sizeof(float) + 255) & ~(size_t)255));
    const int nb = grid_for(nvox, RB);
    const int64_t vpb = bfm_cdiv64(nvox, nb);
    const int nb2 = (int)bfm_cdiv64(nvox, vpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(nb2), dim3(256), 0, st, dRaw, n_out, nvox, vpb, cpart);
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(n_out), dim3(64), 0, st, cpart, nb2, n_out, db);
    return bfm_launch_status();
}

extern "C" int bfm_head_bwd(const float* dRaw, const float* Fn, const float* head_w, int n_out, int C, int64_t nvox,
                            float* dW, float* db, float* dFn, void* workspace, size_t workspace_bytes,
                            bfm_stream_t stream) {
    return head_bwd_launch(dRaw, 0, Fn, head_w, n_out, C, nvox, dW, db, dFn, workspace, workspace_bytes, stream);
}

extern "C" int bfm_head_bwd_rows(const float* dRaw_rows, int64_t row_stride, const float* Fn, const float* head_w, int n_out,
                                 int C, int64_t nvox, float* dW, float* db, float* dFn, void* workspace,
                                 size_t workspace_bytes, bfm_stream_t stream) {
    if (row_stride < nvox) return BFM_E_ARG;
    return head_bwd_launch(dRaw_rows, row_stride, Fn, head_w, n_out, C, nvox, dW, db, dFn, workspace, workspace_bytes, stream);
}

// ---- rows layout of the head outputs: same arguments as the channels-last entries with (raw, n_out) -> (rows, pitch)
extern "C" int bfm_loss_l1_multi_rows(const float* raw_rows, int64_t row_stride, int n_out, int64_t nvox, int n,
                                      const int32_t* cols, const int32_t* l2, const float* clampv, const float* coef,
                                      const float* const* targets, const float* const* weights, const float* const* masks,
                                      float* dRaw_rows, double* loss_out, void* workspace, size_t workspace_bytes,
                                      bfm_stream_t stream) {
    if (!raw_rows || nvox <= 0 || row_stride < nvox || n_out <= 0 || n <= 0 || n > L1M_MAX || !cols || !l2 || !clampv ||
        !coef || !targets || !loss_out || !workspace)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_l1_multi_workspace()) return BFM_E_WORKSPACE;
    L1Table tab{};
    tab.n = n;
    for (int k = 0; k < n; ++k) {
        if (cols[k] < 0 || cols[k] >= n_out || !targets[k]) return BFM_E_ARG;
        tab.e[k] = L1Entry{cols[k], l2[k], clampv[k], coef[k] / (float)nvox, targets[k], weights ? weights[k] : nullptr,
                           masks ? masks[k] : nullptr};
    }
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipStream_t st = bfm_s(stream);
    hipLaunchKernelGGL(l1_multi_rows_kernel, dim3(nb), dim3(256), 0, st, raw_rows, row_stride, nvox, tab, dRaw_rows, part);
    hipLaunchKernelGGL(l1_multi_fold_kernel, dim3(n), dim3(64), 0, st, part, nb, n, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

extern "C" int bfm_loss_grad_l1_multi_rows(const float* raw_rows, int64_t row_stride, int n_out, int n, const int32_t* cols,
                                           const float* coef, const float* const* targets, const float* const* weights,
                                           int D, int H, int W, float* dRaw_rows, double* loss_out, void* workspace,
                                           size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw_rows || !cols || !coef || !targets || !loss_out || !workspace || n <= 0 || n > L1M_MAX || D <= 0 || H <= 0 ||
        W <= 0 || n_out <= 0)
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_l1_multi_workspace()) return BFM_E_WORKSPACE;
    const int64_t nvox = (int64_t)D * H * W;
    if (row_stride < nvox) return BFM_E_ARG;
    GL1Table tab{};
    tab.n = n;
    for (int k = 0; k < n; ++k) {
        if (cols[k] < 0 || cols[k] >= n_out || !targets[k]) return BFM_E_ARG;
        for (int j = 0; j < k; ++j)
            if (cols[j] == cols[k]) return BFM_E_ARG;               // two entries on one column would race on dRaw
        tab.e[k] = GL1Entry{cols[k], coef[k] / (float)nvox, targets[k], weights ? weights[k] : nullptr};
    }
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipLaunchKernelGGL(grad_l1_multi_rows_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), raw_rows, row_stride, tab, D, H, W,
                       dRaw_rows, part);
    hipLaunchKernelGGL(l1_multi_fold_kernel, dim3(n), dim3(64), 0, bfm_s(stream), part, nb, n, 1.0 / (double)nvox, loss_out);
    return bfm_launch_status();
}

// P = softmax probabilities as [ns] rows of nvox (scratch / output); ns <= 64
extern "C" int bfm_loss_seg_rows(const float* raw_rows, int64_t row_stride, int n_out, int c0, int ns, const float* target,
                                 const float* wce, const float* wdice, int64_t nvox, float coef_ce, float coef_dice,
                                 float* P, float* dRaw_rows, double* loss_out, void* workspace, size_t workspace_bytes,
                                 bfm_stream_t stream) {
    if (!raw_rows || !target || !wce || !wdice || !P || !loss_out || !workspace || nvox <= 0 || row_stride < nvox ||
        ns <= 0 || c0 < 0 || c0 + ns > n_out)
        return BFM_E_ARG;
    if (ns > 64) return BFM_E_SHAPE;
    if (workspace_bytes < bfm_loss_workspace(ns)) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(nvox, RB);
    hipStream_t st = bfm_s(stream);
    const int64_t vpb = bfm_cdiv64(nvox, nb);
    if (ns <= 32)
        hipLaunchKernelGGL(seg_fwd_rows_kernel<32>, dim3(nb), dim3(256), 0, st, raw_rows, row_stride, c0, ns, target, wce, nvox, P, part);
    else
        hipLaunchKernelGGL(seg_fwd_rows_kernel<64>, dim3(nb), dim3(256), 0, st, raw_rows, row_stride, c0, ns, target, wce, nvox, P, part);
    // blocks past the last voxel chunk still write zeros: the fold below reads all nb rows
    hipLaunchKernelGGL(seg_class_sums_rows_kernel, dim3(nb, ns), dim3(256), 0, st, P, target, ns, nvox, vpb, part);
    double* sums = part + (size_t)RB * (1 + 2 * ns);             // [1 + 2 ns]
    hipLaunchKernelGGL(seg_fold_kernel, dim3(1 + 2 * ns), dim3(64), 0, st, part, nb, ns, sums);
    if (dRaw_rows) {
        if (ns <= 32)
            hipLaunchKernelGGL(seg_bwd_rows_kernel<32>, dim3(grid_for(nvox, 2048)), dim3(256), 0, st, P, target, ns, wce, wdice, sums,
                               nvox, coef_ce / (float)nvox, coef_dice, row_stride, c0, dRaw_rows);
        else
            hipLaunchKernelGGL(seg_bwd_rows_kernel<64>, dim3(grid_for(nvox, 2048)), dim3(256), 0, st, P, target, ns, wce, wdice, sums,
                               nvox, coef_ce / (float)nvox, coef_dice, row_stride, c0, dRaw_rows);
    }
    (void)hipMemcpyAsync(loss_out, sums, (size_t)(1 + 2 * ns) * sizeof(double), hipMemcpyDeviceToDevice, st);
    return bfm_launch_status();
}

extern "C" int bfm_normalize_bwd(const float* feat, const float* dFn, int C, int64_t nvox, float eps, float* dfeat,
                                 bfm_stream_t stream) {
    if (!feat || !dFn || !dfeat || C <= 0 || nvox <= 0) return BFM_E_ARG;
    hipStream_t st = bfm_s(stream);
    if (C == 64) hipLaunchKernelGGL(normalize_bwd_rows_kernel<16>, dim3(grid_for(nvox * 16)), dim3(256), 0, st, feat, dFn, nvox, eps, dfeat);
    else if (C == 32) hipLaunchKernelGGL(normalize_bwd_rows_kernel<8>, dim3(grid_for(nvox * 8)), dim3(256), 0, st, feat, dFn, nvox, eps, dfeat);
    else if (C == 16) hipLaunchKernelGGL(normalize_bwd_rows_kernel<4>, dim3(grid_for(nvox * 4)), dim3(256), 0, st, feat, dFn, nvox, eps, dfeat);
    else if (C == 8) hipLaunchKernelGGL(normalize_bwd_rows_kernel<2>, dim3(grid_for(nvox * 2)), dim3(256), 0, st, feat, dFn, nvox, eps, dfeat);
    else hipLaunchKernelGGL(normalize_bwd_kernel, dim3(grid_for(nvox)), dim3(256), 0, st, feat, dFn, C, nvox, eps, dfeat);
    return bfm_launch_status();
}

extern "C" int bfm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float grad_scale, bfm_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return BFM_E_ARG;
    const float bias1 = 1.f - powf(beta1, (float)step);
    const float bias2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bias1, bias2_sqrt, grad_scale);
    return bfm_launch_status();
}

// out[0] = sum of squares (fp64), nonfinite[0] |= 1 if any element is inf / nan
extern "C" int bfm_grad_sumsq(const float* g, int64_t n, double* out, int32_t* nonfinite, void* workspace,
                              size_t workspace_bytes, bfm_stream_t stream) {
    if (!g || !out || !nonfinite || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_workspace(0)) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(n, RB);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), g, n, part, nonfinite);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, bfm_s(stream), part, nb, 1.0, out);
    return bfm_launch_status();
}

// ---- multi-tensor forms: `tensors` (device, ntensors descriptors), chunk_tensor (device, tensor of every chunk of
// chunk_elems elements, tensors in order; first_chunk in the descriptor).  workspace: nchunks doubles.
extern "C" int bfm_grad_sumsq_multi(const bfm_adam_tensor_t* tensors, int ntensors, const int32_t* chunk_tensor, int nchunks,
                                    int64_t chunk_elems, double* sums_out, int32_t* nonfinite, void* workspace,
                                    size_t workspace_bytes, bfm_stream_t stream) {
    if (!tensors || !chunk_tensor || !sums_out || !nonfinite || !workspace || ntensors <= 0 || nchunks <= 0 ||
        chunk_elems <= 0 || (chunk_elems & 3))
        return BFM_E_ARG;
    if (workspace_bytes < (size_t)nchunks * sizeof(double)) return BFM_E_WORKSPACE;
    hipStream_t st = bfm_s(stream);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(nchunks), dim3(256), 0, st, tensors, chunk_tensor, chunk_elems, part, nonfinite);
    hipLaunchKernelGGL(sumsq_multi_fold_kernel, dim3(ntensors), dim3(64), 0, st, tensors, ntensors, chunk_elems, part, sums_out);
    return bfm_launch_status();
}

extern "C" int bfm_adamw_step_multi(const bfm_adam_tensor_t* tensors, int ntensors, const int32_t* chunk_tensor, int nchunks,
                                    int64_t chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    bfm_stream_t stream) {
    if (!tensors || !chunk_tensor || ntensors <= 0 || nchunks <= 0 || chunk_elems <= 0 || (chunk_elems & 3)) return BFM_E_ARG;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, bfm_s(stream), tensors, chunk_tensor, chunk_elems, lr, beta1,
                       beta2, eps, weight_decay);
    return bfm_launch_status();
}


extern "C" size_t bfm_loss_pathol_workspace(void) { return (size_t)(3 * RB + 8) * sizeof(double); }

extern "C" int bfm_loss_pathol(const float* raw, int64_t col_offset, int64_t voxel_stride, const float* target, int64_t nvox,
                               float coef_ce, float coef_dice, float* dRaw, double* loss_ce, double* loss_dice,
                               void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!raw || !target || !workspace || nvox <= 0 || col_offset < 0 || voxel_stride <= 0 || (!loss_ce && !loss_dice))
        return BFM_E_ARG;
    if (workspace_bytes < bfm_loss_pathol_workspace()) return BFM_E_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 7) return BFM_E_ARG;
    double* part = static_cast<double*>(workspace);
    double* sums = part + 3 * RB;
    const int nb = grid_for(nvox, RB);
    hipStream_t st = bfm_s(stream);
    hipLaunchKernelGGL(pathol_fwd_kernel, dim3(nb), dim3(256), 0, st, raw, col_offset, voxel_stride, target, nvox, part);
    hipLaunchKernelGGL(pathol_fold_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.0 / (double)nvox, sums, loss_ce, loss_dice);
    if (dRaw)
        hipLaunchKernelGGL(pathol_bwd_kernel, dim3(grid_for(nvox)), dim3(256), 0, st, raw, col_offset, voxel_stride, target, nvox,
                           sums, loss_ce ? coef_ce / (float)nvox : 0.f, loss_dice ? coef_dice : 0.f, dRaw);
    return bfm_launch_status();
}
