// GroupNorm statistics for channels-last-3D fp32 activations (HBM-bound).
//
// Replaces nn.GroupNorm's moment pass (Trainer/models/unet3d/buildingblocks.py:48-60)
// for the SingleConv input, including the decoder's virtual concatenation
// cat((skip, nearest_up(x))) (buildingblocks.py:265-276): the upsampled half is
// never materialised -- its moments are the low-res moments weighted by the
// per-voxel replication count of the nearest-neighbour map.
//
// Two launches, both deterministic (no atomics):
//   gn_partial : per-block, per-channel {sum, sum of squares} in fp64 and
//                {min, max} in fp32 over a contiguous voxel range;
//   gn_finalize: one block per group; fixed-order reduction of the partials,
//                mean / rstd, folded affine scale/shift per channel and the
//                group's bound on |x*scale+shift| (used by the MFMA conv to
//                choose its fp16 operand scale).
#include "bfm_common.h"

namespace {

constexpr int TPB = 256;

struct RepView {
    int d, h, w;                       // dims of the tensor being reduced
    const int32_t *rd, *rh, *rw;       // replication counts (nullptr -> weight 1)
};

__device__ __forceinline__ double vox_weight(const RepView& r, int64_t v) {
    if (r.rd == nullptr) return 1.0;
    int x = (int)(v % r.w);
    int64_t t = v / r.w;
    int y = (int)(t % r.h);
    int z = (int)(t / r.h);
    return (double)r.rd[z] * (double)r.rh[y] * (double)r.rw[x];
}

template <int VEC>
struct Acc {
    double s[VEC], q[VEC];
    float mn[VEC], mx[VEC];
    __device__ void init() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s[i] = 0.0; q[i] = 0.0; mn[i] = INFINITY; mx[i] = -INFINITY; }
    }
    __device__ void add(const float* x, double wgt) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            double xv = (double)x[i];
            s[i] += wgt * xv;
            q[i] += wgt * xv * xv;
            mn[i] = fminf(mn[i], x[i]);
            mx[i] = fmaxf(mx[i], x[i]);
        }
    }
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float* out) {
    if constexpr (VEC == 4) {
        float4 v = *reinterpret_cast<const float4*>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
        out[0] = *p;
    }
}

// partial tables: psum/psq [nblocks][C] double, pmin/pmax [nblocks][C] float
template <int VEC>
__global__ void __launch_bounds__(TPB) gn_partial(const float* __restrict__ X, int C, int64_t nvox,
                                                  int64_t vox_per_block, RepView rep, double* __restrict__ psum,
                                                  double* __restrict__ psq, float* __restrict__ pmin,
                                                  float* __restrict__ pmax) {
    extern __shared__ double smem_d[];
    const int t = threadIdx.x;
    {   // blockIdx.y = sample of a batch: tensors [S][nvox][C], partial tables [S][gridDim.x][C]
        const size_t so = (size_t)blockIdx.y * gridDim.x * C;
        X += (size_t)blockIdx.y * nvox * C;
        psum += so; psq += so; pmin += so; pmax += so;
    }
    const int CV = C / VEC;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block;
    const int64_t v1 = min(nvox, v0 + vox_per_block);
    const size_t row = (size_t)blockIdx.x * C;

    if (CV > TPB) {
        // wide tensors: each thread owns a column, walks every voxel of the range
        for (int cb = 0; cb < CV; cb += TPB) {
            int col = cb + t;
            if (col >= CV) continue;
            Acc<VEC> a; a.init();
            for (int64_t v = v0; v < v1; ++v) {
                float x[VEC];
                load_vec<VEC>(X + v * C + (size_t)col * VEC, x);
                a.add(x, vox_weight(rep, v));
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                psum[row + col * VEC + i] = a.s[i];
                psq[row + col * VEC + i] = a.q[i];
                pmin[row + col * VEC + i] = a.mn[i];
                pmax[row + col * VEC + i] = a.mx[i];
            }
        }
        return;
    }

    const int RP = TPB / CV;              // voxel rows in flight per pass
    const bool active = t < RP * CV;
    const int r0 = t / CV, col = t % CV;
    Acc<VEC> a; a.init();
    if (active) {
        int64_t v = v0 + r0;
        for (; v + 3 * (int64_t)RP < v1; v += 4 * (int64_t)RP) {      // 4 independent loads in flight
            float x0[VEC], x1[VEC], x2[VEC], x3[VEC];
            load_vec<VEC>(X + v * C + (size_t)col * VEC, x0);
            load_vec<VEC>(X + (v + RP) * C + (size_t)col * VEC, x1);
            load_vec<VEC>(X + (v + 2 * (int64_t)RP) * C + (size_t)col * VEC, x2);
            load_vec<VEC>(X + (v + 3 * (int64_t)RP) * C + (size_t)col * VEC, x3);
            a.add(x0, vox_weight(rep, v));
            a.add(x1, vox_weight(rep, v + RP));
            a.add(x2, vox_weight(rep, v + 2 * (int64_t)RP));
            a.add(x3, vox_weight(rep, v + 3 * (int64_t)RP));
        }
        for (; v < v1; v += RP) {
            float x[VEC];
            load_vec<VEC>(X + v * C + (size_t)col * VEC, x);
            a.add(x, vox_weight(rep, v));
        }
    }
    // reduce the RP rows in fixed order through LDS
    double* ls = smem_d;                       // [RP][C]
    double* lq = ls + (size_t)RP * C;          // [RP][C]
    float* lmn = reinterpret_cast<float*>(lq + (size_t)RP * C);
    float* lmx = lmn + (size_t)RP * C;
    if (active) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            int idx = r0 * C + col * VEC + i;
            ls[idx] = a.s[i]; lq[idx] = a.q[i]; lmn[idx] = a.mn[i]; lmx[idx] = a.mx[i];
        }
    }
    __syncthreads();
    // second stage: thread t folds RP/STEP rows, then a serial tail; keep it simple & ordered
    for (int c = t; c < C; c += TPB) {
        double s = 0.0, q = 0.0; float mn = INFINITY, mx = -INFINITY;
        for (int r = 0; r < RP; ++r) {
            s += ls[r * C + c]; q += lq[r * C + c];
            mn = fminf(mn, lmn[r * C + c]); mx = fmaxf(mx, lmx[r * C + c]);
        }
        psum[row + c] = s; psq[row + c] = q; pmin[row + c] = mn; pmax[row + c] = mx;
    }
}

struct PartTab {
    const double *psum, *psq;
    const float *pmin, *pmax;
    int nb, C;
    double wgt;                        // uniform replication weight of this source's voxels (1, or 8 for an exact 2x upsample)
};

// Moment rows written by a producer's epilogue (conv_mfma*, conv_stem_mfma): [nrows][C] of {sum, sumsq} (fp64) and
// {min, max} (fp32), one row per producer tile.  rows_reduce folds them to at most RR_MAX rows in fixed order so
// that gn_finalize stays a few microseconds; the activation itself is never re-read.
constexpr int RR_MAX = 128;

__global__ void __launch_bounds__(TPB) rows_reduce(const double* __restrict__ rsum, const double* __restrict__ rsq,
                                                   const float* __restrict__ rmn, const float* __restrict__ rmx,
                                                   int nrows, int C, int rows_per_block, double* __restrict__ psum,
                                                   double* __restrict__ psq, float* __restrict__ pmin,
                                                   float* __restrict__ pmax) {
    extern __shared__ double smem_d[];
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(nrows, r0 + rows_per_block);
    const int CP = C < TPB ? C : TPB;              // columns in flight
    const int RP = TPB / CP;                        // rows in flight
    double* ls = smem_d;
    double* lq = ls + TPB;
    float* lmn = reinterpret_cast<float*>(lq + TPB);
    float* lmx = lmn + TPB;
    for (int cb = 0; cb < C; cb += CP) {
        const int col = cb + t % CP, part = t / CP;
        double sv = 0.0, qv = 0.0;
        float mn = INFINITY, mx = -INFINITY;
        if (part < RP && col < C) {
            for (int r = r0 + part; r < r1; r += RP) {
                const size_t i = (size_t)r * C + col;
                sv += rsum[i]; qv += rsq[i];
                mn = fminf(mn, rmn[i]); mx = fmaxf(mx, rmx[i]);
            }
        }
        ls[t] = sv; lq[t] = qv; lmn[t] = mn; lmx[t] = mx;
        __syncthreads();
        if (part == 0 && col < C) {
            for (int pp = 1; pp < RP; ++pp) {
                const int i = pp * CP + (t % CP);
                sv += ls[i]; qv += lq[i];
                mn = fminf(mn, lmn[i]); mx = fmaxf(mx, lmx[i]);
            }
            const size_t o = (size_t)blockIdx.x * C + col;
            psum[o] = sv; psq[o] = qv; pmin[o] = mn; pmax[o] = mx;
        }
        __syncthreads();
    }
}

// From a group's channel totals (LDS: chan_*[cpg]) to its moments and the per-channel affine.  Called by every thread
// of the workgroup after the barrier that follows the writes of chan_*; red_mx: TPB / 64 floats of scratch.
__device__ __forceinline__ void group_affine(int g, int cpg, int c_first, double count_per_channel, float eps,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* __restrict__ scale, float* __restrict__ shift,
                                             float* __restrict__ bound, float* __restrict__ mean_out,
                                             float* __restrict__ rstd_out, const double* chan_s, const double* chan_q,
                                             const float* chan_mn, const float* chan_mx, float* red_mx) {
    const int t = threadIdx.x;
    // group moments: wave 0 folds the channel totals (strided, then an xor butterfly: fixed order)
    __shared__ float sh_mean, sh_rstd;
    if (t < 64) {
        double S = 0.0, Q = 0.0;
        for (int c = t; c < cpg; c += 64) { S += chan_s[c]; Q += chan_q[c]; }
        S = wave_reduce_sum(S);
        Q = wave_reduce_sum(Q);
        if (t == 0) {
            double n = count_per_channel * (double)cpg;
            double mean = S / n;
            double var = Q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            sh_mean = (float)mean;
            sh_rstd = (float)(1.0 / sqrt(var + (double)eps));
            if (mean_out) mean_out[g] = sh_mean;                   // kept for the backward pass (training)
            if (rstd_out) rstd_out[g] = sh_rstd;
        }
    }
    __syncthreads();
    const float mean = sh_mean, rstd = sh_rstd;
    float bmax = 0.f;
    for (int cg = t; cg < cpg; cg += TPB) {
        int c = c_first + cg;
        float sc = rstd * gamma[c];
        float sh = -sc * mean + beta[c];
        scale[c] = sc; shift[c] = sh;
        float b0 = fabsf(fmaf(chan_mn[cg], sc, sh));
        float b1 = fabsf(fmaf(chan_mx[cg], sc, sh));
        bmax = fmaxf(bmax, fmaxf(b0, b1));
    }
    bmax = wave_reduce_max(bmax);
    if ((t & 63) == 0) red_mx[t >> 6] = bmax;
    __syncthreads();
    if (t == 0) {
        float b = 0.f;
        for (int i = 0; i < TPB / 64; ++i) b = fmaxf(b, red_mx[i]);
        bound[g] = b;
    }
}

// rows_reduce + gn_finalize in ONE launch for tables of more than RR_MAX rows (the full-resolution levels: up to 16 000
// rows, 229 of these per 256^3 volume).  Grid (G, KS): workgroup (g, k) folds row slice k of BOTH sources for the
// channels of group g alone -- so what a group's finalize needs afterwards is KS partials per channel, not 128 rows of
// every channel -- writes them through to memory (agent-scope stores: eight XCDs, eight L2s), waits for the stores'
// acknowledgement, meets, and takes a ticket of ITS GROUP with a relaxed agent-scope add.  The workgroup that draws a
// group's last ticket reads the group's KS x cpg partials with agent-scope loads, sums them in slice order, and runs
// group_affine: a few hundred loads in one round, not the megabyte a single finalizing workgroup had to pull in the
// first attempt (profiles/r03_gn_one_launch_experiment.txt).  One release / acquire pair per group; every sum has a fixed
// order (rows of a slice: RP interleaved partial sums combined in order; slices in order), so results do not depend
// on which workgroup arrives last.  tickets[g] are zero on entry and are left zero.
constexpr int KS = 16;

template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct RowsSrc {
    const double *rsum, *rsq;
    const float *rmn, *rmx;
    int nrows, C;
    double wgt;
};

__global__ void __launch_bounds__(TPB) rows_group_finalize(RowsSrc sa, RowsSrc sb, int G, double count_per_channel, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ bound, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, double* __restrict__ psum,
                                                           double* __restrict__ psq, float* __restrict__ pmn,
                                                           float* __restrict__ pmx, int* __restrict__ tickets) {
    extern __shared__ double smem_d[];
    __shared__ int is_last;
    const int t = threadIdx.x;
    const int g = blockIdx.x, k = blockIdx.y;
    const int Ctot = sa.C + sb.C;
    const int cpg = Ctot / G;
    const int c_first = g * cpg;
    double* chan_s = smem_d;
    double* chan_q = chan_s + cpg;
    double* red_s = chan_q + cpg;
    double* red_q = red_s + TPB;
    float* chan_mn = reinterpret_cast<float*>(red_q + TPB);
    float* chan_mx = chan_mn + cpg;
    float* red_mn = chan_mx + cpg;
    float* red_mx = red_mn + TPB;

    const int CP = cpg < TPB ? cpg : TPB;           // columns in flight
    const int RP = TPB / CP;                         // rows in flight
    const bool active = t < RP * CP;
    const int cl = t % CP, part = t / CP;
    for (int cb = 0; cb < cpg; cb += CP) {
        const int cg = cb + cl;
        double sv = 0.0, qv = 0.0;
        float mn = INFINITY, mx = -INFINITY;
        if (active && cg < cpg) {
            const int c = c_first + cg;
            const RowsSrc& S = (c < sa.C) ? sa : sb;
            const int cc = (c < sa.C) ? c : c - sa.C;
            const int per = (S.nrows + KS - 1) / KS;
            const int r0 = k * per, r1 = min(S.nrows, r0 + per);
            for (int r = r0 + part; r < r1; r += RP) {
                const size_t i = (size_t)r * S.C + cc;
                sv += S.rsum[i]; qv += S.rsq[i];
                mn = fminf(mn, S.rmn[i]); mx = fmaxf(mx, S.rmx[i]);
            }
        }
        red_s[t] = sv; red_q[t] = qv; red_mn[t] = mn; red_mx[t] = mx;
        __syncthreads();
        if (active && part == 0 && cg < cpg) {
            for (int pp = 1; pp < RP; ++pp) {
                const int i = pp * CP + cl;
                sv += red_s[i]; qv += red_q[i];
                mn = fminf(mn, red_mn[i]); mx = fmaxf(mx, red_mx[i]);
            }
            const size_t o = ((size_t)g * KS + k) * cpg + cg;
            st_agent(psum + o, sv); st_agent(psq + o, qv); st_agent(pmn + o, mn); st_agent(pmx + o, mx);
        }
        __syncthreads();
    }
    // ---- publish, take this group's ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores are acknowledged
    __syncthreads();
    if (t == 0) {
        // release: the workgroup's partials (written through and acknowledged above) are ordered before the ticket in the
        // memory model too, not only by the vmcnt wait.  The counter must be 0 when a launch starts: every completed launch
        // leaves it there (the last arriver resets it), and the host entry point zeroes the tickets when a launch fails --
        // the kernel itself cannot heal a stale count (round 4's modulo did not: ADVICE r4), so that memset is the guard
        const int got = __hip_atomic_fetch_add(tickets + g, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        is_last = got == KS - 1;
        if (is_last) __hip_atomic_store(tickets + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all KS have arrived
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // pairs with the other workgroups' release adds
    // ---- the group's last arriver: channel totals = the KS partials in slice order (all loads of a thread in one round)
    for (int cg = t; cg < cpg; cg += TPB) {
        const int c = c_first + cg;
        const double w = (c < sa.C) ? sa.wgt : sb.wgt;
        double ps[KS], pq[KS];
        float pn[KS], px[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const size_t o = ((size_t)g * KS + kk) * cpg + cg;
            ps[kk] = ld_agent(psum + o); pq[kk] = ld_agent(psq + o); pn[kk] = ld_agent(pmn + o); px[kk] = ld_agent(pmx + o);
        }
        double sv = 0.0, qv = 0.0;
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { sv += ps[kk]; qv += pq[kk]; mn = fminf(mn, pn[kk]); mx = fmaxf(mx, px[kk]); }
        chan_s[cg] = sv * w; chan_q[cg] = qv * w; chan_mn[cg] = mn; chan_mx[cg] = mx;
    }
    __syncthreads();
    group_affine(g, cpg, c_first, count_per_channel, eps, gamma, beta, scale, shift, bound, mean_out, rstd_out, chan_s, chan_q,
                 chan_mn, chan_mx, red_mx);
}

// One block per group.  Dynamic LDS: chan_s[cpg], chan_q[cpg] (double), chan_mn[cpg], chan_mx[cpg] (float),
// red_s[TPB], red_q[TPB] (double), red_mn[TPB], red_mx[TPB] (float)
__global__ void __launch_bounds__(TPB) gn_finalize(PartTab ta, PartTab tb, int G, double count_per_channel,
                                                   float eps, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ scale,
                                                   float* __restrict__ shift, float* __restrict__ bound,
                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    extern __shared__ double smem_d[];
    const int t = threadIdx.x;
    const int g = blockIdx.x;
    const int Ctot = ta.C + tb.C;
    const int cpg = Ctot / G;
    const int c_first = g * cpg;
    if (blockIdx.y) {   // sample of a batch: tables [S][nb][C] inside each plane, outputs [S][Ctot] / [S][G]
        const size_t oa = (size_t)blockIdx.y * ta.nb * ta.C, ob = (size_t)blockIdx.y * tb.nb * tb.C;
        ta.psum += oa; ta.psq += oa; ta.pmin += oa; ta.pmax += oa;
        if (tb.C) { tb.psum += ob; tb.psq += ob; tb.pmin += ob; tb.pmax += ob; }
        scale += (size_t)blockIdx.y * Ctot; shift += (size_t)blockIdx.y * Ctot; bound += (size_t)blockIdx.y * G;
        if (mean_out) mean_out += (size_t)blockIdx.y * G;
        if (rstd_out) rstd_out += (size_t)blockIdx.y * G;
    }

    double* chan_s = smem_d;
    double* chan_q = chan_s + cpg;
    double* red_s = chan_q + cpg;
    double* red_q = red_s + TPB;
    float* chan_mn = reinterpret_cast<float*>(red_q + TPB);
    float* chan_mx = chan_mn + cpg;
    float* red_mn = chan_mx + cpg;
    float* red_mx = red_mn + TPB;

    const int cpgP = cpg < TPB ? cpg : TPB;
    const int P = TPB / cpgP;
    const bool active = t < P * cpgP;
    const int cl = t % cpgP, part = t / cpgP;

    for (int cb = 0; cb < cpg; cb += cpgP) {
        const int cg = cb + cl;                 // channel within group
        double s = 0.0, q = 0.0; float mn = INFINITY, mx = -INFINITY;
        if (active && cg < cpg) {
            int c = c_first + cg;
            const PartTab& T = (c < ta.C) ? ta : tb;
            int cc = (c < ta.C) ? c : c - ta.C;
            for (int b = part; b < T.nb; b += P) {
                size_t i = (size_t)b * T.C + cc;
                s += T.psum[i]; q += T.psq[i];
                mn = fminf(mn, T.pmin[i]); mx = fmaxf(mx, T.pmax[i]);
            }
            s *= T.wgt; q *= T.wgt;
        }
        red_s[t] = s; red_q[t] = q; red_mn[t] = mn; red_mx[t] = mx;
        __syncthreads();
        if (active && part == 0 && cg < cpg) {
            for (int p = 1; p < P; ++p) {
                int i = p * cpgP + cl;
                s += red_s[i]; q += red_q[i];
                mn = fminf(mn, red_mn[i]); mx = fmaxf(mx, red_mx[i]);
            }
            chan_s[cg] = s; chan_q[cg] = q; chan_mn[cg] = mn; chan_mx[cg] = mx;
        }
        __syncthreads();
    }

    group_affine(g, cpg, c_first, count_per_channel, eps, gamma, beta, scale, shift, bound, mean_out, rstd_out, chan_s, chan_q,
                 chan_mn, chan_mx, red_mx);
}

struct Plan {
    int nbA, nbB;
    int64_t vpbA, vpbB;
    size_t offA_sum, offA_sq, offA_mn, offA_mx, offB_sum, offB_sq, offB_mn, offB_mx, total;
};

int blocks_for(int64_t nvox, int C, int64_t* vpb) {
    // enough blocks to stream from HBM (>= 8 waves/CU at 4 loads in flight each), few enough that the
    // G-block finalize stays a few microseconds: nb * C ~ 32K partial entries
    int64_t nb = 32768 / C;
    if (nb > 2048) nb = 2048;
    if (nb < 8) nb = 8;
    int64_t by_vox = bfm_cdiv64(nvox, 16);     // at least 16 voxels per block
    if (nb > by_vox) nb = by_vox;
    if (nb < 1) nb = 1;
    *vpb = bfm_cdiv64(nvox, nb);
    nb = bfm_cdiv64(nvox, *vpb);
    return (int)nb;
}

Plan make_plan(int CA, int CB, int64_t nvoxA, int64_t nvoxB, int S = 1) {
    Plan p{};
    p.nbA = blocks_for(nvoxA, CA, &p.vpbA);                 // blocks per sample: the same split as the one-sample call
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    p.offA_sum = take((size_t)S * p.nbA * CA * 8); p.offA_sq = take((size_t)S * p.nbA * CA * 8);
    p.offA_mn = take((size_t)S * p.nbA * CA * 4); p.offA_mx = take((size_t)S * p.nbA * CA * 4);
    if (CB > 0) {
        p.nbB = blocks_for(nvoxB, CB, &p.vpbB);
        p.offB_sum = take((size_t)S * p.nbB * CB * 8); p.offB_sq = take((size_t)S * p.nbB * CB * 8);
        p.offB_mn = take((size_t)S * p.nbB * CB * 4); p.offB_mx = take((size_t)S * p.nbB * CB * 4);
    }
    p.total = off;
    return p;
}

void launch_partial(const float* X, int C, int64_t nvox, int nb, int64_t vpb, RepView rep, char* ws, size_t o_sum,
                    size_t o_sq, size_t o_mn, size_t o_mx, hipStream_t st, int S = 1) {
    double* ps = reinterpret_cast<double*>(ws + o_sum);
    double* pq = reinterpret_cast<double*>(ws + o_sq);
    float* pn = reinterpret_cast<float*>(ws + o_mn);
    float* px = reinterpret_cast<float*>(ws + o_mx);
    bool vec4 = (C % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    int CV = vec4 ? C / 4 : C;
    int RP = CV > TPB ? 0 : TPB / CV;
    size_t smem = (size_t)RP * C * 24;
    if (vec4) hipLaunchKernelGGL(gn_partial<4>, dim3(nb, S), dim3(TPB), smem, st, X, C, nvox, vpb, rep, ps, pq, pn, px);
    else hipLaunchKernelGGL(gn_partial<1>, dim3(nb, S), dim3(TPB), smem, st, X, C, nvox, vpb, rep, ps, pq, pn, px);
}

}  // namespace

extern "C" size_t bfm_gn_stats_workspace(int CA, int CB, int D, int H, int W, const bfm_upsample_t* up) {
    int64_t nvoxA = (int64_t)D * H * W;
    int64_t nvoxB = (CB > 0 && up) ? (int64_t)up->d * up->h * up->w : 0;
    return make_plan(CA, CB, nvoxA, nvoxB).total;
}

static int gn_stats_launch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                           const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps, float* scale,
                           float* shift, float* bound, float* mean_out, float* rstd_out, void* workspace,
                           size_t workspace_bytes, bfm_stream_t stream);

extern "C" int bfm_gn_stats_train(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                  const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps,
                                  float* scale, float* shift, float* bound, float* mean_out, float* rstd_out,
                                  void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    return gn_stats_launch(A, CA, B, CB, 1, D, H, W, up, gamma, beta, G, eps, scale, shift, bound, mean_out, rstd_out,
                           workspace, workspace_bytes, stream);
}

extern "C" size_t bfm_gn_stats_batch_workspace(int CA, int CB, int S, int D, int H, int W, const bfm_upsample_t* up) {
    int64_t nvoxA = (int64_t)D * H * W;
    int64_t nvoxB = (CB > 0 && up) ? (int64_t)up->d * up->h * up->w : 0;
    return make_plan(CA, CB, nvoxA, nvoxB, S < 1 ? 1 : S).total;
}

extern "C" int bfm_gn_stats_batch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                                  const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps,
                                  float* scale, float* shift, float* bound, void* workspace, size_t workspace_bytes,
                                  bfm_stream_t stream) {
    if (S < 1 || S > 65535) return BFM_E_ARG;
    return gn_stats_launch(A, CA, B, CB, S, D, H, W, up, gamma, beta, G, eps, scale, shift, bound, nullptr, nullptr,
                           workspace, workspace_bytes, stream);
}

static int gn_stats_launch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                           const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps, float* scale,
                           float* shift, float* bound, float* mean_out, float* rstd_out, void* workspace,
                           size_t workspace_bytes, bfm_stream_t stream) {
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !gamma || !beta || !scale || !shift || !bound || !workspace)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || up->d <= 0 || up->h <= 0 || up->w <= 0 || !up->repD || !up->repH ||
                              !up->repW)))
        return BFM_E_ARG;
    const int Ctot = CA + CB;
    if (G <= 0 || Ctot % G != 0) return BFM_E_SHAPE;
    const int cpg = Ctot / G;
    // LDS of gn_partial: RP*C*24 bytes must fit 64 KiB
    auto partial_ok = [](int C) {
        int CV = (C % 4 == 0) ? C / 4 : C;
        int RP = CV > TPB ? 0 : TPB / CV;
        return (size_t)RP * C * 24 <= 64 * 1024;
    };
    if (!partial_ok(CA) || (CB > 0 && !partial_ok(CB))) return BFM_E_SHAPE;
    size_t fin_smem = (size_t)cpg * 24 + (size_t)TPB * 24;
    if (fin_smem > 64 * 1024) return BFM_E_SHAPE;

    const int64_t nvoxA = (int64_t)D * H * W;
    const int64_t nvoxB = CB > 0 ? (int64_t)up->d * up->h * up->w : 0;
    Plan p = make_plan(CA, CB, nvoxA, nvoxB, S);
    if (workspace_bytes < p.total) return BFM_E_WORKSPACE;
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = bfm_s(stream);

    RepView none{D, H, W, nullptr, nullptr, nullptr};
    launch_partial(A, CA, nvoxA, p.nbA, p.vpbA, none, ws, p.offA_sum, p.offA_sq, p.offA_mn, p.offA_mx, st, S);
    PartTab ta{reinterpret_cast<double*>(ws + p.offA_sum), reinterpret_cast<double*>(ws + p.offA_sq),
               reinterpret_cast<float*>(ws + p.offA_mn), reinterpret_cast<float*>(ws + p.offA_mx), p.nbA, CA, 1.0};
    PartTab tb{nullptr, nullptr, nullptr, nullptr, 0, 0, 1.0};
    if (CB > 0) {
        RepView rep{up->d, up->h, up->w, up->repD, up->repH, up->repW};
        launch_partial(B, CB, nvoxB, p.nbB, p.vpbB, rep, ws, p.offB_sum, p.offB_sq, p.offB_mn, p.offB_mx, st, S);
        tb = PartTab{reinterpret_cast<double*>(ws + p.offB_sum), reinterpret_cast<double*>(ws + p.offB_sq),
                     reinterpret_cast<float*>(ws + p.offB_mn), reinterpret_cast<float*>(ws + p.offB_mx), p.nbB, CB, 1.0};
    }
    hipLaunchKernelGGL(gn_finalize, dim3(G, S), dim3(TPB), fin_smem, st, ta, tb, G, (double)nvoxA, eps, gamma, beta,
                       scale, shift, bound, mean_out, rstd_out);
    return bfm_launch_status();
}

extern "C" int bfm_gn_stats(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                            const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps,
                            float* scale, float* shift, float* bound, void* workspace, size_t workspace_bytes,
                            bfm_stream_t stream) {
    return bfm_gn_stats_train(A, CA, B, CB, D, H, W, up, gamma, beta, G, eps, scale, shift, bound, nullptr, nullptr,
                              workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Statistics from producer-written moment rows (no pass over the activation).
namespace {

struct RowsView {
    const double *sum, *sq;
    const float *mn, *mx;
};

// rows: a table of `total` rows (four planes of total * C entries); the view starts at its row `first`
RowsView rows_view(const void* rows, int total, int C, int first = 0) {
    const char* b = static_cast<const char*>(rows);
    const size_t n = (size_t)total * C, o = (size_t)first * C;
    return RowsView{reinterpret_cast<const double*>(b) + o, reinterpret_cast<const double*>(b + n * 8) + o,
                    reinterpret_cast<const float*>(b + n * 16) + o, reinterpret_cast<const float*>(b + n * 20) + o};
}

size_t rows_ws_bytes(int nrows, int C) {            // reduced table (only when nrows > RR_MAX)
    if (nrows <= RR_MAX) return 0;
    return (((size_t)RR_MAX * C * 24) + 255) & ~(size_t)255;
}

// returns the table gn_finalize should read; launches rows_reduce into ws when the row count is large
PartTab rows_source(const void* rows, int nrows, int C, double wgt, char* ws, hipStream_t st, int total = 0, int first = 0) {
    RowsView v = rows_view(rows, total > 0 ? total : nrows, C, first);
    if (nrows <= RR_MAX) return PartTab{v.sum, v.sq, v.mn, v.mx, nrows, C, wgt};
    const int rpb = bfm_cdiv(nrows, RR_MAX);
    const int nb = bfm_cdiv(nrows, rpb);
    const size_t n = (size_t)RR_MAX * C;
    double* ps = reinterpret_cast<double*>(ws);
    double* pq = reinterpret_cast<double*>(ws + n * 8);
    float* pn = reinterpret_cast<float*>(ws + n * 16);
    float* px = reinterpret_cast<float*>(ws + n * 20);
    hipLaunchKernelGGL(rows_reduce, dim3(nb), dim3(TPB), (size_t)TPB * 24, st, v.sum, v.sq, v.mn, v.mx, nrows, C, rpb,
                       ps, pq, pn, px);
    return PartTab{ps, pq, pn, px, nb, C, wgt};
}

}  // namespace

extern "C" size_t bfm_moment_rows_bytes(int nrows, int C) {
    if (nrows <= 0 || C <= 0) return 0;
    return (size_t)nrows * C * 24;
}

extern "C" size_t bfm_gn_stats_rows_workspace(int nrowsA, int CA, int nrowsB, int CB) {
    return rows_ws_bytes(nrowsA, CA) + (CB > 0 ? rows_ws_bytes(nrowsB, CB) : 0) + 256;
}

static int gn_rows_sliced(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB,
                          double weightB, int64_t nvox, const float* gamma, const float* beta, int G,
                          float eps, float* scale, float* shift, float* bound, float* mean_out,
                          float* rstd_out, void* workspace, size_t workspace_bytes, void* ticket,
                          bfm_stream_t stream, int totalA, int firstA, int totalB, int firstB);

extern "C" int bfm_gn_stats_rows_train(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB,
                                       double weightB, int64_t nvox, const float* gamma, const float* beta, int G,
                                       float eps, float* scale, float* shift, float* bound, float* mean_out,
                                       float* rstd_out, void* workspace, size_t workspace_bytes, void* ticket,
                                       bfm_stream_t stream) {
    return gn_rows_sliced(rowsA, nrowsA, CA, rowsB, nrowsB, CB, weightB, nvox, gamma, beta, G, eps, scale, shift, bound,
                          mean_out, rstd_out, workspace, workspace_bytes, ticket, stream, nrowsA, 0, nrowsB, 0);
}

// The same with each table given as rows [first, first + nrows) of a larger one of `total` rows: one sample's rows of a
// batched producer (bfm_conv3x3x3_mfma_batch writes [S * nrows][C] planes), read where they lie.
extern "C" int bfm_gn_stats_rows_sliced(const void* rowsA, int totalA, int firstA, int nrowsA, int CA, const void* rowsB,
                                        int totalB, int firstB, int nrowsB, int CB, double weightB, int64_t nvox,
                                        const float* gamma, const float* beta, int G, float eps, float* scale,
                                        float* shift, float* bound, void* workspace, size_t workspace_bytes, void* ticket,
                                        bfm_stream_t stream) {
    if (firstA < 0 || nrowsA <= 0 || firstA + nrowsA > totalA) return BFM_E_ARG;
    if (CB > 0 && (firstB < 0 || nrowsB <= 0 || firstB + nrowsB > totalB)) return BFM_E_ARG;
    return gn_rows_sliced(rowsA, nrowsA, CA, rowsB, nrowsB, CB, weightB, nvox, gamma, beta, G, eps, scale, shift, bound,
                          nullptr, nullptr, workspace, workspace_bytes, ticket, stream, totalA, firstA, CB > 0 ? totalB : 0,
                          CB > 0 ? firstB : 0);
}

static int gn_rows_sliced(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB,
                          double weightB, int64_t nvox, const float* gamma, const float* beta, int G,
                          float eps, float* scale, float* shift, float* bound, float* mean_out,
                          float* rstd_out, void* workspace, size_t workspace_bytes, void* ticket,
                          bfm_stream_t stream, int totalA, int firstA, int totalB, int firstB) {
    if (!rowsA || nrowsA <= 0 || CA <= 0 || nvox <= 0 || !gamma || !beta || !scale || !shift || !bound) return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!rowsB || nrowsB <= 0 || !(weightB > 0.0)))) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(rowsA) & 7) || (CB > 0 && (reinterpret_cast<uintptr_t>(rowsB) & 7))) return BFM_E_ARG;
    const int Ctot = CA + CB;
    if (G <= 0 || Ctot % G != 0) return BFM_E_SHAPE;
    const int cpg = Ctot / G;
    const size_t fin_smem = (size_t)cpg * 24 + (size_t)TPB * 24;
    if (fin_smem > 64 * 1024) return BFM_E_SHAPE;
    const size_t needA = rows_ws_bytes(nrowsA, CA), needB = CB > 0 ? rows_ws_bytes(nrowsB, CB) : 0;
    if (needA + needB > 0 && (!workspace || workspace_bytes < needA + needB)) return BFM_E_WORKSPACE;
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 7)) return BFM_E_ARG;
    hipStream_t st = bfm_s(stream);
    char* ws = static_cast<char*>(workspace);
    if (ticket && (reinterpret_cast<uintptr_t>(ticket) & 3)) return BFM_E_ARG;
    // one launch (rows_group_finalize) when a table is large, the caller gave G zeroed tickets and the KS x Ctot
    // partials fit the workspace the two-launch form would have used
    if (ticket && needA + needB > 0 && G <= BFM_GN_TICKETS && (size_t)KS * Ctot * 24 <= needA + needB &&
        (size_t)cpg * 24 + (size_t)TPB * 24 <= 64 * 1024) {
        RowsView va = rows_view(rowsA, totalA, CA, firstA);
        RowsSrc sa{va.sum, va.sq, va.mn, va.mx, nrowsA, CA, 1.0};
        RowsSrc sb{nullptr, nullptr, nullptr, nullptr, 0, 0, 1.0};
        if (CB > 0) {
            RowsView vb = rows_view(rowsB, totalB, CB, firstB);
            sb = RowsSrc{vb.sum, vb.sq, vb.mn, vb.mx, nrowsB, CB, weightB};
        }
        const size_t n = (size_t)KS * Ctot;
        hipLaunchKernelGGL(rows_group_finalize, dim3(G, KS), dim3(TPB), fin_smem, st, sa, sb, G, (double)nvox, eps, gamma, beta,
                           scale, shift, bound, mean_out, rstd_out, reinterpret_cast<double*>(ws),
                           reinterpret_cast<double*>(ws + n * 8), reinterpret_cast<float*>(ws + n * 16),
                           reinterpret_cast<float*>(ws + n * 20), static_cast<int*>(ticket));
        const int rc = bfm_launch_status();
        if (rc != BFM_OK) (void)hipMemsetAsync(ticket, 0, (size_t)G * sizeof(int), st);    // never leave a partial count behind
        return rc;
    }
    PartTab ta = rows_source(rowsA, nrowsA, CA, 1.0, ws, st, totalA, firstA);
    PartTab tb{nullptr, nullptr, nullptr, nullptr, 0, 0, 1.0};
    if (CB > 0) tb = rows_source(rowsB, nrowsB, CB, weightB, ws + needA, st, totalB, firstB);
    hipLaunchKernelGGL(gn_finalize, dim3(G), dim3(TPB), fin_smem, st, ta, tb, G, (double)nvox, eps, gamma, beta, scale,
                       shift, bound, mean_out, rstd_out);
    return bfm_launch_status();
}

// S same-shape samples: rowsA / rowsB are the tables of batched producers ([S * nrows][C] per plane), every output [S][..]
extern "C" int bfm_gn_stats_rows_batch(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB,
                                       double weightB, int64_t nvox, int S, const float* gamma, const float* beta, int G,
                                       float eps, float* scale, float* shift, float* bound, bfm_stream_t stream) {
    if (!rowsA || nrowsA <= 0 || CA <= 0 || nvox <= 0 || S < 1 || S > 65535 || !gamma || !beta || !scale || !shift || !bound)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!rowsB || nrowsB <= 0 || !(weightB > 0.0)))) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(rowsA) & 7) || (CB > 0 && (reinterpret_cast<uintptr_t>(rowsB) & 7))) return BFM_E_ARG;
    if (nrowsA > RR_MAX || (CB > 0 && nrowsB > RR_MAX)) return BFM_E_SHAPE;     // per-sample tables small enough to finalize directly
    const int Ctot = CA + CB;
    if (G <= 0 || Ctot % G != 0) return BFM_E_SHAPE;
    const int cpg = Ctot / G;
    const size_t fin_smem = (size_t)cpg * 24 + (size_t)TPB * 24;
    if (fin_smem > 64 * 1024) return BFM_E_SHAPE;
    RowsView va = rows_view(rowsA, S * nrowsA, CA);
    PartTab ta{va.sum, va.sq, va.mn, va.mx, nrowsA, CA, 1.0};
    PartTab tb{nullptr, nullptr, nullptr, nullptr, 0, 0, 1.0};
    if (CB > 0) {
        RowsView vb = rows_view(rowsB, S * nrowsB, CB);
        tb = PartTab{vb.sum, vb.sq, vb.mn, vb.mx, nrowsB, CB, weightB};
    }
    hipLaunchKernelGGL(gn_finalize, dim3(G, S), dim3(TPB), fin_smem, bfm_s(stream), ta, tb, G, (double)nvox, eps, gamma,
                       beta, scale, shift, bound, (float*)nullptr, (float*)nullptr);
    return bfm_launch_status();
}

extern "C" int bfm_gn_stats_rows(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB,
                                 double weightB, int64_t nvox, const float* gamma, const float* beta, int G, float eps,
                                 float* scale, float* shift, float* bound, void* workspace, size_t workspace_bytes,
                                 void* ticket, bfm_stream_t stream) {
    return bfm_gn_stats_rows_train(rowsA, nrowsA, CA, rowsB, nrowsB, CB, weightB, nvox, gamma, beta, G, eps, scale, shift,
                                   bound, nullptr, nullptr, workspace, workspace_bytes, ticket, stream);
}
