// The generator item (BrainIDGen.__getitem__, Generator/datasets.py:700-757) without host round trips: kernels that
// work on case volumes RESIDENT in HBM (the reference re-reads NIfTI files and crops on the host for every item), take
// their scalar operands (min / max / sums / order statistics) from device memory, and fuse the short chains between two
// reductions.  Same arithmetic, same order, -ffp-contract=off: results equal the unfused chains bit for bit.
//
//   randn_philox          : torch.randn of the generator (own Philox4x32-10 + Box-Muller stream)
//   deform_minmax / write : BaseGen.deform_grid with myzoom_torch(Fsmall) folded in  datasets.py:187-303, utils.py:200-257
//   gather_targets        : read_and_deform(+_image/_distance/_registration)         utils.py:296-322,331-345,376-400,462-473
//   gather_onehot         : read_and_deform_segmentation                              utils.py:402-425
//   percentile_f64        : np.percentile(noise, q) on the device                     ShapeID/perlin3d.py:84-90
//   shape_from_noise      : threshold + binarize + sum                                perlin3d.py:84-90, utils.py:65-72
//   pathology_mask/encode : generate_sample / encode_pathology                        datasets.py:388-404,496-518
//   interp3d_linear_axes  : resample_resolution's meshgrid sample                     utils.py:591-609
//   sample_finalize       : I/max, SR residual                                        datasets.py:340-352
#include "bfm_common.h"

namespace {

inline int grid_for(int64_t n, int tpb = 256, int cap = 8192) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float ld_tex(const float* p) {          // past the per-CU L1 (HISTORY.md section 3.3)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t ld_tex(const int32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The two z-neighbours of a trilinear sample sit side by side in memory: ONE 8-byte load (aux 16 = sc1, agent scope) through
// a buffer descriptor whose range check makes the one-past-the-end read of a clamped upper corner harmless (it returns 0 and
// the value is not used).  Halves the L2 requests of the gathers (round 4: interp_linear 37 -> 20 us at 160^3).  Round 5 had
// gone back to two 4-byte loads because the 8-byte forms "returned zeros beside conv_wino4d"; round 6 found that the zeros were
// corner WEIGHTS lost by a packed-FP32 multiply, not texels (profiles/r06_hazard_root_cause.txt), and the library is built
// without packed-FP32 instructions since -- so the fast form is back.
typedef int v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tex_rsrc(const float* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void ld_tex2(__amdgpu_buffer_rsrc_t r, int64_t elem, bool second_distinct, float& a, float& b) {
    const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)(elem << 2), 0, 16);
    a = __int_as_float(v.x);
    b = second_distinct ? __int_as_float(v.y) : a;
}

__device__ __forceinline__ float nan_to_num(float x) {
    return x != x ? 0.f : (x == INFINITY ? 3.402823466e+38f : (x == -INFINITY ? -3.402823466e+38f : x));
}

// ------------------------------------------------------------------ Philox4x32-10 normals
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4(uint64_t ctr, uint64_t offset, uint64_t seed, uint32_t (&c)[4]) {
    c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = (uint32_t)offset; c[3] = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 8) + 0.5f) * 5.9604644775390625e-8f;      // (0, 1)
    const float u2 = ((float)(b >> 8) + 0.5f) * 5.9604644775390625e-8f;
    // hardware log2 / sqrt / sin / cos (the arguments are in (0, 1): no range reduction needed; v_sin / v_cos take turns)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // -2 ln u = -2 ln2 log2 u
    z0 = r * __builtin_amdgcn_cosf(u2); z1 = r * __builtin_amdgcn_sinf(u2);
}

__global__ void randn_philox(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset, float scale) {
    const int64_t n4 = (n + 3) >> 2;
    GRID_STRIDE(q, n4) {
        uint32_t c[4];
        philox4((uint64_t)q, offset, seed, c);
        float z[4];
        box_muller(c[0], c[1], z[0], z[1]);
        box_muller(c[2], c[3], z[2], z[3]);
        const int64_t i = q << 2;
        if (i + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            *reinterpret_cast<float4*>(out + i) = make_float4(z[0] * scale, z[1] * scale, z[2] * scale, z[3] * scale);
        } else {
            for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = z[k] * scale;
        }
    }
}

// ------------------------------------------------------------------ deformation grid with the zoom of Fsmall folded in
struct ZoomTabs {
    const int32_t *fx, *cx, *fy, *cy, *fz, *cz;
    const float *wfx, *wcx, *wfy, *wcy, *wfz, *wcz;
};
struct DefP { float a[9]; float c[3]; int shp[3]; float lo[3]; };

// F(x,y,z,ch) of myzoom_torch(Fsmall, size / small) -- the expression of zoom_linear (synth_interp.hip), same order
__device__ __forceinline__ void zoomed_F(const float* __restrict__ Fs, int ny, int nz, const ZoomTabs& t, int ii, int j,
                                         int k, float (&f)[3]) {
    const int fx = t.fx[ii], cx = t.cx[ii], fy = t.fy[j], cy = t.cy[j], fz = t.fz[k], cz = t.cz[k];
    const float wfx = t.wfx[ii], wcx = t.wcx[ii], wfy = t.wfy[j], wcy = t.wcy[j], wfz = t.wfz[k], wcz = t.wcz[k];
    const int64_t sx = (int64_t)ny * nz * 3, sy = (int64_t)nz * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto at = [&](int a, int b, int d) { return Fs[a * sx + b * sy + (int64_t)d * 3 + c]; };
        const float a00 = wfx * at(fx, fy, fz) + wcx * at(cx, fy, fz);
        const float a10 = wfx * at(fx, cy, fz) + wcx * at(cx, cy, fz);
        const float a01 = wfx * at(fx, fy, cz) + wcx * at(cx, fy, cz);
        const float a11 = wfx * at(fx, cy, cz) + wcx * at(cx, cy, cz);
        const float b0 = wfy * a00 + wcy * a10;
        const float b1 = wfy * a01 + wcy * a11;
        f[c] = wfz * b0 + wcz * b1;
    }
}

// MODE 0: block min / max partials of the clamped coordinates only (nothing else is written);
// MODE 1: the coordinates minus P.lo (and F when Fout is given)
template <int MODE>
__global__ void deform_zoom(const float* __restrict__ Fs, int fnx, int fny, int fnz, ZoomTabs t, int photo_zero_y,
                            int sx, int sy, int sz, DefP P, float* __restrict__ xx, float* __restrict__ yy,
                            float* __restrict__ zz, float* __restrict__ Fout, float* __restrict__ part) {
    const int64_t n = (int64_t)sx * sy * sz;
    const float cx = (float)((sx - 1) / 2.0), cy = (float)((sy - 1) / 2.0), cz = (float)((sz - 1) / 2.0);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    GRID_STRIDE(i, n) {
        const int z = (int)(i % sz);
        const int y = (int)((i / sz) % sy);
        const int x = (int)(i / ((int64_t)sy * sz));
        float x1 = (float)x - cx, y1 = (float)y - cy, z1 = (float)z - cz;
        if (Fs) {
            float f[3];
            zoomed_F(Fs, fny, fnz, t, x, y, z, f);
            if (photo_zero_y) f[1] = 0.f;
            x1 = x1 + f[0]; y1 = y1 + f[1]; z1 = z1 + f[2];
            if (MODE == 1 && Fout) { Fout[i * 3 + 0] = f[0]; Fout[i * 3 + 1] = f[1]; Fout[i * 3 + 2] = f[2]; }
        }
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = ((P.a[k * 3 + 0] * x1 + P.a[k * 3 + 1] * y1) + P.a[k * 3 + 2] * z1) + P.c[k];
            v = v < 0.f ? 0.f : v;
            const float hi = (float)(P.shp[k] - 1);
            v = v > hi ? hi : v;
            r[k] = v;
            if (MODE == 0) { mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
        }
        if (MODE == 1) { xx[i] = r[0] - P.lo[0]; yy[i] = r[1] - P.lo[1]; zz[i] = r[2] - P.lo[2]; }
    }
    if (MODE == 0) {
        __shared__ float red[6][4];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = wave_reduce_min(mn[k]), b = wave_reduce_max(mx[k]);
            if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = a; red[3 + k][threadIdx.x >> 6] = b; }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const int k = threadIdx.x;
            float v = red[k][0];
            for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
            part[(size_t)blockIdx.x * 6 + k] = v;
        }
    }
}

// 6 waves: wave k folds component k of the block partials (min for k < 3, max otherwise)
__global__ void minmax6_fold(const float* __restrict__ part, int nb, float* __restrict__ out) {
    const int k = threadIdx.x >> 6, l = threadIdx.x & 63;
    float v = k < 3 ? INFINITY : -INFINITY;
    for (int b = l; b < nb; b += 64) v = k < 3 ? fminf(v, part[(size_t)b * 6 + k]) : fmaxf(v, part[(size_t)b * 6 + k]);
    v = k < 3 ? wave_reduce_min(v) : wave_reduce_max(v);
    if (l == 0) out[k] = v;
}

// ------------------------------------------------------------------ targets: multi-volume trilinear gather
constexpr int GJ_MAX = BFM_GATHER_MAX_JOBS;
struct GJob {
    const float* src; float* out;
    float mean, scale, post_div, clo, chi, sign;
    int pre, defmax, clamp, stat;
};
struct GJobs { GJob j[GJ_MAX]; int n; };
struct Box { int x1, y1, z1, cnx, cny, cnz; };          // crop origin and crop dims inside the full volume

__device__ __forceinline__ float pre_op(float v, const GJob& J) {
    if (J.pre >= 1) v = nan_to_num(v);
    if (J.pre >= 2) v = (v - J.mean) / J.scale;
    return v;
}

// max over the crop box of nan_to_num(x) per job with defmax (blockIdx.y = job): partial per block
__global__ void box_max_partial(GJobs J, int ny, int nz, Box B, float* __restrict__ part) {
    const GJob& jb = J.j[blockIdx.y];
    float m = -INFINITY;
    if (jb.defmax) {
        const int64_t rows = (int64_t)B.cnx * B.cny;
        const int lane = threadIdx.x & 63;
        for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
            const int x = (int)(r / B.cny), y = (int)(r - (int64_t)x * B.cny);
            const float* s = jb.src + ((int64_t)(B.x1 + x) * ny + (B.y1 + y)) * nz + B.z1;
            for (int z = lane; z < B.cnz; z += 64) m = fmaxf(m, nan_to_num(s[z]));
        }
    }
    __shared__ float red[4];
    m = wave_reduce_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one wave per job: the crop's maximum AFTER (x - mean) / scale (monotone, so it is the transform of the maximum)
__global__ void box_max_fold(GJobs J, const float* __restrict__ part, int nb, double* __restrict__ scal) {
    const int j = blockIdx.x, l = threadIdx.x;
    if (!J.j[j].defmax) return;
    float m = -INFINITY;
    for (int b = l; b < nb; b += 64) m = fmaxf(m, part[(size_t)j * nb + b]);
    m = wave_reduce_max(m);
    if (l == 0) {
        if (J.j[j].pre >= 2) m = (m - J.j[j].mean) / J.j[j].scale;
        scal[j] = (double)m;
    }
}

// out_j[i] = post_j(trilinear(pre_j(src_j)) at the crop-space coordinate (II,JJ,KK)[i]); fast_3D_interp_torch's validity
// test and corner clamps are taken in CROP space (Generator/utils.py:140-192 on the cropped array)
__global__ void gather_targets(GJobs J, int ny, int nz, Box B, uint32_t vol_bytes, const float* __restrict__ II,
                               const float* __restrict__ JJ, const float* __restrict__ KK, int64_t n, int sx,
                               int64_t syz, int flip, const double* __restrict__ scal, float* __restrict__ part) {
    const bool paired = vol_bytes != 0;                           // volumes of 4 GB and more: single loads
    float smin[GJ_MAX], smax[GJ_MAX];
#pragma unroll
    for (int j = 0; j < GJ_MAX; ++j) { smin[j] = INFINITY; smax[j] = -INFINITY; }
    GRID_STRIDE(i, n) {
        const float x = II[i], y = JJ[i], z = KK[i];
        const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= (float)(B.cnx - 1)) && (y <= (float)(B.cny - 1)) &&
                        (z <= (float)(B.cnz - 1));
        int64_t o = i;
        if (flip) {
            const int64_t xi = i / syz;
            o = ((int64_t)sx - 1 - xi) * syz + (i - xi * syz);
        }
        int64_t o000 = 0, o100 = 0, o010 = 0, o110 = 0, o001 = 0, o101 = 0, o011 = 0, o111 = 0;
        float wcx = 0.f, wcy = 0.f, wcz = 0.f, wfx = 0.f, wfy = 0.f, wfz = 0.f;
        bool zpair = false;
        if (ok) {
            const float fxf = floorf(x), fyf = floorf(y), fzf = floorf(z);
            const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
            const int cx = min(fx + 1, B.cnx - 1), cy = min(fy + 1, B.cny - 1), cz = min(fz + 1, B.cnz - 1);
            wcx = x - fxf; wcy = y - fyf; wcz = z - fzf;
            wfx = 1.f - wcx; wfy = 1.f - wcy; wfz = 1.f - wcz;
            const int64_t stx = (int64_t)ny * nz;
            const int64_t ax = (int64_t)(B.x1 + fx) * stx, bx = (int64_t)(B.x1 + cx) * stx;
            const int64_t ay = (int64_t)(B.y1 + fy) * nz, by = (int64_t)(B.y1 + cy) * nz;
            const int64_t az = B.z1 + fz, bz = B.z1 + cz;
            o000 = ax + ay + az; o100 = bx + ay + az; o010 = ax + by + az; o110 = bx + by + az;
            o001 = ax + ay + bz; o101 = bx + ay + bz; o011 = ax + by + bz; o111 = bx + by + bz;
            zpair = cz != fz;
        }
#pragma unroll
        for (int j = 0; j < GJ_MAX; ++j) {
            if (j >= J.n) continue;
            const GJob& jb = J.j[j];
            float r;
            if (ok) {
                const float* X = jb.src;
                float t000, t001, t100, t101, t010, t011, t110, t111;
                if (paired) {
                    const __amdgpu_buffer_rsrc_t R = tex_rsrc(X, vol_bytes);
                    ld_tex2(R, o000, zpair, t000, t001); ld_tex2(R, o100, zpair, t100, t101);
                    ld_tex2(R, o010, zpair, t010, t011); ld_tex2(R, o110, zpair, t110, t111);
                } else {
                    t000 = ld_tex(X + o000); t001 = ld_tex(X + o001); t100 = ld_tex(X + o100); t101 = ld_tex(X + o101);
                    t010 = ld_tex(X + o010); t011 = ld_tex(X + o011); t110 = ld_tex(X + o110); t111 = ld_tex(X + o111);
                }
                const float c00 = pre_op(t000, jb) * wfx + pre_op(t100, jb) * wcx;
                const float c01 = pre_op(t001, jb) * wfx + pre_op(t101, jb) * wcx;
                const float c10 = pre_op(t010, jb) * wfx + pre_op(t110, jb) * wcx;
                const float c11 = pre_op(t011, jb) * wfx + pre_op(t111, jb) * wcx;
                const float c0 = c00 * wfy + c10 * wcy;
                const float c1 = c01 * wfy + c11 * wcy;
                r = c0 * wfz + c1 * wcz;
            } else {
                r = jb.defmax ? (float)scal[j] : 0.f;
            }
            if (jb.post_div != 0.f) r = r / jb.post_div;
            if (jb.clamp) r = fminf(fmaxf(r, jb.clo), jb.chi);
            if (jb.sign != 0.f) r = r * jb.sign;
            jb.out[o] = r;
            if (jb.stat) { smin[j] = fminf(smin[j], r); smax[j] = fmaxf(smax[j], r); }
        }
    }
    __shared__ float red[2][4];
#pragma unroll
    for (int j = 0; j < GJ_MAX; ++j) {
        if (j >= J.n || !J.j[j].stat) continue;                      // uniform over the block
        const float a = wave_reduce_min(smin[j]), b = wave_reduce_max(smax[j]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            part[((size_t)j * gridDim.x + blockIdx.x) * 2 + 0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
            part[((size_t)j * gridDim.x + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        }
    }
}

// one wave per job with stat: scal[GJ_MAX + 2j] = min, [.. + 1] = max of the job's output
__global__ void gather_stat_fold(GJobs J, const float* __restrict__ part, int nb, double* __restrict__ scal) {
    const int j = blockIdx.x, l = threadIdx.x;
    if (!J.j[j].stat) return;
    float a = INFINITY, b = -INFINITY;
    for (int k = l; k < nb; k += 64) {
        a = fminf(a, part[((size_t)j * nb + k) * 2 + 0]);
        b = fmaxf(b, part[((size_t)j * nb + k) * 2 + 1]);
    }
    a = wave_reduce_min(a); b = wave_reduce_max(b);
    if (l == 0) { scal[GJ_MAX + 2 * j] = (double)a; scal[GJ_MAX + 2 * j + 1] = (double)b; }
}

// I -= min(I); I /= max(I)  (read_and_deform_image, utils.py:340-342); max(I - m) = fl(max(I) - m): rounding is monotone
__global__ void minmax_normalise(float* __restrict__ x, int64_t n, const double* __restrict__ mm) {
    const float lo = (float)mm[0], hi = (float)mm[1] - lo;
    GRID_STRIDE(i, n) x[i] = (x[i] - lo) / hi;
}

// nearest gather in crop space + lut + one-hot rows; with flip: out[x,y,z,c] = onehot[sx-1-x, y, z, vflip[c]]
__global__ void __launch_bounds__(256) gather_onehot(const int32_t* __restrict__ S, int ny, int nz, Box B,
                                                     const float* __restrict__ II, const float* __restrict__ JJ,
                                                     const float* __restrict__ KK, int64_t n, int sx, int64_t syz, int flip,
                                                     const int32_t* __restrict__ lut, int nlut, int nl,
                                                     const int32_t* __restrict__ vflip, float* __restrict__ out, int rows) {
    __shared__ int cls[256];
    __shared__ int vf[256];
    for (int c = threadIdx.x; c < nl; c += 256) vf[c] = vflip ? vflip[c] : c;
    const int64_t nblk = bfm_cdiv64(n, 256);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t o = blk * 256 + threadIdx.x;
        __syncthreads();
        if (o < n) {
            int64_t i = o;
            if (flip) {
                const int64_t xi = o / syz;
                i = ((int64_t)sx - 1 - xi) * syz + (o - xi * syz);
            }
            int x = (int)rintf(II[i]), y = (int)rintf(JJ[i]), z = (int)rintf(KK[i]);
            x = min(max(x, 0), B.cnx - 1); y = min(max(y, 0), B.cny - 1); z = min(max(z, 0), B.cnz - 1);
            int s = ld_tex(S + ((int64_t)(B.x1 + x) * ny + (B.y1 + y)) * nz + (B.z1 + z));
            s = min(max(s, 0), nlut - 1);
            cls[threadIdx.x] = lut[s];
        }
        __syncthreads();
        if (rows) {                                             // [nl] rows of n voxels: what the permuted view of the reference is read as
            if (o < n) {
                const int mine = cls[threadIdx.x];
                for (int c = 0; c < nl; ++c) out[(int64_t)c * n + o] = mine == vf[c] ? 1.f : 0.f;
            }
            continue;
        }
        const int64_t base = blk * 256 * (int64_t)nl;
        const int cnt = (int)min((int64_t)256, n - blk * 256) * nl;
        for (int e = threadIdx.x; e < cnt; e += 256) {
            const int v = e / nl, c = e - v * nl;
            out[base + e] = cls[v] == vf[c] ? 1.f : 0.f;
        }
    }
}

// ------------------------------------------------------------------ order statistics of an fp64 field, on the device
__device__ __forceinline__ uint64_t key_of(double v) {
    v = v == 0.0 ? 0.0 : v;                                            // -0.0 and +0.0 are one value to np.percentile
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

struct SelState { uint64_t prefix; int64_t k; int64_t below; uint64_t count_eq; };
constexpr int SEL_BINS = 2048;

__global__ void __launch_bounds__(256) sel_count(const double* __restrict__ x, int64_t n, const SelState* __restrict__ st,
                                                 int shift, int bits, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[SEL_BINS];
    for (int b = threadIdx.x; b < SEL_BINS; b += 256) sh[b] = 0;
    __syncthreads();
    const uint64_t prefix = st->prefix;
    const bool top = shift + bits >= 64;
    const uint32_t mask = (1u << bits) - 1u;
    GRID_STRIDE(i, n) {
        const uint64_t k = key_of(x[i]);
        if (top || (k >> (shift + bits)) == prefix) atomicAdd(&sh[(uint32_t)(k >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SEL_BINS; b += 256)
        if (sh[b]) atomicAdd(&hist[b], sh[b]);
}

// one block of 1024 threads: the bin holding rank st->k, then the histogram is cleared for the next pass
__global__ void __launch_bounds__(1024) sel_pick(uint32_t* __restrict__ hist, SelState* __restrict__ st, int bits) {
    __shared__ uint32_t h[SEL_BINS];
    __shared__ uint64_t wsum[16];
    const int t = threadIdx.x;
    const uint32_t a = hist[2 * t], b = hist[2 * t + 1];
    h[2 * t] = a; h[2 * t + 1] = b;
    hist[2 * t] = 0; hist[2 * t + 1] = 0;
    uint64_t incl = (uint64_t)a + b;                                  // inclusive scan of the pair sums over the block
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t v = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += v;
    }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    uint64_t base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
    incl += base;
    const uint64_t excl = incl - ((uint64_t)a + b);
    const uint64_t k = (uint64_t)st->k;
    __syncthreads();
    if (k >= excl && k < incl) {                                      // exactly one thread
        const int d = (k < excl + a) ? 2 * t : 2 * t + 1;
        const uint64_t before = d == 2 * t ? excl : excl + a;
        st->prefix = (st->prefix << bits) | (uint64_t)d;
        st->k = (int64_t)(k - before);
        st->below += (int64_t)before;
        st->count_eq = h[d];
    }
}

__global__ void sel_init(SelState* st, int64_t k, uint32_t* hist) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->prefix = 0; st->k = k; st->below = 0; st->count_eq = 0; }
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < SEL_BINS; b += gridDim.x * blockDim.x) hist[b] = 0;
}

// smallest element strictly above the selected one (partials; +inf when none)
__global__ void sel_min_above(const double* __restrict__ x, int64_t n, const SelState* __restrict__ st,
                              double* __restrict__ part) {
    const double a = val_of(st->prefix);
    double m = INFINITY;
    GRID_STRIDE(i, n) {
        const double v = x[i];
        if (v > a) m = fmin(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}

// np.percentile's linear method between the order statistics k and k + 1 (numpy _lerp)
__global__ void sel_finish(const SelState* __restrict__ st, const double* __restrict__ part, int nb, int need_next,
                           double t, double* __restrict__ out) {
    double m = INFINITY;
    for (int b = threadIdx.x; b < nb; b += 64) m = fmin(m, part[b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) {
        const double a = val_of(st->prefix);
        double b = a;
        if (need_next && (uint64_t)(st->k + 1) >= st->count_eq) b = m;
        const double diff = b - a;
        double r = a + diff * t;
        if (t >= 0.5) r = b - diff * (1 - t);
        if (diff == 0) r = a;
        out[0] = r;
        out[1] = a;
        out[2] = b;
    }
}

// masked = x * (x >= thr); block partial maxima of masked
__global__ void shape_threshold(const double* __restrict__ x, int64_t n, const double* __restrict__ thr,
                                double* __restrict__ masked, double* __restrict__ mask, double* __restrict__ part) {
    const double th = thr[0];
    double mx = -INFINITY;
    GRID_STRIDE(i, n) {
        const double v = x[i];
        const double m = v >= th ? 1.0 : 0.0;
        const double r = v * m;
        if (mask) mask[i] = m;
        masked[i] = r;
        mx = fmax(mx, r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__global__ void fold_max_f64(const double* __restrict__ part, int nb, double* __restrict__ out) {
    double m = -INFINITY;
    for (int b = threadIdx.x; b < nb; b += 64) m = fmax(m, part[b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) out[0] = m;
}

// binarize (utils.py:65-72): P = (p >= thres * max(p)) in p's dtype; block partial sums of P
template <typename T>
__global__ void shape_binarize(const T* __restrict__ p, int64_t n, const double* __restrict__ pmax, double thres,
                               T* __restrict__ P, double* __restrict__ part) {
    double s = 0.0;
    if (sizeof(T) == 8) {
        const double t = thres * pmax[0];
        GRID_STRIDE(i, n) { const double m = (double)p[i] >= t ? 1.0 : 0.0; P[i] = (T)m; s += m; }
    } else {
        const float t = (float)thres * (float)pmax[0];
        GRID_STRIDE(i, n) { const float m = (float)p[i] >= t ? 1.f : 0.f; P[i] = (T)m; s += (double)m; }
    }
    s = wave_reduce_sum(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// deterministic fold of up to 1024 partial sums: lane l adds its strided partials, then a fixed xor tree
__global__ void fold_sum_f64(const double* __restrict__ part, int nb, int nout, double* __restrict__ out) {
    const int q = blockIdx.x;                                         // partials of quantity q at part[q * nb ..]
    if (q >= nout) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) s += part[(size_t)q * nb + b];
    s = wave_reduce_sum(s);
    if (threadIdx.x == 0) out[q] = s;
}

// ------------------------------------------------------------------ pathology inside generate_sample / encode_pathology
// target['pathology'][cer == 0] = 0, same for pathology_prob (in place, own dtype); partial sums of the masked P
template <typename T>
__global__ void pathology_mask(T* __restrict__ P, T* __restrict__ Pprob, const float* __restrict__ cer, int64_t n,
                               double* __restrict__ part) {
    double s = 0.0;
    GRID_STRIDE(i, n) {
        T p = P[i];
        if (cer[i] == 0.f) { p = (T)0; P[i] = p; Pprob[i] = (T)0; }
        s += (double)p;
    }
    s = wave_reduce_sum(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// partials of sum(I * P) (product in the promoted dtype, like torch) and sum(P)
template <typename T>
__global__ void dot_sum_partial(const float* __restrict__ I, const T* __restrict__ P, int64_t n,
                                double* __restrict__ part, int nb) {
    double a = 0.0, b = 0.0;
    GRID_STRIDE(i, n) {
        const T p = P[i];
        a += (double)((T)I[i] * p);
        b += (double)p;
    }
    a = wave_reduce_sum(a); b = wave_reduce_sum(b);
    __shared__ double red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[(size_t)nb + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

struct EncP { float u[4]; int direction; };                            // u = rand draws [mu0, mu1, sigma0, sigma1]

// I + Pprob * (mu[round(P)] + sigma[round(P)] * randn), clamped at 0; mu / sigma from I_mu = sum(I*P) / sum(P) on the
// device; direction: 1 / 0 given by the host, -1 = (gm_mean > wm_mean) from the class sums in `stats`
template <typename T>
__global__ void pathology_encode_dev(const float* __restrict__ I, const T* __restrict__ P, const T* __restrict__ Pprob,
                                     const float* __restrict__ rn, const double* __restrict__ dotsum,
                                     const double* __restrict__ stats, EncP E, int64_t n, float* __restrict__ out) {
    const float I_mu = (float)(dotsum[0] / dotsum[1]);
    bool dir = E.direction != 0;
    if (E.direction < 0) {
        const double wm = stats[1] != 0.0 ? stats[0] / stats[1] : NAN, gm = stats[3] != 0.0 ? stats[2] / stats[3] : NAN;
        dir = gm > wm;
    }
    float mu0 = (3.f * I_mu) / 4.f + (I_mu / 4.f) * E.u[0], mu1 = (3.f * I_mu) / 4.f + (I_mu / 4.f) * E.u[1];
    if (!dir) { mu0 = -mu0; mu1 = -mu1; }
    const float s0 = (I_mu / 4.f) * E.u[2], s1 = (I_mu / 4.f) * E.u[3];
    GRID_STRIDE(i, n) {
        const bool one = rint((double)P[i]) >= 1.0;
        const float g = (one ? mu1 : mu0) + (one ? s1 : s0) * rn[i];
        const float v = (float)((T)I[i] + Pprob[i] * (T)g);               // fp64 operands promote the sum, like torch
        out[i] = v < 0.f ? 0.f : v;
    }
}

// ------------------------------------------------------------------ separable-coordinate trilinear sample
__global__ void interp_linear_axes(const float* __restrict__ X, int nx, int ny, int nz, const float* __restrict__ ax,
                                   const float* __restrict__ ay, const float* __restrict__ az, int ox, int oy, int oz,
                                   float defv, uint32_t vol_bytes, float* __restrict__ out) {
    const int64_t n = (int64_t)ox * oy * oz;
    GRID_STRIDE(i, n) {
        const int k = (int)(i % oz);
        const int j = (int)((i / oz) % oy);
        const int ii = (int)(i / ((int64_t)oy * oz));
        const float x = ax[ii], y = ay[j], z = az[k];
        const bool ok = (x > 0.f) && (y > 0.f) && (z > 0.f) && (x <= (float)(nx - 1)) && (y <= (float)(ny - 1)) &&
                        (z <= (float)(nz - 1));
        if (!ok) { out[i] = defv; continue; }
        const float fxf = floorf(x), fyf = floorf(y), fzf = floorf(z);
        const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
        const int cx = min(fx + 1, nx - 1), cy = min(fy + 1, ny - 1), cz = min(fz + 1, nz - 1);
        const float wcx = x - fxf, wcy = y - fyf, wcz = z - fzf;
        const float wfx = 1.f - wcx, wfy = 1.f - wcy, wfz = 1.f - wcz;
        const int64_t sx = (int64_t)ny * nz, sy = nz;
        float t000, t001, t100, t101, t010, t011, t110, t111;
        if (vol_bytes) {
            const __amdgpu_buffer_rsrc_t R = tex_rsrc(X, vol_bytes);
            const bool zp = cz != fz;
            ld_tex2(R, fx * sx + fy * sy + fz, zp, t000, t001); ld_tex2(R, cx * sx + fy * sy + fz, zp, t100, t101);
            ld_tex2(R, fx * sx + cy * sy + fz, zp, t010, t011); ld_tex2(R, cx * sx + cy * sy + fz, zp, t110, t111);
        } else {
            auto at = [&](int a, int b, int c) { return ld_tex(X + a * sx + b * sy + c); };
            t000 = at(fx, fy, fz); t001 = at(fx, fy, cz); t100 = at(cx, fy, fz); t101 = at(cx, fy, cz);
            t010 = at(fx, cy, fz); t011 = at(fx, cy, cz); t110 = at(cx, cy, fz); t111 = at(cx, cy, cz);
        }
        const float c00 = t000 * wfx + t100 * wcx;
        const float c01 = t001 * wfx + t101 * wcx;
        const float c10 = t010 * wfx + t110 * wcx;
        const float c11 = t011 * wfx + t111 * wcx;
        const float c0 = c00 * wfy + c10 * wcy;
        const float c1 = c01 * wfy + c11 * wcy;
        out[i] = c0 * wfz + c1 * wcz;
    }
}

// I_final = I / maxi; residual = high_res / maxi - I_final   (datasets.py:340-347); optional flip along axis 0
__global__ void sample_finalize(const float* __restrict__ I, const float* __restrict__ hr, int64_t n,
                                const double* __restrict__ maxi, int sx, int64_t syz, int flip, float* __restrict__ out,
                                float* __restrict__ res) {
    const float m = (float)maxi[0];
    GRID_STRIDE(i, n) {
        int64_t o = i;
        if (flip) {
            const int64_t xi = i / syz;
            o = ((int64_t)sx - 1 - xi) * syz + (i - xi * syz);
        }
        const float f = I[i] / m;
        out[o] = f;
        if (res) res[o] = hr[i] / m + (-1.f) * f;
    }
}

// elementwise with the scalar operand in device memory: 0: x / s   1: x >= thres_rel * s (fp32 product)
__global__ void ew_dev(int op, const float* __restrict__ x, int64_t n, const double* __restrict__ s, float a,
                       float* __restrict__ out) {
    const float sv = (float)s[0];
    GRID_STRIDE(i, n) {
        const float v = x[i];
        out[i] = op == 0 ? v / sv : (v >= a * sv ? 1.f : 0.f);
    }
}

}  // namespace

extern "C" int bfm_randn_philox(float* out, int64_t n, uint64_t seed, uint64_t offset, float scale, bfm_stream_t stream) {
    if (!out || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(randn_philox, dim3(grid_for((n + 3) / 4)), dim3(256), 0, bfm_s(stream), out, n, seed, offset, scale);
    return bfm_launch_status();
}

namespace {
bool zoom_ok(const bfm_zoom_axis_t* ax) {
    for (int a = 0; a < 3; ++a)
        if (!ax[a].f || !ax[a].c || !ax[a].wf || !ax[a].wc) return false;
    return true;
}
ZoomTabs zoom_tabs(const bfm_zoom_axis_t* ax) {
    return ZoomTabs{ax[0].f, ax[0].c, ax[1].f, ax[1].c, ax[2].f, ax[2].c, ax[0].wf, ax[0].wc, ax[1].wf, ax[1].wc,
                    ax[2].wf, ax[2].wc};
}
DefP def_params(const float* A, const float* c2, const int* shp, const float* lo) {
    DefP P;
    for (int i = 0; i < 9; ++i) P.a[i] = A[i];
    for (int i = 0; i < 3; ++i) { P.c[i] = c2[i]; P.shp[i] = shp[i]; P.lo[i] = lo ? lo[i] : 0.f; }
    return P;
}
constexpr int DEF_BLOCKS = 1024;
}  // namespace

extern "C" size_t bfm_deform_zoom_workspace(void) { return (size_t)DEF_BLOCKS * 6 * sizeof(float); }

extern "C" int bfm_deform_zoom_minmax(const float* Fsmall, int fnx, int fny, int fnz, const bfm_zoom_axis_t* ax,
                                      int photo_zero_y, int sx, int sy, int sz, const float* A_host, const float* c2_host,
                                      const int* shp_host, float* minmax6, void* workspace, size_t workspace_bytes,
                                      bfm_stream_t stream) {
    if (!A_host || !c2_host || !shp_host || !minmax6 || !workspace || sx <= 0 || sy <= 0 || sz <= 0) return BFM_E_ARG;
    if (Fsmall && (!ax || !zoom_ok(ax) || fnx <= 0 || fny <= 0 || fnz <= 0)) return BFM_E_ARG;
    if (workspace_bytes < bfm_deform_zoom_workspace()) return BFM_E_WORKSPACE;
    const int nb = grid_for((int64_t)sx * sy * sz, 256, DEF_BLOCKS);
    ZoomTabs t{};
    if (Fsmall) t = zoom_tabs(ax);
    hipLaunchKernelGGL(deform_zoom<0>, dim3(nb), dim3(256), 0, bfm_s(stream), Fsmall, fnx, fny, fnz, t, photo_zero_y, sx,
                       sy, sz, def_params(A_host, c2_host, shp_host, nullptr), (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, static_cast<float*>(workspace));
    hipLaunchKernelGGL(minmax6_fold, dim3(1), dim3(384), 0, bfm_s(stream), static_cast<const float*>(workspace), nb, minmax6);
    return bfm_launch_status();
}

extern "C" int bfm_deform_zoom_write(const float* Fsmall, int fnx, int fny, int fnz, const bfm_zoom_axis_t* ax,
                                     int photo_zero_y, int sx, int sy, int sz, const float* A_host, const float* c2_host,
                                     const int* shp_host, const float* lo_host, float* xx, float* yy, float* zz,
                                     float* F_out, bfm_stream_t stream) {
    if (!A_host || !c2_host || !shp_host || !lo_host || !xx || !yy || !zz || sx <= 0 || sy <= 0 || sz <= 0) return BFM_E_ARG;
    if (Fsmall && (!ax || !zoom_ok(ax) || fnx <= 0 || fny <= 0 || fnz <= 0)) return BFM_E_ARG;
    ZoomTabs t{};
    if (Fsmall) t = zoom_tabs(ax);
    hipLaunchKernelGGL(deform_zoom<1>, dim3(grid_for((int64_t)sx * sy * sz)), dim3(256), 0, bfm_s(stream), Fsmall, fnx, fny,
                       fnz, t, photo_zero_y, sx, sy, sz, def_params(A_host, c2_host, shp_host, lo_host), xx, yy, zz, F_out,
                       (float*)nullptr);
    return bfm_launch_status();
}

namespace {
constexpr int GT_BLOCKS = 2048;
constexpr int BM_BLOCKS = 256;
bool make_jobs(const bfm_gather_job_t* jobs, int njobs, GJobs& J) {
    if (!jobs || njobs <= 0 || njobs > GJ_MAX) return false;
    J.n = njobs;
    for (int j = 0; j < GJ_MAX; ++j) {
        GJob g{};
        if (j < njobs) {
            const bfm_gather_job_t& s = jobs[j];
            if (!s.src || !s.out || s.pre < 0 || s.pre > 2 || (s.pre == 2 && s.scale == 0.f)) return false;
            g.src = s.src; g.out = s.out; g.mean = s.mean; g.scale = s.scale; g.post_div = s.post_div;
            g.clo = s.clamp_lo; g.chi = s.clamp_hi; g.sign = s.sign; g.pre = s.pre; g.defmax = s.default_max ? 1 : 0;
            g.clamp = s.clamp ? 1 : 0; g.stat = s.want_minmax ? 1 : 0;
        }
        J.j[j] = g;
    }
    return true;
}
bool make_box(const int* box6, int nx, int ny, int nz, Box& B) {
    if (!box6) return false;
    const int x2 = std::min(box6[3], nx), y2 = std::min(box6[4], ny), z2 = std::min(box6[5], nz);
    B.x1 = box6[0]; B.y1 = box6[1]; B.z1 = box6[2];
    B.cnx = x2 - B.x1; B.cny = y2 - B.y1; B.cnz = z2 - B.z1;
    return B.x1 >= 0 && B.y1 >= 0 && B.z1 >= 0 && B.cnx > 0 && B.cny > 0 && B.cnz > 0;
}
}  // namespace

extern "C" size_t bfm_gather_targets_workspace(void) {
    return (size_t)GJ_MAX * (BM_BLOCKS + 2 * GT_BLOCKS) * sizeof(float);
}

extern "C" int bfm_gather_targets(const bfm_gather_job_t* jobs, int njobs, int nx, int ny, int nz, const int* box6_host,
                                  const float* II, const float* JJ, const float* KK, int sx, int sy, int sz, int flip0,
                                  double* scalars, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    GJobs J;
    Box B;
    if (!make_jobs(jobs, njobs, J) || !II || !JJ || !KK || !scalars || !workspace || nx <= 0 || ny <= 0 || nz <= 0 ||
        sx <= 0 || sy <= 0 || sz <= 0)
        return BFM_E_ARG;
    if (!make_box(box6_host, nx, ny, nz, B)) return BFM_E_SHAPE;
    if (workspace_bytes < bfm_gather_targets_workspace()) return BFM_E_WORKSPACE;
    float* part_bm = static_cast<float*>(workspace);
    float* part_st = part_bm + (size_t)GJ_MAX * BM_BLOCKS;
    hipStream_t st = bfm_s(stream);
    bool any_max = false, any_stat = false;
    for (int j = 0; j < njobs; ++j) { any_max |= J.j[j].defmax != 0; any_stat |= J.j[j].stat != 0; }
    if (any_max) {
        hipLaunchKernelGGL(box_max_partial, dim3(BM_BLOCKS, njobs), dim3(256), 0, st, J, ny, nz, B, part_bm);
        hipLaunchKernelGGL(box_max_fold, dim3(njobs), dim3(64), 0, st, J, part_bm, BM_BLOCKS, scalars);
    }
    const int64_t n = (int64_t)sx * sy * sz;
    const int nb = grid_for(n, 256, GT_BLOCKS);
    const int64_t vbytes = (int64_t)nx * ny * nz * 4;
    hipLaunchKernelGGL(gather_targets, dim3(nb), dim3(256), 0, st, J, ny, nz, B,
                       (uint32_t)(vbytes < ((int64_t)1 << 32) ? vbytes : 0), II, JJ, KK, n, sx, (int64_t)sy * sz,
                       flip0 ? 1 : 0, scalars, part_st);
    if (any_stat) hipLaunchKernelGGL(gather_stat_fold, dim3(njobs), dim3(64), 0, st, J, part_st, nb, scalars);
    return bfm_launch_status();
}

extern "C" int bfm_minmax_normalise(float* x, int64_t n, const double* minmax_dev, bfm_stream_t stream) {
    if (!x || !minmax_dev || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(minmax_normalise, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), x, n, minmax_dev);
    return bfm_launch_status();
}

static int gather_onehot_launch(const int32_t* S, int nx, int ny, int nz, const int* box6_host, const float* II,
                                const float* JJ, const float* KK, int sx, int sy, int sz, int flip0, const int32_t* lut,
                                int nlut, int n_labels, const int32_t* vflip, float* out, int rows, bfm_stream_t stream) {
    Box B;
    if (!S || !II || !JJ || !KK || !lut || !out || nlut <= 0 || n_labels <= 0 || n_labels > 256 || sx <= 0 || sy <= 0 ||
        sz <= 0 || nx <= 0 || ny <= 0 || nz <= 0)
        return BFM_E_ARG;
    if (!make_box(box6_host, nx, ny, nz, B)) return BFM_E_SHAPE;
    const int64_t n = (int64_t)sx * sy * sz;
    hipLaunchKernelGGL(gather_onehot, dim3(grid_for(n, 256, 4096)), dim3(256), 0, bfm_s(stream), S, ny, nz, B, II, JJ, KK,
                       n, sx, (int64_t)sy * sz, flip0 ? 1 : 0, lut, nlut, n_labels, vflip, out, rows);
    return bfm_launch_status();
}

extern "C" int bfm_gather_onehot(const int32_t* S, int nx, int ny, int nz, const int* box6_host, const float* II,
                                 const float* JJ, const float* KK, int sx, int sy, int sz, int flip0, const int32_t* lut,
                                 int nlut, int n_labels, const int32_t* vflip, float* out, bfm_stream_t stream) {
    return gather_onehot_launch(S, nx, ny, nz, box6_host, II, JJ, KK, sx, sy, sz, flip0, lut, nlut, n_labels, vflip, out, 0,
                                stream);
}

// the same with out as [n_labels][sx][sy][sz] -- the memory the reference's one_hot(...).permute([3, 0, 1, 2]) view is read
// as by every consumer (the training criterion walks one class at a time): no strided 0.9 GB copy per 160^3 sample later
extern "C" int bfm_gather_onehot_rows(const int32_t* S, int nx, int ny, int nz, const int* box6_host, const float* II,
                                      const float* JJ, const float* KK, int sx, int sy, int sz, int flip0, const int32_t* lut,
                                      int nlut, int n_labels, const int32_t* vflip, float* out, bfm_stream_t stream) {
    return gather_onehot_launch(S, nx, ny, nz, box6_host, II, JJ, KK, sx, sy, sz, flip0, lut, nlut, n_labels, vflip, out, 1,
                                stream);
}

namespace {
constexpr int SEL_BLOCKS = 512;
constexpr int RED_BLOCKS = 1024;
}

extern "C" size_t bfm_percentile_workspace(void) {
    return 256 + (size_t)SEL_BINS * sizeof(uint32_t) + (size_t)RED_BLOCKS * sizeof(double);
}

extern "C" int bfm_percentile_f64(const double* x, int64_t n, int64_t k_lo, int need_next, double t, double* out3,
                                  void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!x || !out3 || !workspace || n <= 0 || k_lo < 0 || k_lo >= n) return BFM_E_ARG;
    if (workspace_bytes < bfm_percentile_workspace()) return BFM_E_WORKSPACE;
    SelState* st = static_cast<SelState*>(workspace);
    uint32_t* hist = reinterpret_cast<uint32_t*>(static_cast<char*>(workspace) + 256);
    double* part = reinterpret_cast<double*>(static_cast<char*>(workspace) + 256 + SEL_BINS * sizeof(uint32_t));
    hipStream_t s = bfm_s(stream);
    hipLaunchKernelGGL(sel_init, dim3(2), dim3(1024), 0, s, st, k_lo, hist);
    const int shifts[6] = {53, 42, 31, 20, 9, 0}, bits[6] = {11, 11, 11, 11, 11, 9};
    const int nb = grid_for(n, 256, SEL_BLOCKS);
    for (int p = 0; p < 6; ++p) {
        hipLaunchKernelGGL(sel_count, dim3(nb), dim3(256), 0, s, x, n, st, shifts[p], bits[p], hist);
        hipLaunchKernelGGL(sel_pick, dim3(1), dim3(1024), 0, s, hist, st, bits[p]);
    }
    const int nbm = grid_for(n, 256, RED_BLOCKS);
    hipLaunchKernelGGL(sel_min_above, dim3(nbm), dim3(256), 0, s, x, n, st, part);
    hipLaunchKernelGGL(sel_finish, dim3(1), dim3(64), 0, s, st, part, nbm, need_next ? 1 : 0, t, out3);
    return bfm_launch_status();
}

extern "C" size_t bfm_shape_workspace(void) { return (size_t)RED_BLOCKS * sizeof(double); }

extern "C" int bfm_shape_threshold_f64(const double* noise, int64_t n, const double* thr_dev, double* masked, double* mask,
                                       double* max_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!noise || !thr_dev || !masked || !max_out || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_shape_workspace()) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(n, 256, RED_BLOCKS);
    hipLaunchKernelGGL(shape_threshold, dim3(nb), dim3(256), 0, bfm_s(stream), noise, n, thr_dev, masked, mask, part);
    hipLaunchKernelGGL(fold_max_f64, dim3(1), dim3(64), 0, bfm_s(stream), part, nb, max_out);
    return bfm_launch_status();
}

extern "C" int bfm_shape_binarize(const void* p, int is_f64, int64_t n, const double* max_dev, double thres, void* P,
                                  double* sum_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!p || !max_dev || !P || !sum_out || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_shape_workspace()) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(n, 256, RED_BLOCKS);
    if (is_f64)
        hipLaunchKernelGGL(shape_binarize<double>, dim3(nb), dim3(256), 0, bfm_s(stream), (const double*)p, n, max_dev,
                           thres, (double*)P, part);
    else
        hipLaunchKernelGGL(shape_binarize<float>, dim3(nb), dim3(256), 0, bfm_s(stream), (const float*)p, n, max_dev, thres,
                           (float*)P, part);
    hipLaunchKernelGGL(fold_sum_f64, dim3(1), dim3(64), 0, bfm_s(stream), part, nb, 1, sum_out);
    return bfm_launch_status();
}

extern "C" int bfm_pathology_mask(void* P, void* Pprob, int is_f64, const float* cerebral, int64_t n, double* sum_out,
                                  void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!P || !Pprob || !cerebral || !sum_out || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_shape_workspace()) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(n, 256, RED_BLOCKS);
    if (is_f64)
        hipLaunchKernelGGL(pathology_mask<double>, dim3(nb), dim3(256), 0, bfm_s(stream), (double*)P, (double*)Pprob,
                           cerebral, n, part);
    else
        hipLaunchKernelGGL(pathology_mask<float>, dim3(nb), dim3(256), 0, bfm_s(stream), (float*)P, (float*)Pprob,
                           cerebral, n, part);
    hipLaunchKernelGGL(fold_sum_f64, dim3(1), dim3(64), 0, bfm_s(stream), part, nb, 1, sum_out);
    return bfm_launch_status();
}

extern "C" size_t bfm_pathology_encode_workspace(void) { return (size_t)2 * RED_BLOCKS * sizeof(double); }

extern "C" int bfm_pathology_encode_dev(const float* I, const void* P, const void* Pprob, int is_f64, const float* randn,
                                        const float* u4_host, int direction, const double* class_stats_dev, int64_t n,
                                        float* out, double* dotsum_out, void* workspace, size_t workspace_bytes,
                                        bfm_stream_t stream) {
    if (!I || !P || !Pprob || !randn || !u4_host || !out || !dotsum_out || !workspace || n <= 0) return BFM_E_ARG;
    if (direction < 0 && !class_stats_dev) return BFM_E_ARG;
    if (workspace_bytes < bfm_pathology_encode_workspace()) return BFM_E_WORKSPACE;
    double* part = static_cast<double*>(workspace);
    const int nb = grid_for(n, 256, RED_BLOCKS);
    hipStream_t s = bfm_s(stream);
    EncP E;
    for (int k = 0; k < 4; ++k) E.u[k] = u4_host[k];
    E.direction = direction;
    if (is_f64) {
        hipLaunchKernelGGL(dot_sum_partial<double>, dim3(nb), dim3(256), 0, s, I, (const double*)P, n, part, nb);
        hipLaunchKernelGGL(fold_sum_f64, dim3(2), dim3(64), 0, s, part, nb, 2, dotsum_out);
        hipLaunchKernelGGL(pathology_encode_dev<double>, dim3(grid_for(n)), dim3(256), 0, s, I, (const double*)P,
                           (const double*)Pprob, randn, dotsum_out, class_stats_dev, E, n, out);
    } else {
        hipLaunchKernelGGL(dot_sum_partial<float>, dim3(nb), dim3(256), 0, s, I, (const float*)P, n, part, nb);
        hipLaunchKernelGGL(fold_sum_f64, dim3(2), dim3(64), 0, s, part, nb, 2, dotsum_out);
        hipLaunchKernelGGL(pathology_encode_dev<float>, dim3(grid_for(n)), dim3(256), 0, s, I, (const float*)P,
                           (const float*)Pprob, randn, dotsum_out, class_stats_dev, E, n, out);
    }
    return bfm_launch_status();
}

extern "C" int bfm_interp3d_linear_axes(const float* X, int nx, int ny, int nz, const float* ax, const float* ay,
                                        const float* az, int ox, int oy, int oz, float default_value, float* out,
                                        bfm_stream_t stream) {
    if (!X || !ax || !ay || !az || !out || nx <= 0 || ny <= 0 || nz <= 0 || ox <= 0 || oy <= 0 || oz <= 0) return BFM_E_ARG;
    const int64_t vbytes = (int64_t)nx * ny * nz * 4;
    hipLaunchKernelGGL(interp_linear_axes, dim3(grid_for((int64_t)ox * oy * oz)), dim3(256), 0, bfm_s(stream), X, nx, ny,
                       nz, ax, ay, az, ox, oy, oz, default_value, (uint32_t)(vbytes < ((int64_t)1 << 32) ? vbytes : 0), out);
    return bfm_launch_status();
}

extern "C" int bfm_sample_finalize(const float* I, const float* high_res, int sx, int sy, int sz, const double* max_dev,
                                   int flip0, float* input_out, float* residual_out, bfm_stream_t stream) {
    if (!I || !max_dev || !input_out || sx <= 0 || sy <= 0 || sz <= 0 || (residual_out && !high_res)) return BFM_E_ARG;
    const int64_t n = (int64_t)sx * sy * sz;
    hipLaunchKernelGGL(sample_finalize, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), I, high_res, n, max_dev, sx,
                       (int64_t)sy * sz, flip0 ? 1 : 0, input_out, residual_out);
    return bfm_launch_status();
}

extern "C" int bfm_ew_dev(int op, const float* x, int64_t n, const double* scalar_dev, float a, float* out,
                          bfm_stream_t stream) {
    if (!x || !scalar_dev || !out || n <= 0 || op < 0 || op > 1) return BFM_E_ARG;
    hipLaunchKernelGGL(ew_dev, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), op, x, n, scalar_dev, a, out);
    return bfm_launch_status();
}
