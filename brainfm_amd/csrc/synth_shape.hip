// Pathology-shape synthesis kernels: Perlin noise, curl velocity, upwind advection RHS and the
// tensor arithmetic of the Dormand-Prince integrator (all HBM-bound streaming kernels).
//
//   perlin3d            : generate_perlin_noise_3d       ShapeID/perlin3d.py:15-90  (fp64 like NumPy)
//   radix_hist_f64 / threshold_mask : np.percentile selection + mask   perlin3d.py:84-90
//   curl3d              : stream_3D / gradient_c         ShapeID/misc.py:66-80,198-259
//   advect_rhs          : AdvDiffPDE.forward (adv, div-free V, Neumann BC)   DiffEqs/pde.py:616-640,499-509,301-328
//   rk_combine / rk_error_sumsq / scaled_sumsq / dense_eval : DiffEqs/rk_common.py:22-61, misc.py:84-170, interp.py
//
// The reference recomputes full forward AND backward gradient volumes three times per RHS (quirk Q11)
// and materialises every RK stage combination as a chain of torch ops; here one RHS evaluation reads
// C once (7-point stencil through the caches) plus the three velocities and writes one volume.
#include "bfm_common.h"

namespace {

inline int grid_for(int64_t n, int tpb = 256, int cap = 8192) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ Perlin
__device__ __forceinline__ double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }

// One wave per (x, y) row; along z the wave walks one lattice cell at a time (64 voxels per step), so the cell -- and with
// it the eight gradient vectors -- is wave-uniform: the gradients come through the scalar cache and the x / y parts of the
// eight corner products, (a * g0 + b * g1), are formed once per cell instead of once per voxel.  Same expressions in the
// same order as the per-voxel form ((a*g0 + b*g1) + c*g2, the lerps): the same bits.  (Round 3: 49 us for 160^3; the
// 24 fp64 gradient loads per voxel were the cost, not the arithmetic.)
__global__ void __launch_bounds__(256) perlin3d(const double* __restrict__ grad, int sx, int sy, int sz, int rx, int ry,
                                                int rz, double* __restrict__ out) {
    const double dx = (double)rx / (double)sx, dy = (double)ry / (double)sy, dz = (double)rz / (double)sz;
    const int cx = sx / rx, cy = sy / ry, cz = sz / rz;          // d = shape // res
    const int gy = ry + 1, gz = rz + 1;
    const int lane = threadIdx.x & 63;
    const int rows = sx * sy;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        const int x = r / sy, y = r - x * sy;
        // grid = (mgrid = index*delta) % 1 ; lattice cell = index // d  (NumPy semantics, fp64); v % 1 for v >= 0 is
        // v - floor(v), exactly
        const double vx = (double)x * dx, vy = (double)y * dy;
        const double fx = vx - floor(vx), fy = vy - floor(vy);
        const int ix = x / cx, iy = y / cy;
        const double t0 = fade(fx), t1 = fade(fy);
        for (int iz = 0; iz * cz < sz; ++iz) {                     // iz = z // cz can exceed rz - 1 only if sz % rz != 0 (refused)
            const double* g000 = grad + ((int64_t)((ix + 0) * gy + (iy + 0)) * gz + iz) * 3;
            const double* g100 = grad + ((int64_t)((ix + 1) * gy + (iy + 0)) * gz + iz) * 3;
            const double* g010 = grad + ((int64_t)((ix + 0) * gy + (iy + 1)) * gz + iz) * 3;
            const double* g110 = grad + ((int64_t)((ix + 1) * gy + (iy + 1)) * gz + iz) * 3;
            // corner (a,b,c): gradient at cell + (a,b,c), offsets (fx - a, fy - b, fz - c); the +1-in-z corners are 3 doubles on
            const double p000 = fx * g000[0] + fy * g000[1], p100 = (fx - 1) * g100[0] + fy * g100[1];
            const double p010 = fx * g010[0] + (fy - 1) * g010[1], p110 = (fx - 1) * g110[0] + (fy - 1) * g110[1];
            const double p001 = fx * g000[3] + fy * g000[4], p101 = (fx - 1) * g100[3] + fy * g100[4];
            const double p011 = fx * g010[3] + (fy - 1) * g010[4], p111 = (fx - 1) * g110[3] + (fy - 1) * g110[4];
            const double q000 = g000[2], q100 = g100[2], q010 = g010[2], q110 = g110[2];
            const double q001 = g000[5], q101 = g100[5], q011 = g010[5], q111 = g110[5];
            const int zend = min(sz, (iz + 1) * cz);
            for (int z = iz * cz + lane; z < zend; z += 64) {
                const double vz = (double)z * dz;
                const double fz = vz - floor(vz);
                const double n000 = p000 + fz * q000, n100 = p100 + fz * q100;
                const double n010 = p010 + fz * q010, n110 = p110 + fz * q110;
                const double n001 = p001 + (fz - 1) * q001, n101 = p101 + (fz - 1) * q101;
                const double n011 = p011 + (fz - 1) * q011, n111 = p111 + (fz - 1) * q111;
                const double t2 = fade(fz);
                const double n00 = n000 * (1 - t0) + t0 * n100;
                const double n10 = n010 * (1 - t0) + t0 * n110;
                const double n01 = n001 * (1 - t0) + t0 * n101;
                const double n11 = n011 * (1 - t0) + t0 * n111;
                const double n0 = (1 - t1) * n00 + t1 * n10;
                const double n1 = (1 - t1) * n01 + t1 * n11;
                out[(int64_t)r * sz + z] = (1 - t2) * n0 + t2 * n1;
            }
        }
    }
}

// order-preserving map double -> uint64
__device__ __forceinline__ uint64_t key_of(double v) {
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// histogram of the 16-bit digit at `shift` among keys whose higher bits equal `prefix`
__global__ void radix_hist(const double* __restrict__ x, int64_t n, uint64_t prefix, int shift, uint32_t* hist) {
    GRID_STRIDE(i, n) {
        const uint64_t k = key_of(x[i]);
        const bool match = (shift + 16 >= 64) ? true : ((k >> (shift + 16)) == prefix);
        if (match) atomicAdd(&hist[(k >> shift) & 0xFFFF], 1u);
    }
}

__global__ void threshold_mask(const double* __restrict__ x, int64_t n, double thr, double* __restrict__ masked,
                               double* __restrict__ mask) {
    GRID_STRIDE(i, n) {
        const double m = x[i] >= thr ? 1.0 : 0.0;
        mask[i] = m;
        masked[i] = x[i] * m;
    }
}

// ------------------------------------------------------------------ curl of three potentials
__device__ __forceinline__ float grad_c(const double* __restrict__ X, int64_t i, int pos, int len, int64_t stride) {
    if (pos == 0) return (float)(X[i + stride] - X[i]);
    if (pos == len - 1) return (float)(X[i] - X[i - stride]);
    return (float)((X[i + stride] - X[i - stride]) / 2);
}

__global__ void curl3d(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                       int sx, int sy, int sz, float mult, float* __restrict__ Vx, float* __restrict__ Vy,
                       float* __restrict__ Vz) {
    const int64_t n = (int64_t)sx * sy * sz;
    const int64_t stx = (int64_t)sy * sz, sty = sz;
    GRID_STRIDE(i, n) {
        const int z = (int)(i % sz);
        const int y = (int)((i / sz) % sy);
        const int x = (int)(i / stx);
        const float a_y = grad_c(a, i, y, sy, sty), a_z = grad_c(a, i, z, sz, 1);
        const float b_x = grad_c(b, i, x, sx, stx), b_z = grad_c(b, i, z, sz, 1);
        const float c_x = grad_c(c, i, x, sx, stx), c_y = grad_c(c, i, y, sy, sty);
        Vx[i] = (c_y - b_z) * mult;
        Vy[i] = (a_z - c_x) * mult;
        Vz[i] = (b_x - a_y) * mult;
    }
}

// ------------------------------------------------------------------ upwind advection RHS
template <typename T>
__global__ void advect_rhs(const T* __restrict__ C, const float* __restrict__ Vx, const float* __restrict__ Vy,
                           const float* __restrict__ Vz, int sx, int sy, int sz, int neumann,
                           float* __restrict__ out) {
    const int64_t n = (int64_t)sx * sy * sz;
    const int64_t stx = (int64_t)sy * sz, sty = sz;
    GRID_STRIDE(i, n) {
        const int z = (int)(i % sz);
        const int y = (int)((i / sz) % sy);
        const int x = (int)(i / stx);
        // value of the boundary-conditioned field at (a,b,c): faces replaced by the replicate pad of the interior
        auto U = [&](int a, int b, int c) -> T {
            if (neumann) {
                a = min(max(a, 1), sx - 2); b = min(max(b, 1), sy - 2); c = min(max(c, 1), sz - 2);
            }
            return C[a * stx + b * sty + c];
        };
        const T u = U(x, y, z);
        float acc = 0.f;
        bool first = true;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const int pos = ax == 0 ? x : (ax == 1 ? y : z);
            const int len = ax == 0 ? sx : (ax == 1 ? sy : sz);
            const int ex = ax == 0, ey = ax == 1, ez = ax == 2;
            const T up = U(x + ex * (pos < len - 1), y + ey * (pos < len - 1), z + ez * (pos < len - 1));
            const T dn = U(x - ex * (pos > 0), y - ey * (pos > 0), z - ez * (pos > 0));
            // forward difference (backward at the last index) and backward difference (forward at index 0)
            const float df = pos < len - 1 ? (float)(up - u) : (float)(u - dn);
            const float db = pos > 0 ? (float)(u - dn) : (float)(up - u);
            const float V = ax == 0 ? Vx[i] : (ax == 1 ? Vy[i] : Vz[i]);
            const float flag = V > 0.f ? 1.f : 0.f;
            const float d = df * (1.f - flag) + db * flag;
            const float term = V * d;
            acc = first ? term : acc + term;
            first = false;
        }
        out[i] = -acc;
    }
}

// ------------------------------------------------------------------ Runge-Kutta tensor arithmetic
struct KSet { const float* k[7]; float c[7]; int nk; };

template <typename T>
__global__ void rk_combine(const T* __restrict__ y0, KSet ks, T* __restrict__ out, int64_t n) {
    GRID_STRIDE(i, n) {
        float acc = ks.c[0] * ks.k[0][i];
        for (int j = 1; j < ks.nk; ++j) acc = acc + ks.c[j] * ks.k[j][i];
        out[i] = y0 ? (T)(y0[i] + (T)acc) : (T)acc;
    }
}

// sum over i of (err_i / (atol + rtol*max(|y0|,|y1|)))^2, err = sum_j c_j k_j   -> per-block partials (fp64)
template <typename T>
__global__ void rk_error_partial(KSet ks, const T* __restrict__ y0, const T* __restrict__ y1, double atol, double rtol,
                                 int64_t n, double* __restrict__ part) {
    double s = 0.0;
    GRID_STRIDE(i, n) {
        float e = ks.c[0] * ks.k[0][i];
        for (int j = 1; j < ks.nk; ++j) e = e + ks.c[j] * ks.k[j][i];
        const T tol = (T)atol + (T)rtol * (T)fmax((double)fabs((double)y0[i]), (double)fabs((double)y1[i]));
        const T r = (T)e / tol;
        s += (double)(r * r);
    }
    __shared__ double red[4];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// sum of ((a - b) / (atol + rtol*|y0|))^2  (b may be null)
template <typename TA, typename T>
__global__ void scaled_sumsq_partial(const TA* __restrict__ a, const TA* __restrict__ b, const T* __restrict__ y0,
                                     double atol, double rtol, int64_t n, double* __restrict__ part) {
    double s = 0.0;
    GRID_STRIDE(i, n) {
        const T scale = (T)atol + (T)fabs((double)y0[i]) * (T)rtol;
        const T num = b ? (T)(a[i] - b[i]) : (T)a[i];
        const double r = (double)(num / scale);
        s += r * r;
    }
    __shared__ double red[4];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void sum_partials(const double* __restrict__ part, int nb, double* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// generic reductions for the augmentation chain: op 0 min, 1 max, 2 sum(x), 3 sum(x*y)
template <typename TX>
__device__ __forceinline__ double red_step(int op, double s, TX x, TX y) {
    const double v = op == 3 ? (double)(x * y) : (double)x;
    return op == 0 ? fmin(s, v) : (op == 1 ? fmax(s, v) : s + v);
}

template <typename TX>
__global__ void reduce_partial(int op, const TX* __restrict__ x, const TX* __restrict__ y, int64_t n,
                               double* __restrict__ part) {
    double s = op == 0 ? INFINITY : (op == 1 ? -INFINITY : 0.0);
    constexpr int V = 16 / sizeof(TX);                                  // elements per 16-byte access
    typedef TX VT __attribute__((ext_vector_type(V)));
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (!y || (reinterpret_cast<uintptr_t>(y) & 15) == 0);
    int64_t done = 0;
    if (vec) {
        const int64_t nv = n / V;
        const VT* xv = reinterpret_cast<const VT*>(x);
        const VT* yv = reinterpret_cast<const VT*>(y);
        GRID_STRIDE(i, nv) {
            const VT a = xv[i];
            VT b = a;
            if (op == 3) b = yv[i];
#pragma unroll
            for (int k = 0; k < V; ++k) s = red_step<TX>(op, s, a[k], b[k]);
        }
        done = nv * V;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s = red_step<TX>(op, s, x[i], op == 3 ? y[i] : x[i]);
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(s, o, 64);
        s = op == 0 ? fmin(s, t) : (op == 1 ? fmax(s, t) : s + t);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int w = 1; w < 4; ++w) r = op == 0 ? fmin(r, red[w]) : (op == 1 ? fmax(r, red[w]) : r + red[w]);
        part[blockIdx.x] = r;
    }
}

// one wave: lane l folds its strided partials in order, then a fixed xor tree (deterministic; the serial fold of 64 lane
// results on lane 0 cost 6 us per reduction)
__global__ void reduce_final(int op, const double* __restrict__ part, int nb, double* __restrict__ out) {
    const int l = threadIdx.x;
    double r = op == 0 ? INFINITY : (op == 1 ? -INFINITY : 0.0);
    for (int i = l; i < nb; i += 64) r = op == 0 ? fmin(r, part[i]) : (op == 1 ? fmax(r, part[i]) : r + part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(r, o, 64);
        r = op == 0 ? fmin(r, t) : (op == 1 ? fmax(r, t) : r + t);
    }
    if (l == 0) out[0] = r;
}

// _interp_fit_dopri5 + _interp_evaluate fused (dopri5.py:41-47, interp.py:5-65)
template <typename T>
__global__ void dense_eval(const T* __restrict__ y0, const T* __restrict__ y1, KSet mid, T dt, T x,
                           T* __restrict__ out, int64_t n) {
    const T x2 = x * x, x3 = x2 * x, x4 = x3 * x;
    GRID_STRIDE(i, n) {
        float m = mid.c[0] * mid.k[0][i];
        for (int j = 1; j < mid.nk; ++j) m = m + mid.c[j] * mid.k[j][i];
        const T a0 = y0[i], a1 = y1[i];
        const T ym = a0 + (T)m;
        const float f0 = mid.k[0][i], f1 = mid.k[mid.nk - 1][i];
        // _dot_product: sum(x*y) left to right; products of a 0-dim state-dtype scalar with an fp32 tensor are fp32
        // the two derivative terms are fp32 tensors in the reference and are added in fp32 first
        const T ca = ((((T)((float)(-2 * dt) * f0 + (float)(2 * dt) * f1)) + (T)(-8) * a0) + (T)(-8) * a1) + (T)16 * ym;
        const T cb = ((((T)((float)(5 * dt) * f0 + (float)(-3 * dt) * f1)) + (T)18 * a0) + (T)14 * a1) + (T)(-32) * ym;
        const T cc = ((((T)((float)(-4 * dt) * f0 + (float)dt * f1)) + (T)(-11) * a0) + (T)(-5) * a1) + (T)16 * ym;
        const float cd = (float)dt * f0;
        out[i] = (((ca * x4 + cb * x3) + cc * x2) + (T)(cd * (float)x)) + a0 * (T)1;
    }
}

KSet make_kset(const bfm_kset_t* s) {
    KSet k{};
    k.nk = s->nk;
    for (int j = 0; j < 7; ++j) { k.k[j] = j < s->nk ? s->k[j] : nullptr; k.c[j] = j < s->nk ? s->coef[j] : 0.f; }
    return k;
}

bool kset_ok(const bfm_kset_t* s) {
    if (!s || s->nk < 1 || s->nk > 7) return false;
    for (int j = 0; j < s->nk; ++j) if (!s->k[j]) return false;
    return true;
}

constexpr int RED_BLOCKS = 512;

}  // namespace

extern "C" int bfm_perlin3d(const double* grad, int sx, int sy, int sz, int rx, int ry, int rz, double* out,
                            bfm_stream_t stream) {
    if (!grad || !out || sx <= 0 || sy <= 0 || sz <= 0 || rx <= 0 || ry <= 0 || rz <= 0) return BFM_E_ARG;
    if (sx % rx || sy % ry || sz % rz) return BFM_E_SHAPE;          // "shape must be a multiple of res"
    if ((int64_t)sx * sy > INT32_MAX) return BFM_E_SHAPE;
    hipLaunchKernelGGL(perlin3d, dim3(grid_for((int64_t)sx * sy, 4)), dim3(256), 0, bfm_s(stream), grad, sx, sy, sz, rx, ry,
                       rz, out);
    return bfm_launch_status();
}

extern "C" int bfm_radix_hist_f64(const double* x, int64_t n, uint64_t prefix, int shift, uint32_t* hist65536,
                                  bfm_stream_t stream) {
    if (!x || !hist65536 || n <= 0 || shift < 0 || shift > 48 || (shift % 16)) return BFM_E_ARG;
    hipLaunchKernelGGL(radix_hist, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), x, n, prefix, shift, hist65536);
    return bfm_launch_status();
}

extern "C" int bfm_threshold_mask_f64(const double* x, int64_t n, double thr, double* masked, double* mask,
                                      bfm_stream_t stream) {
    if (!x || !masked || !mask || n <= 0) return BFM_E_ARG;
    hipLaunchKernelGGL(threshold_mask, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), x, n, thr, masked, mask);
    return bfm_launch_status();
}

extern "C" int bfm_curl3d(const double* a, const double* b, const double* c, int sx, int sy, int sz, float mult,
                          float* Vx, float* Vy, float* Vz, bfm_stream_t stream) {
    if (!a || !b || !c || !Vx || !Vy || !Vz || sx < 2 || sy < 2 || sz < 2) return BFM_E_ARG;
    int64_t n = (int64_t)sx * sy * sz;
    hipLaunchKernelGGL(curl3d, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), a, b, c, sx, sy, sz, mult, Vx, Vy, Vz);
    return bfm_launch_status();
}

extern "C" int bfm_advect_upwind_rhs(const void* C, int c_is_f64, const float* Vx, const float* Vy, const float* Vz,
                                     int sx, int sy, int sz, int neumann_bc, float* out, bfm_stream_t stream) {
    if (!C || !Vx || !Vy || !Vz || !out) return BFM_E_ARG;
    if (sx < 3 || sy < 3 || sz < 3) return BFM_E_SHAPE;
    int64_t n = (int64_t)sx * sy * sz;
    if (c_is_f64)
        hipLaunchKernelGGL(advect_rhs<double>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const double*)C, Vx, Vy,
                           Vz, sx, sy, sz, neumann_bc, out);
    else
        hipLaunchKernelGGL(advect_rhs<float>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const float*)C, Vx, Vy, Vz,
                           sx, sy, sz, neumann_bc, out);
    return bfm_launch_status();
}

extern "C" int bfm_rk_combine(const void* y0, int is_f64, const bfm_kset_t* ks, void* out, int64_t n,
                              bfm_stream_t stream) {
    if (!kset_ok(ks) || !out || n <= 0) return BFM_E_ARG;
    KSet k = make_kset(ks);
    if (is_f64)
        hipLaunchKernelGGL(rk_combine<double>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const double*)y0, k,
                           (double*)out, n);
    else
        hipLaunchKernelGGL(rk_combine<float>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const float*)y0, k,
                           (float*)out, n);
    return bfm_launch_status();
}

extern "C" size_t bfm_reduce_workspace(void) { return (size_t)RED_BLOCKS * sizeof(double); }

extern "C" int bfm_rk_error_sumsq(const bfm_kset_t* ks, const void* y0, const void* y1, int is_f64, double atol,
                                  double rtol, int64_t n, double* out, void* workspace, size_t workspace_bytes,
                                  bfm_stream_t stream) {
    if (!kset_ok(ks) || !y0 || !y1 || !out || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_reduce_workspace()) return BFM_E_WORKSPACE;
    KSet k = make_kset(ks);
    const int nb = grid_for(n, 256, RED_BLOCKS);
    double* part = static_cast<double*>(workspace);
    if (is_f64)
        hipLaunchKernelGGL(rk_error_partial<double>, dim3(nb), dim3(256), 0, bfm_s(stream), k, (const double*)y0,
                           (const double*)y1, atol, rtol, n, part);
    else
        hipLaunchKernelGGL(rk_error_partial<float>, dim3(nb), dim3(256), 0, bfm_s(stream), k, (const float*)y0,
                           (const float*)y1, atol, rtol, n, part);
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, bfm_s(stream), part, nb, out);
    return bfm_launch_status();
}

extern "C" int bfm_scaled_sumsq(const void* a, const void* b, int ab_is_f64, const void* y0, int y_is_f64, double atol,
                                double rtol, int64_t n, double* out, void* workspace, size_t workspace_bytes,
                                bfm_stream_t stream) {
    if (!a || !y0 || !out || !workspace || n <= 0) return BFM_E_ARG;
    if (workspace_bytes < bfm_reduce_workspace()) return BFM_E_WORKSPACE;
    const int nb = grid_for(n, 256, RED_BLOCKS);
    double* part = static_cast<double*>(workspace);
    hipStream_t st = bfm_s(stream);
    if (ab_is_f64 && y_is_f64)
        hipLaunchKernelGGL((scaled_sumsq_partial<double, double>), dim3(nb), dim3(256), 0, st, (const double*)a,
                           (const double*)b, (const double*)y0, atol, rtol, n, part);
    else if (!ab_is_f64 && y_is_f64)
        hipLaunchKernelGGL((scaled_sumsq_partial<float, double>), dim3(nb), dim3(256), 0, st, (const float*)a,
                           (const float*)b, (const double*)y0, atol, rtol, n, part);
    else if (!ab_is_f64 && !y_is_f64)
        hipLaunchKernelGGL((scaled_sumsq_partial<float, float>), dim3(nb), dim3(256), 0, st, (const float*)a,
                           (const float*)b, (const float*)y0, atol, rtol, n, part);
    else
        return BFM_E_SHAPE;
    hipLaunchKernelGGL(sum_partials, dim3(1), dim3(256), 0, st, part, nb, out);
    return bfm_launch_status();
}

extern "C" int bfm_reduce_f32(int op, const float* x, const float* y, int64_t n, double* out, void* workspace,
                              size_t workspace_bytes, bfm_stream_t stream) {
    if (!x || !out || !workspace || n <= 0 || op < 0 || op > 3 || (op == 3 && !y)) return BFM_E_ARG;
    if (workspace_bytes < bfm_reduce_workspace()) return BFM_E_WORKSPACE;
    const int nb = grid_for(n, 256, RED_BLOCKS);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(reduce_partial<float>, dim3(nb), dim3(256), 0, bfm_s(stream), op, x, y, n, part);
    hipLaunchKernelGGL(reduce_final, dim3(1), dim3(64), 0, bfm_s(stream), op, part, nb, out);
    return bfm_launch_status();
}

extern "C" int bfm_reduce_f64(int op, const double* x, const double* y, int64_t n, double* out, void* workspace,
                              size_t workspace_bytes, bfm_stream_t stream) {
    if (!x || !out || !workspace || n <= 0 || op < 0 || op > 3 || (op == 3 && !y)) return BFM_E_ARG;
    if (workspace_bytes < bfm_reduce_workspace()) return BFM_E_WORKSPACE;
    const int nb = grid_for(n, 256, RED_BLOCKS);
    double* part = static_cast<double*>(workspace);
    hipLaunchKernelGGL(reduce_partial<double>, dim3(nb), dim3(256), 0, bfm_s(stream), op, x, y, n, part);
    hipLaunchKernelGGL(reduce_final, dim3(1), dim3(64), 0, bfm_s(stream), op, part, nb, out);
    return bfm_launch_status();
}

extern "C" int bfm_dopri5_dense_eval(const void* y0, const void* y1, int is_f64, const bfm_kset_t* mid, double dt,
                                     double x, void* out, int64_t n, bfm_stream_t stream) {
    if (!y0 || !y1 || !kset_ok(mid) || !out || n <= 0) return BFM_E_ARG;
    KSet k = make_kset(mid);
    if (is_f64)
        hipLaunchKernelGGL(dense_eval<double>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const double*)y0,
                           (const double*)y1, k, dt, x, (double*)out, n);
    else
        hipLaunchKernelGGL(dense_eval<float>, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), (const float*)y0,
                           (const float*)y1, k, (float)dt, (float)x, (float*)out, n);
    return bfm_launch_status();
}
