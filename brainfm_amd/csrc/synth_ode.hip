// Dormand-Prince integration of the advection PDE with the step controller ON the device
// (ShapeID/DiffEqs/dopri5.py:58-172, rk_common.py:22-61, misc.py:145-170, interp.py:5-65; RHS: pde.py:616-640).
//
// Round 1-3: every stage was two launches (rk_combine, advect_rhs), every step ended in a host read-back of the error
// norm and host arithmetic for accept / reject / next step (13 launches + 1 sync per step, Python-bound).  Here
//   * a stage is ONE kernel: the stage state y + dt * sum_j beta_ij k_j is evaluated at the 7 stencil points straight
//     from y and the k_j (the same expressions as rk_combine, so the same bits) -- the stage state is never written;
//   * the error norm, accept / reject, the clamped next step (dopri5.py:150-169) and the dense output at the requested
//     times are computed by kernels that read and update a small state block in device memory;
//   * the host enqueues steps in chunks and reads back one flag per chunk; kernels of steps past the end exit at once.
// The state keeps y0's dtype (fp64 for Perlin shapes, fp32 for file maps) with fp32 stages, like the reference.
#include "bfm_common.h"

namespace {

inline int grid_for(int64_t n, int tpb = 256, int cap = 8192) {
    int64_t b = bfm_cdiv64(n, tpb);
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

struct OdeState {
    double t0s, t1s, dt, step_dt;      // last accepted interval, next step size, size of the step just taken
    double msr;                        // mean squared error ratio of the step just taken
    int cur, done, next_out, nsteps, naccept, accepted, err, pad;
};

struct OdeArgs {
    void* y[2]; float* f[2]; float* k[5];
    const float *Vx, *Vy, *Vz;
    int sx, sy, sz, neumann;
    OdeState* st;
};

__constant__ double BETA[6][6] = {
    {1 / 5., 0, 0, 0, 0, 0},
    {3 / 40., 9 / 40., 0, 0, 0, 0},
    {44 / 45., -56 / 15., 32 / 9., 0, 0, 0},
    {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729., 0, 0},
    {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656., 0},
    {35 / 384., 0, 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
__constant__ double C_ERR[7] = {35 / 384. - 1951 / 21600., 0, 500 / 1113. - 22642 / 50085., 125 / 192. - 451 / 720.,
                                -2187 / 6784. - -12231 / 42400., 11 / 84. - 649 / 6300., -1. / 60.};
__constant__ double C_MID[7] = {6025192743 / 30085553152. / 2, 0, 51252292925 / 65400821598. / 2,
                                -2691868925 / 45128329728. / 2, 187940372067 / 1594534317056. / 2,
                                -1776094331 / 19743644256. / 2, 11237099 / 235043384. / 2};

// (dt * c) in the state dtype, then the fp32 value torch's 0-dim promotion multiplies the fp32 stage with (misc.py:22-25)
template <typename T>
__device__ __forceinline__ float coef(double dt, double c) { return (float)((T)dt * (T)c); }

// k_STAGE = f(y + dt * sum_{j < STAGE} beta[STAGE-1][j] k_j).  A workgroup owns an 8 x 8 x 32 box of voxels: the stage
// state of the box and its one-voxel halo (10 x 10 x 34 points) is evaluated once into LDS -- the expression of rk_combine
// at the boundary-conditioned position, so the same bits --, then every thread forms the upwind RHS of its 8 voxels from
// LDS.  The stage state never goes to HBM; the k_j are read 1.66x (the halo) instead of 7x (one evaluation per stencil
// point, first form of this kernel: 415 us per step against 283 for the separate rk_combine + advect_rhs launches).
// STAGE 6 also stores y1 (the UNconditioned combination) and f1, and leaves the block partials of the error norm
// (rk_error_partial's expression: err = sum_j c_j k_j with k_6 = the value just computed).
constexpr int TX = 8, TY = 8, TZ = 32;
constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;

template <typename T, int STAGE>
__global__ void __launch_bounds__(256) ode_stage(OdeArgs A, double atol, double rtol, double* __restrict__ part) {
    const OdeState* st = A.st;
    if (st->done) return;
    __shared__ T sU[HX * HY * HZ];
    __shared__ float sE[STAGE == 6 ? TX * TY * TZ : 1];             // stage 6: sum_{j<6} c_err[j] k_j at the box's voxels
    const int cur = st->cur;
    const T* __restrict__ y = static_cast<const T*>(A.y[cur]);
    const float* ks[6] = {A.f[cur], A.k[0], A.k[1], A.k[2], A.k[3], A.k[4]};
    float* __restrict__ out = STAGE == 6 ? A.f[cur ^ 1] : A.k[STAGE - 1];
    T* __restrict__ ynext = static_cast<T*>(A.y[cur ^ 1]);
    float c[STAGE];
#pragma unroll
    for (int j = 0; j < STAGE; ++j) c[j] = coef<T>(st->dt, BETA[STAGE - 1][j]);
    float ce[7];
    if (STAGE == 6) {
#pragma unroll
        for (int j = 0; j < 7; ++j) ce[j] = coef<T>(st->dt, C_ERR[j]);
    }
    const int sx = A.sx, sy = A.sy, sz = A.sz;
    const int64_t stx = (int64_t)sy * sz, sty = sz;
    const int nbz = (sz + TZ - 1) / TZ, nby = (sy + TY - 1) / TY;
    const int bz = blockIdx.x % nbz, by = (blockIdx.x / nbz) % nby, bx = blockIdx.x / (nbz * nby);
    const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
    // ---- phase 1: the stage state on the box + halo
    for (int p = threadIdx.x; p < HX * HY * HZ; p += 256) {
        const int hz = p % HZ, hy = (p / HZ) % HY, hx = p / (HZ * HY);
        int a = x0 + hx - 1, b = y0 + hy - 1, cc = z0 + hz - 1;
        if (A.neumann) {
            a = min(max(a, 1), sx - 2); b = min(max(b, 1), sy - 2); cc = min(max(cc, 1), sz - 2);
        } else {
            a = min(max(a, 0), sx - 1); b = min(max(b, 0), sy - 1); cc = min(max(cc, 0), sz - 1);   // never used beyond the faces
        }
        const int64_t q = a * stx + b * sty + cc;
        float kv[STAGE];
#pragma unroll
        for (int j = 0; j < STAGE; ++j) kv[j] = ks[j][q];
        float acc = c[0] * kv[0];
#pragma unroll
        for (int j = 1; j < STAGE; ++j) acc = acc + c[j] * kv[j];
        sU[p] = (T)(y[q] + (T)acc);
        if (STAGE == 6 && hx >= 1 && hx <= TX && hy >= 1 && hy <= TY && hz >= 1 && hz <= TZ) {
            // interior point of the halo box: q is the voxel itself unless a Neumann face moved it (those are redone below)
            float e = ce[0] * kv[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) e = e + ce[j] * kv[j];
            sE[((hx - 1) * TY + (hy - 1)) * TZ + (hz - 1)] = e;
        }
    }
    __syncthreads();
    // ---- phase 2: upwind right-hand side of the box's voxels
    const int tz = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int yy = y0 + ty, z = z0 + tz;
    double es = 0.0;
    if (yy < sy && z < sz) {
#pragma unroll 2
        for (int tx = 0; tx < TX; ++tx) {
            const int x = x0 + tx;
            if (x >= sx) break;
            const int64_t i = x * stx + yy * sty + z;
            auto U = [&](int dx, int dy, int dz) -> T { return sU[((tx + 1 + dx) * HY + (ty + 1 + dy)) * HZ + (tz + 1 + dz)]; };
            const T u = U(0, 0, 0);
            float acc = 0.f;
            bool first = true;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const int pos = ax == 0 ? x : (ax == 1 ? yy : z);
                const int len = ax == 0 ? sx : (ax == 1 ? sy : sz);
                const int ex = ax == 0, ey = ax == 1, ez = ax == 2;
                const int fw = pos < len - 1 ? 1 : 0, bw = pos > 0 ? 1 : 0;
                const T up = U(ex * fw, ey * fw, ez * fw);
                const T dn = U(-ex * bw, -ey * bw, -ez * bw);
                const float df = pos < len - 1 ? (float)(up - u) : (float)(u - dn);
                const float db = pos > 0 ? (float)(u - dn) : (float)(up - u);
                const float V = ax == 0 ? A.Vx[i] : (ax == 1 ? A.Vy[i] : A.Vz[i]);
                const float flag = V > 0.f ? 1.f : 0.f;
                const float d = df * (1.f - flag) + db * flag;
                const float term = V * d;
                acc = first ? term : acc + term;
                first = false;
            }
            const float kn = -acc;
            out[i] = kn;
            if (STAGE == 6) {
                // y1 is the UNconditioned combination at i (rk_combine); the LDS value is boundary-conditioned, which
                // differs on the faces under Neumann conditions
                const bool face = A.neumann && (x == 0 || x == sx - 1 || yy == 0 || yy == sy - 1 || z == 0 || z == sz - 1);
                T y1v = u;
                if (face) {
                    float a6 = c[0] * ks[0][i];
#pragma unroll
                    for (int j = 1; j < STAGE; ++j) a6 = a6 + c[j] * ks[j][i];
                    y1v = (T)(y[i] + (T)a6);
                }
                ynext[i] = y1v;
                float e = sE[(tx * TY + ty) * TZ + tz];
                if (face) {
                    e = ce[0] * ks[0][i];
#pragma unroll
                    for (int j = 1; j < 6; ++j) e = e + ce[j] * ks[j][i];
                }
                e = e + ce[6] * kn;
                const T tol = (T)atol + (T)rtol * (T)fmax((double)fabs((double)y[i]), (double)fabs((double)y1v));
                const T r = (T)e / tol;
                es += (double)(r * r);
            }
        }
    }
    if (STAGE == 6) {
        __shared__ double red[4];
        es = wave_reduce_sum(es);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = es;
        __syncthreads();
        if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

struct CtlP { double tol_min_dt, dt_max, safety, ifactor, dfactor; int64_t n; };

// one block: fold the boxes' partials of the error norm (fixed order), then _optimal_step_size + the forced-accept clamps of
// dopri5.py:150-169 on thread 0
__global__ void ode_control(OdeState* st, const double* __restrict__ part, int nb, CtlP P) {
    if (st->done) return;
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double total = (red[0] + red[1]) + (red[2] + red[3]);
    const double msr = total / (double)P.n;
    const double dt = st->dt;
    if (!(st->t1s + dt > st->t1s)) { st->err = 1; st->done = 1; return; }       // 'underflow in dt'
    const bool accept = msr <= 1;
    double dt_next;
    if (msr == 0) {
        dt_next = dt * P.ifactor;
    } else {
        const double dfactor = msr < 1 ? 1.0 : P.dfactor;
        // Python's min / max keep their first argument unless the second compares smaller / larger: a NaN error norm
        // (a diverged state) gives factor = 1 / ifactor there, and must here
        const double a = pow(sqrt(msr), 1.0 / 5.0) / P.safety, b = 1 / dfactor;
        const double inner = b < a ? b : a;
        const double factor = inner > 1 / P.ifactor ? inner : 1 / P.ifactor;
        dt_next = dt / factor;
    }
    bool advance;
    if (!(dt_next < P.tol_min_dt || dt_next > P.dt_max)) {
        advance = accept;
    } else {
        dt_next = dt_next < P.tol_min_dt ? P.tol_min_dt : dt_next;
        dt_next = dt_next > P.dt_max ? P.dt_max : dt_next;
        advance = true;
    }
    st->msr = msr;
    st->accepted = advance ? 1 : 0;
    if (advance) {
        st->step_dt = dt;
        st->t0s = st->t1s;
        st->t1s = st->t1s + dt;
        st->cur ^= 1;
        st->naccept += 1;
    }
    st->dt = dt_next;
    st->nsteps += 1;
}

// _interp_fit_dopri5 + _interp_evaluate (dopri5.py:41-47, interp.py:5-65) at every requested time inside the interval just
// accepted; the expressions of dense_eval (synth_shape.hip)
template <typename T>
__global__ void ode_dense(OdeArgs A, const double* __restrict__ t_out, int nt, T* __restrict__ sol, int64_t n) {
    const OdeState* st = A.st;
    if (st->done || !st->accepted) return;
    const int cur = st->cur;                                            // already the NEW state
    const T* __restrict__ y0 = static_cast<const T*>(A.y[cur ^ 1]);
    const T* __restrict__ y1 = static_cast<const T*>(A.y[cur]);
    const float* ks[7] = {A.f[cur ^ 1], A.k[0], A.k[1], A.k[2], A.k[3], A.k[4], A.f[cur]};
    const double t0 = st->t0s, t1 = st->t1s;
    const T dt = (T)st->step_dt;
    float cm[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) cm[j] = coef<T>(st->step_dt, C_MID[j]);
    for (int o = st->next_out; o < nt && t_out[o] <= t1; ++o) {
        const T x = (T)(((T)t_out[o] - (T)t0) / ((T)t1 - (T)t0));
        const T x2 = x * x, x3 = x2 * x, x4 = x3 * x;
        T* __restrict__ dst = sol + (int64_t)o * n;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            float m = cm[0] * ks[0][i];
#pragma unroll
            for (int j = 1; j < 7; ++j) m = m + cm[j] * ks[j][i];
            const T a0 = y0[i], a1 = y1[i];
            const T ym = a0 + (T)m;
            const float f0 = ks[0][i], f1 = ks[6][i];
            const T ca = ((((T)((float)(-2 * dt) * f0 + (float)(2 * dt) * f1)) + (T)(-8) * a0) + (T)(-8) * a1) + (T)16 * ym;
            const T cb = ((((T)((float)(5 * dt) * f0 + (float)(-3 * dt) * f1)) + (T)18 * a0) + (T)14 * a1) + (T)(-32) * ym;
            const T cc = ((((T)((float)(-4 * dt) * f0 + (float)dt * f1)) + (T)(-11) * a0) + (T)(-5) * a1) + (T)16 * ym;
            const float cd = (float)dt * f0;
            dst[i] = (((ca * x4 + cb * x3) + cc * x2) + (T)(cd * (float)x)) + a0 * (T)1;
        }
    }
}

// after the dense outputs of a step: move next_out past the interval; the last requested time ends the integration
__global__ void ode_after_dense(OdeState* st, const double* __restrict__ t_out, int nt) {
    if (st->done || !st->accepted) return;
    int o = st->next_out;
    while (o < nt && t_out[o] <= st->t1s) ++o;
    st->next_out = o;
    st->accepted = 0;
    if (o >= nt) st->done = 1;
}

__global__ void ode_init(OdeState* st, double t0, double dt0) {
    st->t0s = t0; st->t1s = t0; st->dt = dt0; st->step_dt = 0; st->msr = 0;
    st->cur = 0; st->done = 0; st->next_out = 1; st->nsteps = 0; st->naccept = 0; st->accepted = 0; st->err = 0; st->pad = 0;
}


bool ode_ok(const bfm_dopri5_advect_t* d) {
    if (!d || !d->y[0] || !d->y[1] || !d->f[0] || !d->f[1] || !d->Vx || !d->Vy || !d->Vz || !d->state || !d->workspace ||
        !d->t_out || !d->sol || d->nt < 2)
        return false;
    for (int j = 0; j < 5; ++j) if (!d->k[j]) return false;
    return d->sx >= 3 && d->sy >= 3 && d->sz >= 3 && (int64_t)d->sx * d->sy <= INT32_MAX;
}

OdeArgs ode_args(const bfm_dopri5_advect_t* d) {
    OdeArgs A;
    A.y[0] = d->y[0]; A.y[1] = d->y[1]; A.f[0] = d->f[0]; A.f[1] = d->f[1];
    for (int j = 0; j < 5; ++j) A.k[j] = d->k[j];
    A.Vx = d->Vx; A.Vy = d->Vy; A.Vz = d->Vz;
    A.sx = d->sx; A.sy = d->sy; A.sz = d->sz; A.neumann = d->neumann_bc ? 1 : 0;
    A.st = static_cast<OdeState*>(d->state);
    return A;
}

int ode_boxes(const bfm_dopri5_advect_t* d) {
    return ((d->sx + TX - 1) / TX) * ((d->sy + TY - 1) / TY) * ((d->sz + TZ - 1) / TZ);
}

template <typename T>
void ode_enqueue_step(const bfm_dopri5_advect_t* d, const OdeArgs& A, hipStream_t s) {
    const int64_t n = (int64_t)d->sx * d->sy * d->sz;
    const int nb = ode_boxes(d);
    const dim3 gs(nb), bs(256);
    double* part = static_cast<double*>(d->workspace);
    hipLaunchKernelGGL((ode_stage<T, 1>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    hipLaunchKernelGGL((ode_stage<T, 2>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    hipLaunchKernelGGL((ode_stage<T, 3>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    hipLaunchKernelGGL((ode_stage<T, 4>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    hipLaunchKernelGGL((ode_stage<T, 5>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    hipLaunchKernelGGL((ode_stage<T, 6>), gs, bs, 0, s, A, d->atol, d->rtol, part);
    CtlP P{d->tol_min_dt, d->dt_max, d->safety, d->ifactor, d->dfactor, n};
    hipLaunchKernelGGL(ode_control, dim3(1), bs, 0, s, A.st, part, nb, P);
    hipLaunchKernelGGL(ode_dense<T>, dim3(grid_for(n)), bs, 0, s, A, d->t_out, d->nt, static_cast<T*>(d->sol), n);
    hipLaunchKernelGGL(ode_after_dense, dim3(1), dim3(1), 0, s, A.st, d->t_out, d->nt);
}

}  // namespace

extern "C" size_t bfm_dopri5_advect_state_bytes(void) { return sizeof(OdeState); }
extern "C" size_t bfm_dopri5_advect_workspace(int sx, int sy, int sz) {
    return (size_t)((sx + TX - 1) / TX) * ((sy + TY - 1) / TY) * ((sz + TZ - 1) / TZ) * sizeof(double);
}

extern "C" int bfm_dopri5_advect_init(const bfm_dopri5_advect_t* d, double t0, double dt0, bfm_stream_t stream) {
    if (!ode_ok(d) || !(dt0 > 0)) return BFM_E_ARG;
    hipLaunchKernelGGL(ode_init, dim3(1), dim3(1), 0, bfm_s(stream), static_cast<OdeState*>(d->state), t0, dt0);
    return bfm_launch_status();
}

extern "C" int bfm_dopri5_advect_steps(const bfm_dopri5_advect_t* d, int nsteps, bfm_stream_t stream) {
    if (!ode_ok(d) || nsteps <= 0) return BFM_E_ARG;
    const OdeArgs A = ode_args(d);
    for (int s = 0; s < nsteps; ++s) {
        if (d->is_f64) ode_enqueue_step<double>(d, A, bfm_s(stream));
        else ode_enqueue_step<float>(d, A, bfm_s(stream));
    }
    return bfm_launch_status();
}
