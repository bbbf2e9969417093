"""Oracle for the data-synthesis kernels: NumPy restatement of
Generator/utils.py, ShapeID/perlin3d.py, ShapeID/misc.py, ShapeID/DiffEqs/*, utils/interpol/iso1.py
and the pieces of Generator/datasets.py that do arithmetic.  Test infrastructure only.

Arithmetic is written in the reference's operation order and dtype (fp32 where the reference
computes in fp32, fp64 where it runs NumPy float64), so most results are expected bit for bit.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- K10 / K11
def interp3d_linear(X, II, JJ, KK, default_value=0.0):
    """fast_3D_interp_torch(..., 'linear'), Generator/utils.py:140-192.  X: (nx,ny,nz[,C]) fp32."""
    X = np.asarray(X, dtype=F32)
    if X.ndim == 3:
        X = X[..., None]
    nx, ny, nz, C = X.shape
    II, JJ, KK = (np.asarray(a, dtype=F32) for a in (II, JJ, KK))
    ok = (II > 0) & (JJ > 0) & (KK > 0) & (II <= nx - 1) & (JJ <= ny - 1) & (KK <= nz - 1)
    IIv, JJv, KKv = II[ok], JJ[ok], KK[ok]
    fx = np.floor(IIv).astype(np.int64); cx = np.minimum(fx + 1, nx - 1)
    wcx = (IIv - fx.astype(F32))[..., None]; wfx = F32(1) - wcx
    fy = np.floor(JJv).astype(np.int64); cy = np.minimum(fy + 1, ny - 1)
    wcy = (JJv - fy.astype(F32))[..., None]; wfy = F32(1) - wcy
    fz = np.floor(KKv).astype(np.int64); cz = np.minimum(fz + 1, nz - 1)
    wcz = (KKv - fz.astype(F32))[..., None]; wfz = F32(1) - wcz
    c00 = X[fx, fy, fz] * wfx + X[cx, fy, fz] * wcx
    c01 = X[fx, fy, cz] * wfx + X[cx, fy, cz] * wcx
    c10 = X[fx, cy, fz] * wfx + X[cx, cy, fz] * wcx
    c11 = X[fx, cy, cz] * wfx + X[cx, cy, cz] * wcx
    c0 = c00 * wfy + c10 * wcy
    c1 = c01 * wfy + c11 * wcy
    c = c0 * wfz + c1 * wcz
    Y = np.zeros(II.shape + (C,), dtype=F32)
    Y[ok] = c
    Y[~ok] = F32(default_value)
    return Y[..., 0] if C == 1 else Y


def interp3d_nearest(X, II, JJ, KK):
    """fast_3D_interp_torch(..., 'nearest'), Generator/utils.py:124-138 (round half to even, clamp)."""
    X = np.asarray(X)
    squeeze = X.ndim == 3
    if squeeze:
        X = X[..., None]
    r = [np.clip(np.rint(np.asarray(a, dtype=F32)).astype(np.int64), 0, n - 1) for a, n in zip((II, JJ, KK), X.shape[:3])]
    Y = X[r[0], r[1], r[2]]
    return Y[..., 0] if squeeze else Y


# --------------------------------------------------------------------------- K12
def torch_cpu_arange_f32(start, end, step, vec=8):
    """torch.arange(start, end, step, dtype=float32) as ATen's CPU kernel evaluates it
    (RangeFactoriesKernel.cpp + cpu/Loops.h vectorized_loop): while at least 2*vec elements remain,
    two vectors of `vec` lanes are produced, each float(float(start + step*idx) + lane*step)
    -- the base is rounded to float first --; the remainder is float(start + step*idx).
    vec = 8 reproduces torch 2.10 CPU here (probed for lengths 8..135).  On CUDA the reference
    would accumulate in float instead; the CPU path is the oracle's definition."""
    cnt = int(math.ceil((end - start) / step))
    out = np.empty(cnt, dtype=F32)
    i = 0
    while cnt - i >= 2 * vec:
        for _ in range(2):
            base = np.float64(F32(start + step * i))
            out[i:i + vec] = (base + np.arange(vec, dtype=np.float64) * step).astype(F32)
            i += vec
    out[i:] = (start + step * np.arange(i, cnt, dtype=np.float64)).astype(F32)
    return out


def zoom_tables(n, factor):
    """Per-axis tables of myzoom_torch (Generator/utils.py:205-235)."""
    delta = (1.0 - factor) / (2.0 * factor)
    new = int(np.round(n * factor))
    v = torch_cpu_arange_f32(delta, delta + new / factor, 1.0 / factor)[:new]
    v = np.where(v < 0, F32(0), v)
    v = np.where(v > n - 1, F32(n - 1), v).astype(F32)
    f = np.floor(v).astype(np.int32)
    c = np.minimum(f + 1, n - 1).astype(np.int32)
    wc = (v - f.astype(F32)).astype(F32)
    wf = (F32(1) - wc).astype(F32)
    return f, c, wf, wc


def myzoom(X, factor):
    """myzoom_torch, Generator/utils.py:200-257 (three separable passes, x then y then z)."""
    X = np.asarray(X, dtype=F32)
    squeeze = X.ndim == 3
    if squeeze:
        X = X[..., None]
    factor = np.asarray(factor, dtype=np.float64)
    tx, ty, tz = (zoom_tables(X.shape[a], float(factor[a])) for a in range(3))
    t1 = tx[2][:, None, None, None] * X[tx[0]] + tx[3][:, None, None, None] * X[tx[1]]
    t2 = ty[2][None, :, None, None] * t1[:, ty[0]] + ty[3][None, :, None, None] * t1[:, ty[1]]
    Y = tz[2][None, None, :, None] * t2[:, :, tz[0]] + tz[3][None, None, :, None] * t2[:, :, tz[1]]
    Y = Y.astype(F32)
    return Y[..., 0] if squeeze else Y


# --------------------------------------------------------------------------- K13
def make_gaussian_kernel(sigma):
    """Generator/utils.py:74-82."""
    sl = int(np.ceil(3 * sigma))
    ts = np.linspace(-sl, sl, 2 * sl + 1).astype(F32)
    g = np.exp((-(ts / F32(sigma)) ** 2 / 2)).astype(F32)
    return (g / g.sum(dtype=F32)).astype(F32)


def gaussian_blur_3d(I, stds):
    """Generator/utils.py:84-94: three 1-D zero-padded correlations."""
    out = np.asarray(I, dtype=F32)
    for ax in range(3):
        if stds[ax] > 0:
            k = make_gaussian_kernel(stds[ax]).astype(np.float64)
            half = len(k) // 2
            pad = [(0, 0)] * 3
            pad[ax] = (half, half)
            P = np.pad(out.astype(np.float64), pad)
            acc = np.zeros(out.shape, dtype=np.float64)
            for j in range(len(k)):
                sl = [slice(None)] * 3
                sl[ax] = slice(j, j + out.shape[ax])
                acc += k[j] * P[tuple(sl)]
            out = acc.astype(F32)
    return out


# --------------------------------------------------------------------------- K14
def gamma_transform(I, gamma):
    """300 * (I/300) ** gamma in fp32 (Generator/utils.py:568-572)."""
    I = np.asarray(I, dtype=F32)
    return (F32(300.0) * np.power(I / F32(300.0), F32(gamma))).astype(F32)


def apply_bias_field(I, BFlog):
    return (np.asarray(I, F32) * np.exp(np.asarray(BFlog, F32))).astype(F32)


def add_noise(I, noise_std, randn):
    out = (np.asarray(I, F32) + F32(noise_std) * np.asarray(randn, F32)).astype(F32)
    out[out < 0] = 0
    return out


def synth_from_labels(G, mus, sigmas, randn):
    """generate_sample core, Generator/datasets.py:366-372: mus[Gr] + sigmas[Gr]*randn, clamp at 0."""
    G = np.asarray(G, dtype=F32).copy()
    G[G == 77] = 2
    Gr = np.rint(G).astype(np.int64)
    syn = (np.asarray(mus, F32)[Gr] + np.asarray(sigmas, F32)[Gr] * np.asarray(randn, F32)).astype(F32)
    syn[syn < 0] = 0
    return syn


def onehot_lut(S, lut, n_labels):
    """onehotmatrix[lut[S]] (Generator/utils.py:408-411): (x,y,z) int -> (x,y,z,n_labels) fp32."""
    idx = np.asarray(lut)[np.asarray(S, dtype=np.int64)]
    return np.eye(n_labels, dtype=F32)[idx]


# --------------------------------------------------------------------------- K15
def deform_grid(size, shp, A, c2, F=None):
    """BaseGen.deform_grid, Generator/datasets.py:264-303 (fp32)."""
    A = np.asarray(A, F32); c2 = np.asarray(c2, F32)
    xx, yy, zz = np.meshgrid(range(size[0]), range(size[1]), range(size[2]), sparse=False, indexing="ij")
    c = ((np.array(size) - 1) / 2).astype(F32)
    xc, yc, zc = xx.astype(F32) - c[0], yy.astype(F32) - c[1], zz.astype(F32) - c[2]
    if F is not None:
        F = np.asarray(F, F32)
        xx1, yy1, zz1 = xc + F[..., 0], yc + F[..., 1], zc + F[..., 2]
    else:
        xx1, yy1, zz1 = xc, yc, zc
    out = []
    for r in range(3):
        v = A[r, 0] * xx1 + A[r, 1] * yy1 + A[r, 2] * zz1 + c2[r]
        v = v.astype(F32)
        v[v < 0] = 0
        v[v > (shp[r] - 1)] = shp[r] - 1
        out.append(v)
    lo = [np.floor(v.min()) for v in out]
    hi = [1 + np.ceil(v.max()) for v in out]
    out = [(v - F32(l)).astype(F32) for v, l in zip(out, lo)]
    return out[0], out[1], out[2], [int(v) for v in lo], [int(v) for v in hi]


# --------------------------------------------------------------------------- K16
def perlin_interpolant(t):
    return t * t * t * (t * (t * 6 - 15) + 10)


def perlin_gradients(theta, phi, tileable=(False, False, False)):
    """Gradient table from the reference's two uniform draws (ShapeID/perlin3d.py:44-55)."""
    g = np.stack((np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)), axis=3)
    if tileable[0]:
        g[-1, :, :] = g[0, :, :]
    if tileable[1]:
        g[:, -1, :] = g[:, 0, :]
    if tileable[2]:
        g[:, :, -1] = g[:, :, 0]
    return g


def perlin_noise_3d(shape, res, gradients):
    """generate_perlin_noise_3d with the gradient table given (ShapeID/perlin3d.py:38-83), fp64."""
    delta = (res[0] / shape[0], res[1] / shape[1], res[2] / shape[2])
    d = (shape[0] // res[0], shape[1] // res[1], shape[2] // res[2])
    grid = np.mgrid[0:res[0]:delta[0], 0:res[1]:delta[1], 0:res[2]:delta[2]]
    grid = grid.transpose(1, 2, 3, 0) % 1
    g = gradients.repeat(d[0], 0).repeat(d[1], 1).repeat(d[2], 2)
    g000 = g[:-d[0], :-d[1], :-d[2]]; g100 = g[d[0]:, :-d[1], :-d[2]]
    g010 = g[:-d[0], d[1]:, :-d[2]]; g110 = g[d[0]:, d[1]:, :-d[2]]
    g001 = g[:-d[0], :-d[1], d[2]:]; g101 = g[d[0]:, :-d[1], d[2]:]
    g011 = g[:-d[0], d[1]:, d[2]:]; g111 = g[d[0]:, d[1]:, d[2]:]
    x, y, z = grid[..., 0], grid[..., 1], grid[..., 2]

    def dot(a, b, c, gg):
        return np.sum(np.stack((a, b, c), axis=3) * gg, 3)
    n000 = dot(x, y, z, g000); n100 = dot(x - 1, y, z, g100)
    n010 = dot(x, y - 1, z, g010); n110 = dot(x - 1, y - 1, z, g110)
    n001 = dot(x, y, z - 1, g001); n101 = dot(x - 1, y, z - 1, g101)
    n011 = dot(x, y - 1, z - 1, g011); n111 = dot(x - 1, y - 1, z - 1, g111)
    t = perlin_interpolant(grid)
    n00 = n000 * (1 - t[..., 0]) + t[..., 0] * n100
    n10 = n010 * (1 - t[..., 0]) + t[..., 0] * n110
    n01 = n001 * (1 - t[..., 0]) + t[..., 0] * n101
    n11 = n011 * (1 - t[..., 0]) + t[..., 0] * n111
    n0 = (1 - t[..., 1]) * n00 + t[..., 1] * n10
    n1 = (1 - t[..., 1]) * n01 + t[..., 1] * n11
    return (1 - t[..., 2]) * n0 + t[..., 2] * n1


def percentile_mask(noise, percentile):
    """perlin3d.py:84-90: threshold at np.percentile (linear), returns (noise*mask, mask)."""
    thr = np.percentile(noise, percentile)
    mask = np.zeros_like(noise)
    mask[noise >= thr] = 1.0
    return noise * mask, mask, thr


# --------------------------------------------------------------------------- K17
def gradient_c(X):
    """ShapeID/misc.py:198-259 (3-D, unbatched): central differences, one-sided at the faces;
    differences in X's dtype, stored as fp32."""
    X = np.asarray(X)
    dX = np.zeros(X.shape + (3,), dtype=F32)
    dX[1:-1, :, :, 0] = (X[2:] - X[:-2]) / 2
    dX[0, :, :, 0] = X[1] - X[0]
    dX[-1, :, :, 0] = X[-1] - X[-2]
    dX[:, 1:-1, :, 1] = (X[:, 2:] - X[:, :-2]) / 2
    dX[:, 0, :, 1] = X[:, 1] - X[:, 0]
    dX[:, -1, :, 1] = X[:, -1] - X[:, -2]
    dX[:, :, 1:-1, 2] = (X[:, :, 2:] - X[:, :, :-2]) / 2
    dX[:, :, 0, 2] = X[:, :, 1] - X[:, :, 0]
    dX[:, :, -1, 2] = X[:, :, -1] - X[:, :, -2]
    return dX


def stream_3d(a, b, c, multiplier=1):
    """stream_3D (ShapeID/misc.py:66-80) x V_multiplier (perlin3d.py:149-156)."""
    da, db, dc = gradient_c(a), gradient_c(b), gradient_c(c)
    Vx = dc[..., 1] - db[..., 2]
    Vy = da[..., 2] - dc[..., 0]
    Vz = db[..., 0] - da[..., 1]
    return (Vx * multiplier).astype(F32), (Vy * multiplier).astype(F32), (Vz * multiplier).astype(F32)


# --------------------------------------------------------------------------- K18
def _bc(C):
    """set_BC 'neumann': interior kept, faces = replicate pad of the interior (pde.py:587-597)."""
    return np.pad(C[1:-1, 1:-1, 1:-1], 1, mode="edge")


def _grad_f(U, ax):
    d = np.zeros(U.shape, dtype=F32)
    sl = lambda a, b: tuple(slice(a, b) if i == ax else slice(None) for i in range(3))
    d[sl(None, -1)] = U[sl(1, None)] - U[sl(None, -1)]
    d[sl(-1, None)] = U[sl(-1, None)] - U[sl(-2, -1)]
    return d


def _grad_b(U, ax):
    d = np.zeros(U.shape, dtype=F32)
    sl = lambda a, b: tuple(slice(a, b) if i == ax else slice(None) for i in range(3))
    d[sl(1, None)] = U[sl(1, None)] - U[sl(None, -1)]
    d[sl(0, 1)] = U[sl(1, 2)] - U[sl(0, 1)]
    return d


def advect_rhs(C, Vx, Vy, Vz, bc="neumann"):
    """AdvDiffPDE.forward with perf_pattern='adv', V_type='vector_div_free'
    (pde.py:616-640, :499-509, :301-328).  C: (s,r,c) fp32 or fp64; returns fp32."""
    U = _bc(C) if bc in ("neumann", "cauchy") else C
    out = None
    for ax, V in enumerate((Vx, Vy, Vz)):
        V = np.asarray(V, F32)
        df, db = _grad_f(U, ax), _grad_b(U, ax)
        flag = (V > 0).astype(F32)
        d = df * (F32(1) - flag) + db * flag
        term = V * d
        out = term if out is None else out + term
    return (-out).astype(F32)


# --------------------------------------------------------------------------- K19 (dopri5 as shipped)
_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1., 1.]
_BETA = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
         [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
         [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
          -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]
_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
          187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _sdp(dt, coef, ks, sdtype):
    """_scaled_dot_product (DiffEqs/misc.py:22-25): sum((dt*c)*k), (dt*c) in the state dtype then
    cast to k's fp32 by torch's 0-dim promotion rule, accumulated left to right in fp32."""
    acc = None
    for c, k in zip(coef, ks):
        term = F32(sdtype(dt) * sdtype(c)) * k
        acc = term if acc is None else acc + term
    return acc


def dopri5_integrate(rhs, y0, t, dt_cfg=0.1, rtol=1e-6, atol=1e-12, stats=None):
    """odeint_adjoint -> Dopri5Solver as shipped (adjoint.py:105-132, dopri5.py:58-172,
    rk_common.py:22-61, interp.py, misc.py:84-170), including the forced-accept step clamps.
    rhs(y) -> fp32 array; y0 fp32 or fp64 (the state keeps y0's dtype); t float64 times."""
    sd = y0.dtype.type
    nfe = [0]

    def f(y):
        nfe[0] += 1
        return rhs(y)

    def rms(x):
        x = np.asarray(x)
        return np.sqrt(np.sum(x.astype(np.float64) ** 2)) / (x.size ** 0.5)

    t = np.asarray(t, dtype=np.float64)
    f0 = f(y0)
    # _select_initial_step (order 4)
    scale = atol + np.abs(y0) * rtol
    d0, d1 = rms(y0 / scale), rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y1 = y0 + sd(h0) * f0
    f1 = f(y1)
    d2 = rms((f1 - f0) / scale) / h0
    if d1 <= 1e-15 and d2 <= 1e-15:
        h1 = max(1e-6, h0 * 1e-3)
    else:
        h1 = (0.01 / max(d1, d2)) ** (1. / 5.)
    dt = float(min(100 * h0, h1))
    y, fcur, t0s, t1s = y0, f0, t[0], t[0]
    interp = None
    sol = [y0]
    tol_min_dt = 0.2 * dt_cfg if 0.1 * dt_cfg >= 0.01 else 0.01
    nsteps = 0
    for ti in t[1:]:
        while ti > t1s:
            assert t1s + dt > t1s
            ks = [fcur]
            for beta in _BETA:
                yi = y + _sdp(dt, beta, ks, sd)
                ks.append(f(yi))
            y1 = yi
            f1 = ks[-1]
            err = _sdp(dt, _C_ERR, ks, sd)
            tol = atol + rtol * np.maximum(np.abs(y), np.abs(y1))
            ratio = err / tol
            msr = float(np.mean(ratio.astype(np.float64) ** 2)) if ratio.dtype == np.float64 else float(
                np.mean((ratio * ratio).astype(np.float64)))
            accept = msr <= 1
            # _optimal_step_size
            if msr == 0:
                dt_next = dt * 10.0
            else:
                dfactor = 1.0 if msr < 1 else 0.2
                er = math.sqrt(msr)
                factor = max(1 / 10.0, min(er ** (1 / 5) / 0.9, 1 / dfactor))
                dt_next = dt / factor
            if not (dt_next < tol_min_dt or dt_next > 0.1):
                if accept:
                    interp = (y, y1, ks, dt)
                    y, fcur, t0s, t1s = y1, f1, t1s, t1s + dt
            else:
                dt_next = tol_min_dt if dt_next < tol_min_dt else dt_next
                dt_next = 0.1 if dt_next > 0.1 else dt_next
                interp = (y, y1, ks, dt)
                y, fcur, t0s, t1s = y1, f1, t1s, t1s + dt
            dt = dt_next
            nsteps += 1
        sol.append(dense_eval(interp, t0s, t1s, ti, sd))
    if stats is not None:
        stats.update(nfe=nfe[0], nsteps=nsteps)
    return np.stack(sol)


def dense_eval(interp, t0, t1, t, sd):
    """_interp_fit_dopri5 + _interp_evaluate (dopri5.py:41-47, interp.py:5-65)."""
    y0, y1, ks, dt = interp
    dts = sd(dt)
    y_mid = y0 + _sdp(dt, _C_MID, ks, sd)
    f0, f1 = ks[0], ks[-1]

    def dotp(cs, xs):
        acc = None
        for c, x in zip(cs, xs):
            term = c * x
            acc = term if acc is None else acc + term
        return acc
    a = dotp([-2 * dts, 2 * dts, -8, -8, 16], [f0, f1, y0, y1, y_mid])
    b = dotp([5 * dts, -3 * dts, 18, 14, -32], [f0, f1, y0, y1, y_mid])
    c = dotp([-4 * dts, dts, -11, -5, 16], [f0, f1, y0, y1, y_mid])
    d = dts * f0
    e = y0
    x = sd((sd(t) - sd(t0)) / (sd(t1) - sd(t0)))
    xs = [sd(1), x]
    for _ in range(2, 5):
        xs.append(xs[-1] * x)
    return dotp([a, b, c, d, e], list(reversed(xs)))


# --------------------------------------------------------------------------- K20 interpol.grid_pull (linear)
BOUND = dict(zero=0, zeros=0, replicate=1, nearest=1, dct1=2, mirror=2, dct2=3, reflect=3, dst1=4, antimirror=4,
             dst2=5, antireflect=5, dft=6, wrap=6)


def _bound_index(i, n, b):
    i = i.copy()
    if b in (0, 1):
        return np.clip(i, 0, n - 1)
    if b in (3, 5):
        n2 = n * 2
        i = np.where(i < 0, n2 - 1 - np.mod(-i - 1, n2), np.mod(i, n2))
        return np.where(i >= n, n2 - 1 - i, i)
    if b == 2:
        if n == 1:
            return np.zeros_like(i)
        n2 = (n - 1) * 2
        i = np.mod(np.abs(i), n2)
        return np.where(i >= n, n2 - i, i)
    if b == 4:
        n2 = 2 * (n + 1)
        i = np.where(i < 0, -i - 2, i)
        i = np.mod(i, n2)
        i = np.where(i > n, n2 - 2 - i, i)
        i = np.where(i == -1, 0, i)
        return np.where(i == n, n - 1, i)
    if b == 6:
        return np.mod(i, n)
    return i


def _bound_sign(i, n, b):
    if b == 4:
        if n == 1:
            return None
        n2 = 2 * (n + 1)
        i = np.where(i < 0, n - 1 - i, i)
        i = np.mod(i, n2)
        x = np.where(i == 0, 0, 1)
        x = np.where(np.mod(i, n + 1) == n, 0, x)
        i = i // (n + 1)
        return np.where(np.mod(i, 2) > 0, -x, x)
    if b == 5:
        i = np.where(i < 0, n - 1 - i, i)
        i = i // n
        return np.where(np.mod(i, 2) > 0, -1, 1)
    if b == 0:
        return np.where((i < 0) | (i >= n), 0, 1)
    return None


def grid_pull_linear(inp, grid, bound="zero", extrapolate=False):
    """interpol.grid_pull(interpolation='linear') -> iso1.pull3d (utils/interpol/iso1.py:28-133,
    bounds.py:24-89, jit_utils.py:241-255).  inp (B,C,X,Y,Z) fp32, grid (B,oX,oY,oZ,3) fp32."""
    inp = np.asarray(inp, F32); grid = np.asarray(grid, F32)
    bnds = [BOUND[bound]] * 3 if isinstance(bound, str) else [BOUND[b] if isinstance(b, str) else int(b) for b in bound]
    ext = {False: 0, True: 1, "hist": 2}.get(extrapolate, extrapolate)
    B, Cc, nx, ny, nz = inp.shape
    oshape = grid.shape[1:4]
    g = grid.reshape(grid.shape[0], -1, 3)
    gx, gy, gz = g[..., 0], g[..., 1], g[..., 2]
    mask = None
    if ext in (0, 2):
        thr = 5e-2 if ext == 0 else 0.5 + 5e-2
        mask = ((gx > -thr) & (gx < nx - 1 + thr) & (gy > -thr) & (gy < ny - 1 + thr) &
                (gz > -thr) & (gz < nz - 1 + thr))
    ws, i0s, i1s, s0s, s1s = [], [], [], [], []
    for gq, n, b in ((gx, nx, bnds[0]), (gy, ny, bnds[1]), (gz, nz, bnds[2])):
        g0 = np.floor(gq).astype(np.int64)
        g1 = g0 + 1
        s1s.append(_bound_sign(g1, n, b)); s0s.append(_bound_sign(g0, n, b))
        i1s.append(_bound_index(g1, n, b)); i0s.append(_bound_index(g0, n, b))
        ws.append((gq - np.floor(gq)).astype(F32))
    flat = inp.reshape(B, Cc, -1)
    out = None
    one = F32(1)
    for cx in (0, 1):
        for cy in (0, 1):
            for cz in (0, 1):
                ix = (i1s if cx else i0s)[0]; iy = (i1s if cy else i0s)[1]; iz = (i1s if cz else i0s)[2]
                idx = iz + iy * nz + ix * (ny * nz)
                val = np.take_along_axis(flat, np.broadcast_to(idx[:, None, :], (B, Cc, idx.shape[-1])), axis=-1)
                sign = None
                for s in ((s1s if cx else s0s)[0], (s1s if cy else s0s)[1], (s1s if cz else s0s)[2]):
                    if s is not None:
                        sign = s if sign is None else sign * s
                if sign is not None:
                    val = val * sign[:, None, :].astype(F32)
                wx = ws[0] if cx else one - ws[0]
                wy = ws[1] if cy else one - ws[1]
                wz = ws[2] if cz else one - ws[2]
                val = val * ((wx * wy) * wz)[:, None, :]
                out = val if out is None else out + val
    if mask is not None:
        out = out * mask[:, None, :].astype(F32)
    return out.reshape(B, Cc, *oshape).astype(F32)


# --------------------------------------------------------------------------- a13
def deformed_atlas(brain_labels, regx, regy, regz, MNI, A):
    """get_deformed_atlas, utils/test_utils.py:45-57 (fp32)."""
    M = brain_labels > 0
    A = np.asarray(A, F32)
    xx, yy, zz = (F32(100) * r[M] for r in (regx, regy, regz))
    ii = A[0, 0] * xx + A[0, 1] * yy + A[0, 2] * zz + A[0, 3]
    jj = A[1, 0] * xx + A[1, 1] * yy + A[1, 2] * zz + A[1, 3]
    kk = A[2, 0] * xx + A[2, 1] * yy + A[2, 2] * zz + A[2, 3]
    vals = interp3d_linear(MNI, ii, jj, kk)
    out = np.zeros_like(regx, dtype=F32)
    out[M] = vals
    return out


# ----------------------------------------------------------------------------- cubic B-spline resize
def bspline3_prefilter_dct2(x):
    """utils/interpol/coeff.py:254-344 for order 3, DCT-II ('nearest' / 'dct2') conditions, every axis of a 3-D array
    (gain, initial condition :141-175, causal recursion, final condition :218-226, anticausal recursion); fp64."""
    import math
    z = math.sqrt(3.0) - 2.0
    c = np.array(x, dtype=np.float64)
    for axis in range(c.ndim):
        n = c.shape[axis]
        if n == 1:
            continue
        c = np.moveaxis(c, axis, 0).copy()
        c *= (1.0 - z) * (1.0 - 1.0 / z)
        polen = z ** n
        pole_last = polen * (1 + 1 / (z + polen * polen))
        i = np.arange(1, n - 1)
        w = z ** i + z ** (2 * n - 1 - i)
        c0 = np.tensordot(w, c[1:-1], axes=(0, 0)) + (c[0] + pole_last * c[-1])
        c[0] = c0 * (z / (1 - polen * polen)) + c[0]
        for k in range(1, n):
            c[k] += z * c[k - 1]
        c[-1] = c[-1] * (z / (z - 1))
        for k in range(n - 2, -1, -1):
            c[k] = z * (c[k + 1] - c[k])
        c = np.moveaxis(c, 0, axis)
    return c


def _bspline3_w(x):
    x = np.abs(x)
    return np.where(x < 1, (x * x * (x - 2.0) * 3.0 + 4.0) / 6.0, (2.0 - x) ** 3 / 6.0)


def resize_cubic_ref(x, shape, anchor="e"):
    """utils/interpol/resize.py:13-119 with interpolation=3, bound='dct2', prefilter=True, extrapolate=True, evaluated
    separably in fp64 (nd.py:36-142: nodes floor(g-1)..+3, weights splines.py:40-43, DCT-II index bounds.py:33-38)."""
    c = bspline3_prefilter_dct2(x)
    for axis in range(3):
        n, m = c.shape[axis], shape[axis]
        if anchor[0] == "e":
            scale = n / m
            g = (np.arange(m, dtype=np.float32) * np.float32(scale) + np.float32(0.5 * (scale - 1))).astype(np.float64)
        else:
            g = np.linspace(0, n - 1, m, dtype=np.float32).astype(np.float64)
        g0 = np.floor(g - 1)
        out = 0
        for node in range(4):
            idx = (g0 + node).astype(np.int64)
            n2 = 2 * n
            idx = np.where(idx < 0, n2 - 1 - ((-idx - 1) % n2), idx % n2)
            idx = np.where(idx >= n, n2 - 1 - idx, idx)
            wgt = _bspline3_w(g - g0 - node)
            shp = [1, 1, 1]
            shp[axis] = m
            out = out + np.take(c, idx, axis=axis) * wgt.reshape(shp)
        c = out
    return c


# ----------------------------------------------------------------------------- grid_push / grid_grad
def _corner_terms(grid, shape, bound, extrapolate):
    """Shared by the push / grad restatements: per-axis (index0, index1, sign0, sign1, weight) and the in-bounds mask
    of iso1's get_weights_and_indices / inbounds_mask_3d (utils/interpol/iso1.py, jit_utils.py:241-255)."""
    nx, ny, nz = shape
    bnds = [BOUND[bound]] * 3 if isinstance(bound, str) else ([int(bound)] * 3 if np.isscalar(bound) else
                                                               [BOUND[b] if isinstance(b, str) else int(b) for b in bound])
    ext = {False: 0, True: 1, "hist": 2}.get(extrapolate, extrapolate)
    g = np.asarray(grid, np.float64).reshape(grid.shape[0], -1, 3)
    gx, gy, gz = g[..., 0], g[..., 1], g[..., 2]
    mask = np.ones(gx.shape)
    if ext in (0, 2):
        thr = 5e-2 if ext == 0 else 0.5 + 5e-2
        mask = ((gx > -thr) & (gx < nx - 1 + thr) & (gy > -thr) & (gy < ny - 1 + thr) &
                (gz > -thr) & (gz < nz - 1 + thr)).astype(np.float64)
    axes = []
    for gq, n, b in ((gx, nx, bnds[0]), (gy, ny, bnds[1]), (gz, nz, bnds[2])):
        g0 = np.floor(gq).astype(np.int64)
        s0, s1 = _bound_sign(g0, n, b), _bound_sign(g0 + 1, n, b)
        s0 = np.ones(g0.shape) if s0 is None else s0.astype(np.float64)
        s1 = np.ones(g0.shape) if s1 is None else s1.astype(np.float64)
        axes.append((_bound_index(g0, n, b), _bound_index(g0 + 1, n, b), s0, s1, gq - np.floor(gq)))
    return axes, mask


def grid_push_linear(inp, grid, shape, bound="zero", extrapolate=False):
    """interpol.grid_push(interpolation=1) -> iso1.push3d (utils/interpol/iso1.py:136-265), fp64.
    inp (B,C,iX,iY,iZ), grid (B,iX,iY,iZ,3) -> (B,C,*shape)."""
    inp = np.asarray(inp, np.float64)
    B, Cc = inp.shape[:2]
    nx, ny, nz = shape
    axes, mask = _corner_terms(grid, shape, bound, extrapolate)
    flat = inp.reshape(B, Cc, -1)
    out = np.zeros((B, Cc, nx * ny * nz))
    for cx in (0, 1):
        for cy in (0, 1):
            for cz in (0, 1):
                (ix, sx, wx), (iy, sy, wy), (iz, sz, wz) = [
                    (a[1] if c else a[0], a[3] if c else a[2], a[4] if c else 1.0 - a[4]) for a, c in zip(axes, (cx, cy, cz))]
                idx = iz + iy * nz + ix * (ny * nz)
                val = flat * (sx * sy * sz * mask * wx * wy * wz)[:, None, :]
                for b in range(B):
                    for c in range(Cc):
                        np.add.at(out[b, c], idx[b], val[b, c])
    return out.reshape(B, Cc, nx, ny, nz)


def grid_grad_linear(inp, grid, bound="zero", extrapolate=False):
    """interpol.grid_grad(interpolation=1) -> iso1.grad3d (utils/interpol/iso1.py:268-387), fp64.
    inp (B,C,X,Y,Z), grid (B,oX,oY,oZ,3) -> (B,C,oX,oY,oZ,3)."""
    inp = np.asarray(inp, np.float64)
    B, Cc, nx, ny, nz = inp.shape
    axes, mask = _corner_terms(grid, (nx, ny, nz), bound, extrapolate)
    flat = inp.reshape(B, Cc, -1)
    out = np.zeros((B, Cc, axes[0][0].shape[-1], 3))
    for cx in (0, 1):
        for cy in (0, 1):
            for cz in (0, 1):
                (ix, sx, wx), (iy, sy, wy), (iz, sz, wz) = [
                    (a[1] if c else a[0], a[3] if c else a[2], a[4] if c else 1.0 - a[4]) for a, c in zip(axes, (cx, cy, cz))]
                idx = iz + iy * nz + ix * (ny * nz)
                val = np.take_along_axis(flat, np.broadcast_to(idx[:, None, :], (B, Cc, idx.shape[-1])), axis=-1)
                val = val * (sx * sy * sz)[:, None, :]
                dx, dy, dz = (1.0 if cx else -1.0), (1.0 if cy else -1.0), (1.0 if cz else -1.0)
                out[..., 0] += val * (dx * wy * wz)[:, None, :]
                out[..., 1] += val * (dy * wx * wz)[:, None, :]
                out[..., 2] += val * (dz * wx * wy)[:, None, :]
    out *= mask[:, None, :, None]
    return out.reshape(B, Cc, *grid.shape[1:4], 3)
