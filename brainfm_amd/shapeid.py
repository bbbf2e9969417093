"""Host-side mirror of ``ShapeID`` (Perlin shapes, curl velocity, advection PDE, dopri5).

  generate_perlin_noise_3d / generate_shape_3d / generate_velocity_3d     ShapeID/perlin3d.py:15-156
  stream_3D                                                               ShapeID/misc.py:66-80
  AdvDiffPDE(...).forward(t, batch_C)                                     ShapeID/DiffEqs/pde.py:563-640
  odeint / odeint_adjoint (method 'dopri5')                               DiffEqs/odeint.py:20-75, adjoint.py:105-132,
                                                                          dopri5.py:58-172, rk_common.py, interp.py, misc.py

The lattice gradients are drawn on the host with ``np.random`` exactly like the reference (two
``rand`` calls per field) and uploaded; everything per voxel runs in libbrainfm_hip.so.  The
Dormand-Prince controller (accept / reject, clamped step) stays on the host and reads one reduced
scalar per step; only the solvers and PDE variants the shipped configs use are provided.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _dev(device):
    d = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
    if d.type != "cuda":
        raise L.BfmError("ShapeID kernels run on a HIP device only; there is no CPU fallback in the product path")
    return d


def _ws(dev):
    return torch.empty(L.load().bfm_reduce_workspace(), dtype=torch.uint8, device=dev)


def tensor_max_f64(x):
    lib = L.load()
    ws = _ws(x.device)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    L.check(lib.bfm_reduce_f64(1, L.ptr(x), None, x.numel(), L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()), "reduce")
    return float(out.item())


def interpolant(t):
    return t * t * t * (t * (t * 6 - 15) + 10)


# ----------------------------------------------------------------------------- Perlin
def perlin_gradients(res, tileable=(False, False, False)):
    """Lattice gradients, ShapeID/perlin3d.py:44-55 (host RNG, fp64)."""
    theta = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)
    phi = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)
    return gradients_from_angles(theta, phi, tileable)


def gradients_from_angles(theta, phi, tileable=(False, False, False)):
    g = np.stack((np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)), axis=3)
    if tileable[0]:
        g[-1, :, :] = g[0, :, :]
    if tileable[1]:
        g[:, -1, :] = g[:, 0, :]
    if tileable[2]:
        g[:, :, -1] = g[:, :, 0]
    return g


def perlin_from_gradients(shape, res, gradients, device):
    lib = L.load()
    dev = _dev(device)
    g = torch.from_numpy(np.ascontiguousarray(gradients, dtype=np.float64)).to(dev)
    out = torch.empty(tuple(shape), dtype=torch.float64, device=dev)
    L.check(lib.bfm_perlin3d(L.ptr(g), shape[0], shape[1], shape[2], res[0], res[1], res[2], L.ptr(out), L.stream_ptr()),
            "perlin3d")
    return out


def kth_smallest_f64(x, k):
    """k-th order statistic (0-based) of a device fp64 tensor by 4 passes of 16-bit radix histograms."""
    lib = L.load()
    n = x.numel()
    prefix = 0
    hist = torch.empty(65536, dtype=torch.int32, device=x.device)
    for shift in (48, 32, 16, 0):
        hist.zero_()
        L.check(lib.bfm_radix_hist_f64(L.ptr(x), n, C.c_uint64(prefix), shift, L.ptr(hist), L.stream_ptr()), "radix_hist")
        h = hist.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        cum = np.cumsum(h)
        d = int(np.searchsorted(cum, k, side="right"))
        k -= int(cum[d - 1]) if d > 0 else 0
        prefix = (prefix << 16) | d
    key = np.uint64(prefix)
    u = (~key) if not (int(key) >> 63) else (key & np.uint64(0x7FFFFFFFFFFFFFFF))
    return float(np.array([u], dtype=np.uint64).view(np.float64)[0])


def percentile_dev(x, q):
    """np.percentile(x, q) (method='linear') of a device fp64 tensor, left on the device: a 3-element fp64 tensor
    {percentile, x_(k), x_(k+1)}.  Radix select + NumPy's lerp inside libbrainfm_hip.so (bfm_percentile_f64): no
    histogram ever travels to the host (round 3: 8 histogram read-backs of 256 KB per percentile)."""
    lib = L.load()
    x = x.contiguous()
    n = x.numel()
    pos = q / 100.0 * (n - 1)
    lo = int(math.floor(pos))
    hi = min(lo + 1, n - 1)
    t = pos - lo
    out = torch.empty(3, dtype=torch.float64, device=x.device)
    ws = torch.empty(lib.bfm_percentile_workspace(), dtype=torch.uint8, device=x.device)
    L.check(lib.bfm_percentile_f64(L.ptr(x), n, lo, int(hi != lo), float(t), L.ptr(out), L.ptr(ws), ws.numel(),
                                   L.stream_ptr()), "percentile")
    return out


def percentile_linear(x, q):
    """np.percentile(x, q) as a host float (one read-back)."""
    return float(percentile_dev(x, q)[0].item())


def generate_perlin_noise_3d(shape, res, tileable=(False, False, False), interpolant=interpolant, percentile=None,
                             device="cuda"):
    """ShapeID/perlin3d.py:15-90.  Returns a device fp64 tensor (or (noise*mask, mask) with a percentile)."""
    if shape[0] % res[0] or shape[1] % res[1] or shape[2] % res[2]:
        raise ValueError("shape must be a multiple of res")
    noise = perlin_from_gradients(shape, res, perlin_gradients(res, tileable), device)
    if percentile is None:
        return noise
    return threshold_at_percentile(noise, percentile)


def threshold_at_percentile(noise, percentile, want_mask=True, want_max=False):
    """(noise * mask, mask) with mask = noise >= np.percentile(noise, percentile) (perlin3d.py:84-90); the threshold never
    leaves the device.  want_max: also the 1-element device tensor max(noise * mask) (binarize's operand)."""
    lib = L.load()
    thr = percentile_dev(noise, percentile)
    masked = torch.empty_like(noise)
    mask = torch.empty_like(noise) if want_mask else None
    mx = torch.empty(1, dtype=torch.float64, device=noise.device)
    ws = torch.empty(lib.bfm_shape_workspace(), dtype=torch.uint8, device=noise.device)
    L.check(lib.bfm_shape_threshold_f64(L.ptr(noise), noise.numel(), L.ptr(thr), L.ptr(masked), L.ptr(mask), L.ptr(mx),
                                        L.ptr(ws), ws.numel(), L.stream_ptr()), "shape_threshold")
    if want_max:
        return masked, mask, mx
    return masked, mask


def generate_shape_3d(shape, perlin_res, percentile, device):
    """perlin3d.py:144-146: returns (mask, noise*mask) like the reference (p, pprob)."""
    pprob, p = generate_perlin_noise_3d(shape, perlin_res, tileable=(True, False, False), percentile=percentile,
                                        device=device)
    return p, pprob


def generate_shape_3d_dev(shape, perlin_res, percentile, device):
    """generate_shape_3d for the generator: (noise * mask, its maximum as a device scalar); the 0/1 mask, which
    read_and_deform_pathology drops (Generator/utils.py:438), is not written.  Same np.random draws."""
    if shape[0] % perlin_res[0] or shape[1] % perlin_res[1] or shape[2] % perlin_res[2]:
        raise ValueError("shape must be a multiple of res")
    noise = perlin_from_gradients(shape, perlin_res, perlin_gradients(perlin_res, (True, False, False)), device)
    masked, _, mx = threshold_at_percentile(noise, percentile, want_mask=False, want_max=True)
    return masked, mx


def stream_3D(Phi_a, Phi_b, Phi_c, batched=False, delta_lst=[1., 1., 1.], multiplier=1.0):
    """ShapeID/misc.py:66-80 on fp64 potentials; returns fp32 (Vx, Vy, Vz) (times `multiplier`)."""
    if batched or list(delta_lst) != [1., 1., 1.]:
        raise NotImplementedError("only the unbatched unit-spacing form used by generate_velocity_3d is provided")
    lib = L.load()
    sx, sy, sz = Phi_a.shape
    dev = Phi_a.device
    V = [torch.empty((sx, sy, sz), dtype=torch.float32, device=dev) for _ in range(3)]
    a, b, c = (t.to(torch.float64).contiguous() for t in (Phi_a, Phi_b, Phi_c))
    L.check(lib.bfm_curl3d(L.ptr(a), L.ptr(b), L.ptr(c), sx, sy, sz, float(multiplier), L.ptr(V[0]), L.ptr(V[1]),
                           L.ptr(V[2]), L.stream_ptr()), "curl3d")
    return V[0], V[1], V[2]


def generate_velocity_3d(shape, perlin_res, V_multiplier, device):
    """perlin3d.py:149-156: curl of three Perlin potentials times V_multiplier."""
    pots = [generate_perlin_noise_3d(shape, perlin_res, tileable=(True, False, False), device=device) for _ in range(3)]
    Vx, Vy, Vz = stream_3D(pots[0], pots[1], pots[2], multiplier=V_multiplier)
    return {"Vx": Vx, "Vy": Vy, "Vz": Vz}


# ----------------------------------------------------------------------------- PDE right-hand side
class AdvDiffPDE(nn.Module):
    """ShapeID/DiffEqs/pde.py:563-640 for the configuration the generator uses
    (Generator/datasets.py:131-138): perf_pattern='adv', V_type='vector_div_free', 3-D, unit spacing."""

    def __init__(self, data_spacing, perf_pattern, D_type="scalar", V_type="vector", BC=None, dt=0.1, V_dict={},
                 D_dict={}, stochastic=False, device="cpu"):
        super().__init__()
        if len(data_spacing) != 3 or list(data_spacing) != [1., 1., 1.]:
            raise NotImplementedError("only 3-D unit-spacing volumes are on the synthesis path")
        if "diff" in perf_pattern or V_type != "vector_div_free" or stochastic:
            raise NotImplementedError("only perf_pattern='adv' with V_type='vector_div_free' is on the synthesis path")
        if BC not in (None, "neumann", "cauchy"):
            raise NotImplementedError("Unsupported B.C.!")
        self.BC, self.dt, self.dimension = BC, dt, 3
        self.perf_pattern, self.V_type, self.D_type = perf_pattern, V_type, D_type
        self.V_dict, self.D_dict = V_dict, D_dict
        self.nfe = 0

    def forward(self, t, batch_C):
        """batch_C: (1, s, r, c) fp32 or fp64 -> fp32 (1, s, r, c)."""
        lib = L.load()
        if batch_C.shape[0] != 1:
            return torch.cat([self.forward(t, batch_C[b:b + 1]) for b in range(batch_C.shape[0])], 0)
        Cc = batch_C.contiguous()
        _, sx, sy, sz = Cc.shape
        out = torch.empty((1, sx, sy, sz), dtype=torch.float32, device=Cc.device)
        V = self.V_dict
        L.check(lib.bfm_advect_upwind_rhs(L.ptr(Cc), 1 if Cc.dtype == torch.float64 else 0, L.ptr(V["Vx"]),
                                          L.ptr(V["Vy"]), L.ptr(V["Vz"]), sx, sy, sz,
                                          1 if self.BC in ("neumann", "cauchy") else 0, L.ptr(out), L.stream_ptr()),
                "advect_upwind_rhs")
        self.nfe += 1
        return out


# ----------------------------------------------------------------------------- Dormand-Prince (as shipped)
_BETA = [[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
         [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
         [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
_C_ERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
          -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1. / 60.]
_C_MID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
          187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _kset(ks, dt, coef, sd):
    """(dt*c) in the state dtype, then the fp32 value torch's 0-dim promotion multiplies the fp32 stage with
    (DiffEqs/misc.py:22-25)."""
    s = L.KSet()
    s.nk = len(ks)
    for j, (k, c) in enumerate(zip(ks, coef)):
        s.k[j] = k.data_ptr()
        s.coef[j] = float(np.float32(sd(dt) * sd(c)))
    return s


class Dopri5Solver:
    """Dopri5Solver with the reference's forced-accept step clamps (dopri5.py:58-172)."""

    def __init__(self, func, y0, rtol, atol, dt, safety=0.9, ifactor=10.0, dfactor=0.2):
        self.func, self.y0, self.rtol, self.atol, self.dt_cfg = func, y0, rtol, atol, dt
        self.safety, self.ifactor, self.dfactor = safety, ifactor, dfactor
        self.lib = L.load()
        self.f64 = y0.dtype == torch.float64
        self.sd = np.float64 if self.f64 else np.float32
        self.n = y0.numel()
        self.ws = _ws(y0.device)
        self.scalar = torch.empty(1, dtype=torch.float64, device=y0.device)
        self.nsteps = 0

    def _combine(self, y, ks, dt, coef):
        out = torch.empty_like(y)
        s = _kset(ks, dt, coef, self.sd)
        L.check(self.lib.bfm_rk_combine(L.ptr(y), int(self.f64), C.byref(s), L.ptr(out), self.n, L.stream_ptr()),
                "rk_combine")
        return out

    def _rms_scaled(self, a, b, y0):
        L.check(self.lib.bfm_scaled_sumsq(L.ptr(a), L.ptr(b) if b is not None else None,
                                          int(a.dtype == torch.float64), L.ptr(y0), int(self.f64), self.atol, self.rtol,
                                          self.n, L.ptr(self.scalar), L.ptr(self.ws), self.ws.numel(), L.stream_ptr()),
                "scaled_sumsq")
        return math.sqrt(float(self.scalar.item())) / (self.n ** 0.5)

    def _initial_step(self, t0, y0, f0):
        """_select_initial_step, DiffEqs/misc.py:84-143 (order 4)."""
        d0 = self._rms_scaled(y0, None, y0)
        d1 = self._rms_scaled(f0, None, y0)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        s = L.KSet(); s.nk = 1; s.k[0] = f0.data_ptr(); s.coef[0] = float(np.float32(self.sd(h0)))
        y1 = torch.empty_like(y0)
        L.check(self.lib.bfm_rk_combine(L.ptr(y0), int(self.f64), C.byref(s), L.ptr(y1), self.n, L.stream_ptr()),
                "rk_combine")
        f1 = self.func(t0 + h0, y1)
        d2 = self._rms_scaled(f1, f0, y0) / h0
        if d1 <= 1e-15 and d2 <= 1e-15:
            h1 = max(1e-6, h0 * 1e-3)
        else:
            h1 = (0.01 / max(d1, d2)) ** (1. / 5.)
        return float(min(100 * h0, h1))

    def integrate(self, t):
        t = [float(v) for v in t]
        if isinstance(self.func, AdvDiffPDE) and self.y0.dim() == 4 and self.y0.shape[0] == 1 and len(t) >= 2 \
                and os.environ.get("BFM_ODE_DEVICE", "1") != "0":
            return self._integrate_device(t)
        return self._integrate_host(t)

    def _integrate_device(self, t):
        """The same integration with the controller on the device (bfm_dopri5_advect_*, csrc/synth_ode.hip): the host
        selects the first step (three read-backs, once), then enqueues steps in chunks and reads one state block per
        chunk; accept / reject, the clamped next step and the dense outputs never leave the GPU.  Same expressions as
        the host loop below: same bits up to the last ulp of pow() in the step-size rule."""
        lib = self.lib
        func = self.func
        y0 = self.y0
        dev = y0.device
        _, sx, sy, sz = y0.shape
        n = self.n
        f0 = func(t[0], y0)
        dt0 = self._initial_step(t[0], y0, f0)
        nt = len(t)
        sol = torch.empty((nt,) + tuple(y0.shape), dtype=y0.dtype, device=dev)
        sol[0].copy_(y0)
        ybuf = [y0.clone(), torch.empty_like(y0)]
        fbuf = [f0, torch.empty_like(f0)]
        ks = [torch.empty_like(f0) for _ in range(5)]
        t_out = torch.tensor(t, dtype=torch.float64, device=dev)
        nstate = lib.bfm_dopri5_advect_state_bytes()
        state = torch.zeros(nstate, dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.bfm_dopri5_advect_workspace(sx, sy, sz), dtype=torch.uint8, device=dev)
        d = L.Dopri5Advect()
        d.y[0], d.y[1] = ybuf[0].data_ptr(), ybuf[1].data_ptr()
        d.f[0], d.f[1] = fbuf[0].data_ptr(), fbuf[1].data_ptr()
        for j in range(5):
            d.k[j] = ks[j].data_ptr()
        V = func.V_dict
        d.Vx, d.Vy, d.Vz = V["Vx"].data_ptr(), V["Vy"].data_ptr(), V["Vz"].data_ptr()
        d.sx, d.sy, d.sz = sx, sy, sz
        d.neumann_bc = 1 if func.BC in ("neumann", "cauchy") else 0
        d.is_f64 = int(self.f64)
        d.atol, d.rtol = self.atol, self.rtol
        d.tol_min_dt = 0.2 * self.dt_cfg if 0.1 * self.dt_cfg >= 0.01 else 0.01
        d.dt_max = 0.1
        d.safety, d.ifactor, d.dfactor = self.safety, self.ifactor, self.dfactor
        d.t_out, d.nt, d.sol = t_out.data_ptr(), nt, sol.data_ptr()
        d.state, d.workspace = state.data_ptr(), ws.data_ptr()
        L.check(lib.bfm_dopri5_advect_init(C.byref(d), t[0], dt0, L.stream_ptr()), "dopri5_advect_init")
        # steps needed if every one were as long as allowed; the first chunk covers that, later ones what is left
        chunk = max(4, min(16, int(math.ceil((t[-1] - t[0]) / d.dt_max)) + 2))
        while True:
            L.check(lib.bfm_dopri5_advect_steps(C.byref(d), chunk, L.stream_ptr()), "dopri5_advect_steps")
            ints = state[40:72].cpu().numpy().view(np.int32)
            cur, done, next_out, nsteps, naccept, accepted, err = (int(v) for v in ints[:7])
            if err:
                raise AssertionError("underflow in dt")
            if done:
                break
            chunk = 8
        self.nsteps = nsteps
        func.nfe += 6 * nsteps
        return sol

    def _integrate_host(self, t):
        y = self.y0
        f = self.func(t[0], y)
        dt = self._initial_step(t[0], y, f)
        t0s = t1s = t[0]
        interp = None
        sol = [y]
        tol_min_dt = 0.2 * self.dt_cfg if 0.1 * self.dt_cfg >= 0.01 else 0.01
        for ti in t[1:]:
            while ti > t1s:
                assert t1s + dt > t1s, "underflow in dt {}".format(dt)
                ks = [f]
                yi = y
                for beta in _BETA:
                    yi = self._combine(y, ks, dt, beta)
                    ks.append(self.func(t1s, yi))
                y1, f1 = yi, ks[-1]
                es = _kset(ks, dt, _C_ERR, self.sd)
                L.check(self.lib.bfm_rk_error_sumsq(C.byref(es), L.ptr(y), L.ptr(y1), int(self.f64), self.atol,
                                                    self.rtol, self.n, L.ptr(self.scalar), L.ptr(self.ws),
                                                    self.ws.numel(), L.stream_ptr()), "rk_error_sumsq")
                msr = float(self.scalar.item()) / self.n
                accept = msr <= 1
                if msr == 0:                                   # _optimal_step_size, misc.py:160-170
                    dt_next = dt * self.ifactor
                else:
                    dfactor = 1.0 if msr < 1 else self.dfactor
                    factor = max(1 / self.ifactor, min(math.sqrt(msr) ** (1 / 5) / self.safety, 1 / dfactor))
                    dt_next = dt / factor
                if not (dt_next < tol_min_dt or dt_next > 0.1):
                    if accept:
                        interp = (y, y1, ks, dt)
                        y, f, t0s, t1s = y1, f1, t1s, t1s + dt
                else:                                          # forced accept with a clamped next step (dopri5.py:159-168)
                    dt_next = tol_min_dt if dt_next < tol_min_dt else dt_next
                    dt_next = 0.1 if dt_next > 0.1 else dt_next
                    interp = (y, y1, ks, dt)
                    y, f, t0s, t1s = y1, f1, t1s, t1s + dt
                if os.environ.get("BFM_ODE_TRACE"):
                    print("host", self.nsteps, msr, dt, dt_next, t1s)
                dt = dt_next
                self.nsteps += 1
            sol.append(self._dense(interp, t0s, t1s, ti))
        return torch.stack(sol)

    def _dense(self, interp, t0, t1, t):
        y0, y1, ks, dt = interp
        sd = self.sd
        x = float(sd((sd(t) - sd(t0)) / (sd(t1) - sd(t0))))
        out = torch.empty_like(y0)
        ms = _kset(ks, dt, _C_MID, sd)
        L.check(self.lib.bfm_dopri5_dense_eval(L.ptr(y0), L.ptr(y1), int(self.f64), C.byref(ms), float(sd(dt)), x,
                                               L.ptr(out), self.n, L.stream_ptr()), "dense_eval")
        return out


def odeint(func, y0, t, dt, step_size=None, rtol=1e-7, atol=1e-9, method=None, options=None):
    """DiffEqs/odeint.py:20-75 for tensor states and the 'dopri5' method."""
    if method not in (None, "dopri5"):
        raise NotImplementedError("only 'dopri5' is used by the shipped configs (cfgs/generator/default.yaml:117)")
    if not torch.is_tensor(y0):
        raise NotImplementedError("tuple states are not used on the synthesis path")
    if not torch.is_floating_point(y0):
        raise TypeError("`y0` must be a floating point Tensor but is a {}".format(y0.type()))
    tt = torch.as_tensor(t).detach().cpu().to(torch.float64).tolist()
    assert all(b > a for a, b in zip(tt[:-1], tt[1:])), "t must be strictly increasing or decrasing"
    y0 = y0.contiguous()
    return Dopri5Solver(func, y0, rtol, atol, dt).integrate(tt)


def odeint_adjoint(func, y0, t, dt, rtol=1e-6, atol=1e-12, method=None, options=None):
    """DiffEqs/adjoint.py:105-132: the generator only ever runs the forward pass (under no_grad)."""
    if not isinstance(func, nn.Module):
        raise ValueError("func is required to be an instance of nn.Module.")
    with torch.no_grad():
        return odeint(func, y0, t, dt, rtol=rtol, atol=atol, method=method, options=options)
