"""Host-side mirror of ``Generator/utils.py`` (the kernels that feed training-data synthesis).

Same function names, argument meaning and error behaviour as the reference; tensors stay
torch tensors on the HIP device, the arithmetic runs in libbrainfm_hip.so:

  fast_3D_interp_torch(X, II, JJ, KK, mode, default_value_linear)    Generator/utils.py:119-196
  myzoom_torch(X, factor, aff)                                       :200-257
  make_gaussian_kernel / gaussian_blur_3d                            :74-94
  add_gamma_transform / add_bias_field / resample_resolution / add_noise   :568-638
  make_affine_matrix / binarize / resolution_sampler                 :34-57,:65-72,:102-116
  augment_pathology                                                  :542-560

Random draws keep the reference's call order on ``np.random`` so a seeded host stream lines up;
``torch.randn`` is drawn on the device (RNG-stream parity across devices is not a goal).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L


def _f32(t, device=None):
    t = torch.as_tensor(t)
    if device is not None:
        t = t.to(device)
    return t.to(torch.float32).contiguous()


def _require_cuda(t, what):
    if t.device.type != "cuda":
        raise L.BfmError("%s runs on a HIP device only; there is no CPU fallback in the product path" % what)


# ----------------------------------------------------------------------------- sampling helpers (host)
class DeviceDraws:
    """Where the generator's torch draws come from.  The reference draws them with torch.rand / torch.randn on its
    device; here the volume-sized normal fields are drawn on the GPU and the small tables on the host (RNG-stream parity
    across devices is not a goal).  Tests replace ``generator_utils.draws`` with ReplayDraws to feed the reference's own
    recorded draws through the chain (tests/golden/make_golden_gen.py)."""

    _last_state = None
    _same_state_calls = 0

    def _key(self):
        """63-bit Philox key of the next field: a hash of torch's default CPU generator STATE plus the number of fields
        drawn since that state last changed.  Nothing is drawn from the host stream (round 4 took a randint per field,
        which shifted every later torch.rand / np-independent host draw relative to the reference's order: ADVICE r4);
        torch.manual_seed(s) restarts the fields with the host stream, and two runs from the same seed draw the same
        fields.  (One corner cannot be seen from the state: re-seeding to the state the generator is already in -- the same
        seed again with no host draw since -- continues the count; reseed() restarts it explicitly.)  Device generators
        (torch.cuda.manual_seed) play no part."""
        import hashlib
        st = torch.random.get_rng_state()
        h = int.from_bytes(hashlib.blake2b(st.numpy().tobytes(), digest_size=8).digest(), "little")
        if h == DeviceDraws._last_state:
            DeviceDraws._same_state_calls += 1
        else:
            DeviceDraws._last_state, DeviceDraws._same_state_calls = h, 0
        return (h + DeviceDraws._same_state_calls * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)

    @staticmethod
    def reseed():
        """Restart the count of fields drawn at the current host-generator state (see _key)."""
        DeviceDraws._last_state, DeviceDraws._same_state_calls = None, 0

    def randn(self, shape, device):
        """N(0,1) field from bfm_randn_philox (Philox4x32-10 + Box-Muller on 24-bit uniforms: |x| <= 5.9 sigma), keyed by
        _key(): repeatable under torch.manual_seed, no host draw consumed."""
        key = self._key()
        out = torch.empty(list(shape), dtype=torch.float, device=device)
        if out.numel():
            L.check(L.load().bfm_randn_philox(L.ptr(out), out.numel(), C.c_uint64(key), C.c_uint64(0), 1.0,
                                              L.stream_ptr()), "randn_philox")
        return out

    def rand(self, n):
        return torch.rand(n, dtype=torch.float)


class ReplayDraws:
    """Hands out recorded draws [(kind, array)] in call order; a draw of another kind or shape is an error (the mirror
    left the reference's call sequence)."""

    def __init__(self, seq):
        self.seq = list(seq)
        self.pos = 0

    def _next(self, kind, shape):
        if self.pos >= len(self.seq):
            raise AssertionError("the chain asked for draw #%d (%s %s) but the reference made only %d"
                                 % (self.pos, kind, tuple(shape), len(self.seq)))
        k, arr = self.seq[self.pos]
        if k != kind or tuple(arr.shape) != tuple(shape):
            raise AssertionError("draw #%d: the chain asks for %s %s, the reference drew %s %s"
                                 % (self.pos, kind, tuple(shape), k, tuple(arr.shape)))
        self.pos += 1
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))

    def randn(self, shape, device):
        return self._next("randn", tuple(shape)).to(device)

    def rand(self, n):
        return self._next("rand", (int(n),))


draws = DeviceDraws()


def resolution_sampler(low_res_only=False):
    """Generator/utils.py:34-57 (host RNG only)."""
    r = (np.random.rand() * 0.5) + 0.5 if low_res_only else np.random.rand()
    if r < 0.25:
        resolution = np.array([1.0, 1.0, 1.0]); thickness = np.array([1.0, 1.0, 1.0])
    elif r < 0.5:
        resolution = np.array([1.0, 1.0, 1.0]); thickness = np.array([1.0, 1.0, 1.0])
        idx = np.random.randint(3)
        resolution[idx] = 2.5 + 6 * np.random.rand()
        thickness[idx] = np.min([resolution[idx], 4.0 + 2.0 * np.random.rand()])
    elif r < 0.75:
        resolution = np.array([1.3, 1.3, 4.8]) + 0.4 * np.random.rand(3)
        thickness = resolution.copy()
    else:
        resolution = 2.0 + 3.0 * np.random.rand(3)
        thickness = resolution.copy()
    return resolution, thickness


def _rotation_about(axis, angle):
    """Right-handed rotation about one coordinate axis: the 2x2 block [[c, -s], [s, c]] on the other two axes taken in
    cyclic order (x: (y,z), y: (z,x), z: (x,y))."""
    i, j = (axis + 1) % 3, (axis + 2) % 3
    R = np.eye(3)
    c, sn = np.cos(angle), np.sin(angle)
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -sn, sn, c
    return R


def _shear_along(axis, sh):
    """Identity whose column `axis` carries the shear coefficients of the other two rows."""
    S = np.eye(3)
    for r in range(3):
        if r != axis:
            S[r, axis] = sh[r]
    return S


def make_affine_matrix(rot, sh, s):
    """Generator/utils.py:102-116: shears (x, y, z) times rotations (x, y, z), rows scaled by s -- the same factors in
    the same order, so the float64 result is the reference's bit for bit (gen_chain.npz)."""
    A = np.eye(3)
    for M in [_shear_along(k, sh) for k in range(3)] + [_rotation_about(k, rot[k]) for k in range(3)]:
        A = A @ M
    return A * np.asarray(s, dtype=np.float64).reshape(3, 1)


# ----------------------------------------------------------------------------- reductions / elementwise
_WS = {}


def workspace(dev, nbytes=1 << 20):
    """One scratch buffer per device for the reductions' block partials (kernels on one stream run in order, so they can
    share it; the generator is single-stream)."""
    key = str(dev)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    return ws


def reduce_dev(op, x, y=None):
    """min (0) / max (1) / sum (2) / sum(x*y) (3) of an fp32 tensor as a 1-element fp64 DEVICE tensor: no host sync."""
    lib = L.load()
    x = x.contiguous()
    ws = workspace(x.device)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    L.check(lib.bfm_reduce_f32(op, L.ptr(x), L.ptr(y.contiguous()) if y is not None else None, x.numel(), L.ptr(out),
                               L.ptr(ws), ws.numel(), L.stream_ptr()), "reduce")
    return out


def _reduce(op, x, y=None):
    return float(reduce_dev(op, x, y).item())


def tensor_min(x): return _reduce(0, x)
def tensor_max(x): return _reduce(1, x)
def tensor_sum(x): return _reduce(2, x)
def tensor_dot(x, y): return _reduce(3, x, y)


def ew_unary(op, x, a=0.0, b=0.0):
    lib = L.load()
    x = x.contiguous()
    out = torch.empty_like(x)
    L.check(lib.bfm_ew_unary(op, L.ptr(x), 1, L.ptr(out), 1, x.numel(), float(a), float(b), L.stream_ptr()), "ew_unary")
    return out


def ew_binary(op, x, y, a=0.0):
    lib = L.load()
    x = x.contiguous(); y = y.contiguous()
    out = torch.empty_like(x)
    ys = 0 if y.numel() == 1 else 1
    L.check(lib.bfm_ew_binary(op, L.ptr(x), 1, L.ptr(y), ys, L.ptr(out), 1, x.numel(), float(a), L.stream_ptr()),
            "ew_binary")
    return out


def binarize(p, thres):
    """Generator/utils.py:65-72: 1 where p >= thres*max(p) else 0 (dtype of p kept)."""
    lib = L.load()
    if p.dtype == torch.float64:
        pc = p.contiguous()
        from .shapeid import tensor_max_f64
        t = thres * tensor_max_f64(pc)
        masked = torch.empty_like(pc)
        mask = torch.empty_like(pc)
        L.check(lib.bfm_threshold_mask_f64(L.ptr(pc), pc.numel(), float(t), L.ptr(masked), L.ptr(mask),
                                           L.stream_ptr()), "threshold_mask")
        return mask
    pf = p.to(torch.float32).contiguous()
    t = float(np.float32(thres) * np.float32(tensor_max(pf)))
    return ew_unary(L.EW_GE, pf, t).to(p.dtype)


def binarize_dev(p, thres, max_dev=None):
    """binarize with every scalar on the device: (P, sum(P) as a 1-element fp64 device tensor).  max_dev: max(p) when the
    producer already has it (generate_shape_3d_dev)."""
    lib = L.load()
    f64 = p.dtype == torch.float64
    pc = p.contiguous() if f64 else p.to(torch.float32).contiguous()
    if max_dev is None:
        if f64:
            max_dev = torch.empty(1, dtype=torch.float64, device=pc.device)
            ws0 = workspace(pc.device)
            L.check(lib.bfm_reduce_f64(1, L.ptr(pc), None, pc.numel(), L.ptr(max_dev), L.ptr(ws0), ws0.numel(),
                                       L.stream_ptr()), "reduce")
        else:
            max_dev = reduce_dev(1, pc)
    P = torch.empty_like(pc)
    psum = torch.empty(1, dtype=torch.float64, device=pc.device)
    ws = workspace(pc.device)
    L.check(lib.bfm_shape_binarize(L.ptr(pc), int(f64), pc.numel(), L.ptr(max_dev), float(thres), L.ptr(P), L.ptr(psum),
                                   L.ptr(ws), ws.numel(), L.stream_ptr()), "shape_binarize")
    if not f64 and p.dtype != torch.float32:
        P = P.to(p.dtype)
    return P, psum


# ----------------------------------------------------------------------------- K10 / K11
def fast_3D_interp_torch(X, II, JJ, KK, mode="linear", default_value_linear=0.0):
    """Generator/utils.py:119-196.  X: (nx,ny,nz[,C]); II/JJ/KK: coordinate tensors of any common shape."""
    if II is None:
        return X
    if mode not in ("linear", "nearest"):
        raise Exception("mode must be linear or nearest")
    _require_cuda(X, "fast_3D_interp_torch")
    lib = L.load()
    dev = X.device
    squeeze = X.dim() == 3
    Xc = (X[..., None] if squeeze else X)
    nx, ny, nz, Cc = Xc.shape
    II, JJ, KK = (_f32(t, dev) for t in (II, JJ, KK))
    n = II.numel()
    if mode == "nearest":
        if II.dim() != 3:
            # the reference indexes Y.shape[3]: it only works with 3-D coordinate grids in this mode
            raise IndexError("tuple index out of range")
        if Xc.dtype in (torch.int32, torch.float32):
            src = Xc.contiguous()
        elif Xc.dtype in (torch.int64, torch.int16, torch.uint8, torch.bool):
            src = Xc.to(torch.int32).contiguous()
        else:
            src = Xc.to(torch.float32).contiguous()
        out = torch.empty(tuple(II.shape) + (Cc,), dtype=src.dtype, device=dev)
        L.check(lib.bfm_interp3d_nearest(L.ptr(src), nx, ny, nz, Cc, L.ptr(II), L.ptr(JJ), L.ptr(KK), n, L.ptr(out),
                                         L.stream_ptr()), "interp3d_nearest")
        out = out.to(Xc.dtype) if out.dtype != Xc.dtype else out
        return out[..., 0] if out.shape[3] == 1 else out
    dv = float(default_value_linear)
    src = Xc.to(torch.float32).contiguous()
    out = torch.empty(tuple(II.shape) + (Cc,), dtype=torch.float32, device=dev)
    L.check(lib.bfm_interp3d_linear(L.ptr(src), nx, ny, nz, Cc, L.ptr(II), L.ptr(JJ), L.ptr(KK), n, dv, L.ptr(out),
                                    L.stream_ptr()), "interp3d_linear")
    return out[..., 0] if Cc == 1 else out


# ----------------------------------------------------------------------------- K12
def torch_cpu_arange_f32(start, end, step, vec=8):
    """torch.arange(start, end, step, dtype=float32) exactly as ATen's CPU kernel evaluates it (the
    reference builds its zoom tables this way, Generator/utils.py:208-210): pairs of 8-lane vectors are
    float(float(start + step*idx) + lane*step); the remainder is float(start + step*idx)."""
    cnt = int(math.ceil((end - start) / step))
    out = np.empty(max(cnt, 0), dtype=np.float32)
    i = 0
    while cnt - i >= 2 * vec:
        for _ in range(2):
            base = np.float64(np.float32(start + step * i))
            out[i:i + vec] = (base + np.arange(vec, dtype=np.float64) * step).astype(np.float32)
            i += vec
    out[i:] = (start + step * np.arange(i, cnt, dtype=np.float64)).astype(np.float32)
    return out


def zoom_tables(n, factor):
    """Per-axis (floor idx, ceil idx, w_floor, w_ceil) of myzoom_torch, Generator/utils.py:205-235."""
    delta = (1.0 - factor) / (2.0 * factor)
    new = int(np.round(n * factor))
    v = torch_cpu_arange_f32(delta, delta + new / factor, 1.0 / factor)[:new]
    v = np.where(v < 0, np.float32(0), v)
    v = np.where(v > n - 1, np.float32(n - 1), v).astype(np.float32)
    f = np.floor(v).astype(np.int32)
    c = np.minimum(f + 1, n - 1).astype(np.int32)
    wc = (v - f.astype(np.float32)).astype(np.float32)
    wf = (np.float32(1) - wc).astype(np.float32)
    return f, c, wf, wc


_ZOOM_CACHE = {}


def _zoom_tables_dev(n, factor, dev):
    """Device copies of zoom_tables(n, factor), cached: the same few (size, factor) pairs recur for every sample."""
    key = (int(n), float(factor), str(dev))
    if key not in _ZOOM_CACHE:
        if len(_ZOOM_CACHE) > 256:
            _ZOOM_CACHE.clear()
        _ZOOM_CACHE[key] = [torch.from_numpy(v).to(dev) for v in zoom_tables(n, float(factor))]
    return _ZOOM_CACHE[key]


def _zoomed_affine(aff, factor):
    """vox2ras of a grid resampled by `factor` (Generator/utils.py:252-255): voxel axes shrink by the factor and the
    first voxel's centre moves by half the change in voxel size along each axis."""
    old_axes = aff[:-1, :-1]
    centre_shift = old_axes @ (0.5 - 0.5 / (factor * np.ones(3)))
    out = aff.copy()
    out[:-1] = out[:-1] / factor
    out[:-1, -1] = out[:-1, -1] - centre_shift
    return out


def myzoom_torch(X, factor, aff=None):
    """Generator/utils.py:200-257: separable linear zoom (fused into one kernel)."""
    _require_cuda(X, "myzoom_torch")
    lib = L.load()
    dev = X.device
    squeeze = X.dim() == 3
    Xc = (X[..., None] if squeeze else X).to(torch.float32).contiguous()
    factor = np.asarray(factor, dtype=np.float64) * np.ones(3)
    nx, ny, nz, Cc = Xc.shape
    tabs, keep, newsize = [], [], []
    axes = (L.ZoomAxis * 3)()
    for a, n in enumerate((nx, ny, nz)):
        t = _zoom_tables_dev(n, float(factor[a]), dev)
        newsize.append(len(t[0]))
        keep.append(t)
        axes[a] = L.ZoomAxis(*[v.data_ptr() for v in t])
    out = torch.empty((newsize[0], newsize[1], newsize[2], Cc), dtype=torch.float32, device=dev)
    L.check(lib.bfm_zoom_linear(L.ptr(Xc), nx, ny, nz, Cc, axes, newsize[0], newsize[1], newsize[2], L.ptr(out),
                                L.stream_ptr()), "zoom_linear")
    Y = out[..., 0] if Cc == 1 else out
    if aff is not None:
        return Y, _zoomed_affine(aff, factor)
    return Y


# ----------------------------------------------------------------------------- K13
_GAUSS_CACHE = {}


def make_gaussian_kernel(sigma, device):
    """Generator/utils.py:74-82 (7-tap example in SURVEY appendix C); device copies cached per sigma."""
    key = (float(sigma), str(device))
    if key in _GAUSS_CACHE:
        return _GAUSS_CACHE[key]
    if len(_GAUSS_CACHE) > 256:
        _GAUSS_CACHE.clear()
    _GAUSS_CACHE[key] = _make_gaussian_kernel(sigma, device)
    return _GAUSS_CACHE[key]


def _make_gaussian_kernel(sigma, device):
    sl = int(np.ceil(3 * sigma))
    ts = np.linspace(-sl, sl, 2 * sl + 1).astype(np.float32)
    g = np.exp((-(ts / np.float32(sigma)) ** 2 / 2)).astype(np.float32)
    return torch.from_numpy((g / g.sum(dtype=np.float32)).astype(np.float32)).to(device)


def gaussian_blur_3d(input, stds, device):
    """Generator/utils.py:84-94: three zero-padded 1-D correlations."""
    _require_cuda(input, "gaussian_blur_3d")
    lib = L.load()
    cur = input.to(torch.float32).contiguous()
    nx, ny, nz = cur.shape
    for ax in range(3):
        if stds[ax] > 0:
            k = make_gaussian_kernel(stds[ax], cur.device)
            out = torch.empty_like(cur)
            L.check(lib.bfm_conv1d_axis(L.ptr(cur), nx, ny, nz, ax, L.ptr(k), k.numel(), L.ptr(out), L.stream_ptr()),
                    "conv1d_axis")
            cur = out
    return cur


# ----------------------------------------------------------------------------- K14 augmentations
def add_gamma_transform(I, aux_dict, cfg, device, **kwargs):
    """Generator/utils.py:568-572: 300 * (I/300) ** exp(gamma_std * N(0,1))."""
    gamma = float(np.exp(cfg.gamma_std * np.random.randn(1)[0]))
    return ew_unary(L.EW_GAMMA, I.to(torch.float32), 300.0, gamma), aux_dict


def add_bias_field(I, aux_dict, cfg, input_mode, setups, size, device, **kwargs):
    """Generator/utils.py:574-589."""
    if input_mode == "CT":
        aux_dict.update({"high_res": I})
        return I, aux_dict
    bf_scale = cfg.bf_scale_min + np.random.rand(1) * (cfg.bf_scale_max - cfg.bf_scale_min)
    size_BF_small = np.round(bf_scale * np.array(size)).astype(int).tolist()
    if setups["photo_mode"]:
        size_BF_small[1] = np.round(size[1] / setups["spac"]).astype(int)
    amp = float(np.float32(cfg.bf_std_min + (cfg.bf_std_max - cfg.bf_std_min) * np.random.rand(1))[0])
    BFsmall = ew_unary(L.EW_AFFINE, draws.randn(size_BF_small, I.device), amp, 0.0)
    BFlog = myzoom_torch(BFsmall, np.array(size) / size_BF_small)
    I_bf = ew_binary(L.EW_MUL_EXP, I.to(torch.float32), BFlog)
    aux_dict.update({"BFlog": BFlog, "high_res": I_bf})
    return I_bf, aux_dict


def resample_resolution(I, aux_dict, setups, res, size, device, **kwargs):
    """Generator/utils.py:591-609: blur to the slice thickness, then trilinear sample on the low-res grid."""
    stds = (0.85 + 0.3 * np.random.rand()) * np.log(5) / np.pi * setups["thickness"] / res
    stds[setups["thickness"] <= res] = 0.0
    I_blur = gaussian_blur_3d(I, stds, device)
    new_size = (np.array(size) * res / setups["resolution"]).astype(int)
    factors = np.array(new_size) / np.array(size)
    delta = (1.0 - factors) / (2.0 * factors)
    v = [np.arange(delta[a], delta[a] + new_size[a] / factors[a], 1 / factors[a])[:new_size[a]] for a in range(3)]
    # the reference samples at np.meshgrid(v0, v1, v2) converted to float32 (:600-606): the grid is separable, so the
    # three float32 axis tables are all the kernel needs (same coordinates, same arithmetic as fast_3D_interp_torch)
    n0, n1, n2 = (len(t) for t in v)
    tabs = torch.from_numpy(np.concatenate(v).astype(np.float32)).to(I.device)
    src = I_blur.to(torch.float32).contiguous()
    I_small = torch.empty((n0, n1, n2), dtype=torch.float32, device=I.device)
    if I_small.numel():
        L.check(L.load().bfm_interp3d_linear_axes(L.ptr(src), src.shape[0], src.shape[1], src.shape[2], L.ptr(tabs),
                                                  C.c_void_p(tabs.data_ptr() + 4 * n0),
                                                  C.c_void_p(tabs.data_ptr() + 4 * (n0 + n1)), n0, n1, n2, 0.0,
                                                  L.ptr(I_small), L.stream_ptr()), "interp3d_linear_axes")
    aux_dict.update({"factors": factors})
    return I_small, aux_dict


def add_noise(I, aux_dict, cfg, device, **kwargs):
    """Generator/utils.py:633-638: I + std * N(0,1), clamped at 0."""
    noise_std = float(np.float32(cfg.noise_std_min + (cfg.noise_std_max - cfg.noise_std_min) * np.random.rand(1))[0])
    rn = draws.randn(I.shape, I.device)
    return ew_binary(L.EW_AXPY_CLAMP0, I.to(torch.float32), rn, noise_std), aux_dict


augmentation_funcs = {"gamma": add_gamma_transform, "bias_field": add_bias_field, "resample": resample_resolution,
                      "noise": add_noise}


# ----------------------------------------------------------------------------- pathology shape augmentation
def augment_pathology(Pprob, pde_func, t, shape_gen_args, device):
    """Generator/utils.py:542-560: advect the probability map along a random divergence-free field."""
    from .shapeid import generate_velocity_3d, odeint_adjoint
    Pprob = torch.squeeze(Pprob)
    nt = np.random.randint(1, shape_gen_args.max_nt + 1)
    if nt <= 1:
        return Pprob
    pde_func.V_dict = generate_velocity_3d(Pprob.shape, shape_gen_args.perlin_res, shape_gen_args.V_multiplier, device)
    return odeint_adjoint(pde_func, Pprob[None], t[:nt], shape_gen_args.dt, method=shape_gen_args.integ_method)[-1, 0]
