"""Host-side mirror of the part of ``utils/interpol`` (vendored torch-interpol) that is on the hot
path: ``grid_pull`` with linear interpolation -> ``iso1.pull3d`` (utils/interpol/api.py:137-200,
autograd.py:125-152, pushpull.py:35-66, iso1.py:28-133, bounds.py:24-89, jit_utils.py:241-255).

grid_pull / grid_push are differentiable to first order through grid_push / grid_grad; grid_grad itself
is forward only.
"""
import ctypes as C

import torch

from . import _lib as L

_BOUND = dict(zero=0, zeros=0, replicate=1, nearest=1, repeat=1, border=1, dct1=2, mirror=2, dct2=3, reflect=3,
              reflection=3, neumann=3, dst1=4, antimirror=4, dst2=5, antireflect=5, dirichlet=5, dft=6, wrap=6,
              circular=6)
_INTER = {"nearest": 0, "linear": 1, 0: 0, 1: 1}


def _bound_list(bound):
    if isinstance(bound, (str, int)):
        bound = [bound] * 3
    out = []
    for b in bound:
        if isinstance(b, str):
            if b.lower() not in _BOUND:
                raise ValueError("Unknown bound {}".format(b))
            out.append(_BOUND[b.lower()])
        else:
            out.append(int(b))
    while len(out) < 3:
        out.append(out[-1])
    return out[:3]


def _prep(input, grid, dim=3):
    """api.py:81-118 _preproc: broadcast batch dims, default channel of 1.  Returns (x (B,C,*in), g (B,*out,3), info)."""
    grid_spatial = tuple(grid.shape[-dim - 1:-1])
    grid_batch = tuple(grid.shape[:-dim - 1])
    in_spatial = tuple(input.shape[-dim:])
    channel = 0 if input.dim() == dim else input.shape[-dim - 1]
    in_batch = tuple(input.shape[:-dim - 1]) if input.dim() > dim else ()
    batch = torch.broadcast_shapes(grid_batch, in_batch)
    g = grid.to(torch.float32).expand(*batch, *grid_spatial, dim).reshape(-1, *grid_spatial, dim).contiguous()
    x = input.to(torch.float32).expand(*batch, channel or 1, *in_spatial).reshape(-1, channel or 1, *in_spatial).contiguous()
    return x, g, (batch, channel, in_spatial, grid_spatial)


def _opts(interpolation, bound, extrapolate):
    orders = interpolation if isinstance(interpolation, (list, tuple)) else [interpolation]
    if any(_INTER.get(o, o) != 1 for o in orders):
        raise NotImplementedError("only first-order (linear) interpolation is implemented for pull/push/grad")
    ext = {False: 0, True: 1, "no": 0, "yes": 1, "hist": 2}.get(extrapolate, extrapolate)
    return (C.c_int * 3)(*_bound_list(bound)), int(ext)


def _pull_raw(x, g, b, ext):
    B, Cc = x.shape[0], x.shape[1]
    gs = tuple(g.shape[1:4])
    out = torch.empty((B, Cc) + gs, dtype=torch.float32, device=x.device)
    L.check(L.load().bfm_grid_pull3d_linear(L.ptr(x), B, Cc, x.shape[2], x.shape[3], x.shape[4], L.ptr(g), B, gs[0],
                                            gs[1], gs[2], b, ext, L.ptr(out), L.stream_ptr()), "grid_pull3d_linear")
    return out


def _push_raw(x, g, shape, b, ext):
    B, Cc = x.shape[0], x.shape[1]
    out = torch.zeros((B, Cc) + tuple(shape), dtype=torch.float32, device=x.device)
    L.check(L.load().bfm_grid_push3d_linear(L.ptr(x), B, Cc, x.shape[2], x.shape[3], x.shape[4], L.ptr(g), B, shape[0],
                                            shape[1], shape[2], b, ext, L.ptr(out), L.stream_ptr()), "grid_push3d_linear")
    return out


def _grad_raw(x, g, b, ext):
    B, Cc = x.shape[0], x.shape[1]
    gs = tuple(g.shape[1:4])
    out = torch.empty((B, Cc) + gs + (3,), dtype=torch.float32, device=x.device)
    L.check(L.load().bfm_grid_grad3d_linear(L.ptr(x), B, Cc, x.shape[2], x.shape[3], x.shape[4], L.ptr(g), B, gs[0],
                                            gs[1], gs[2], b, ext, L.ptr(out), L.stream_ptr()), "grid_grad3d_linear")
    return out


class _GridPull(torch.autograd.Function):
    """autograd.py:125-152 + pushpull.py:262-285: d/dinput = push(grad), d/dgrid = sum_c grad * grid_grad(input)."""

    @staticmethod
    def forward(ctx, x, g, b, ext):
        ctx.opt = (b, ext)
        ctx.save_for_backward(x, g)
        return _pull_raw(x, g, b, ext)

    @staticmethod
    def backward(ctx, grad):
        x, g = ctx.saved_tensors
        b, ext = ctx.opt
        grad = grad.contiguous().to(torch.float32)
        gi = _push_raw(grad, g, x.shape[2:], b, ext) if ctx.needs_input_grad[0] else None
        gg = (_grad_raw(x, g, b, ext) * grad.unsqueeze(-1)).sum(dim=1) if ctx.needs_input_grad[1] else None
        return gi, gg, None, None


class _GridPush(torch.autograd.Function):
    """autograd.py:155-190 + pushpull.py:288-310: d/dinput = pull(grad), d/dgrid = sum_c input * grid_grad(grad)."""

    @staticmethod
    def forward(ctx, x, g, shape, b, ext):
        ctx.opt = (b, ext)
        ctx.save_for_backward(x, g)
        return _push_raw(x, g, shape, b, ext)

    @staticmethod
    def backward(ctx, grad):
        x, g = ctx.saved_tensors
        b, ext = ctx.opt
        grad = grad.contiguous().to(torch.float32)
        gi = _pull_raw(grad, g, b, ext) if ctx.needs_input_grad[0] else None
        gg = (_grad_raw(grad, g, b, ext) * x.unsqueeze(-1)).sum(dim=1) if ctx.needs_input_grad[1] else None
        return gi, gg, None, None, None


def _require_cuda(t, what):
    if t.device.type != "cuda":
        raise L.BfmError("%s runs on a HIP device only; there is no CPU fallback in the product path" % what)


def grid_push(input, grid, shape=None, interpolation="linear", bound="zero", extrapolate=False, prefilter=False):
    """api.py:203-275: splat `input` (..., [channel], *spatial) at the coordinates `grid` (..., *spatial, 3) into a
    volume of `shape` (default: the input's spatial shape).  Differentiable w.r.t. input and grid."""
    _require_cuda(input, "grid_push")
    b, ext = _opts(interpolation, bound, extrapolate)
    x, g, (batch, channel, in_spatial, grid_spatial) = _prep(input, grid)
    if in_spatial != grid_spatial:
        raise ValueError("Input and grid should have the same spatial shape")
    shape = tuple(int(v) for v in (shape if shape is not None else in_spatial))
    out = _GridPush.apply(x, g, shape, b, ext)
    out_channel = [channel] if channel else ([1] if batch else [])
    return out.reshape(*batch, *out_channel, *shape)


def grid_count(grid, shape=None, interpolation="linear", bound="zero", extrapolate=False):
    """api.py:278-340: push of an image of ones (the Jacobian-free 'count' image)."""
    _require_cuda(grid, "grid_count")
    gs = tuple(grid.shape[-4:-1])
    ones = torch.ones(tuple(grid.shape[:-4]) + (1,) + gs, dtype=torch.float32, device=grid.device)
    out = grid_push(ones, grid, shape, interpolation, bound, extrapolate)
    return out


def grid_grad(input, grid, interpolation="linear", bound="zero", extrapolate=False, prefilter=False):
    """api.py:343-412: spatial gradient of the interpolated image at `grid`: (..., [channel], *spatial_out, 3).
    Forward only (its own backward needs pushgrad / hess, not implemented)."""
    _require_cuda(input, "grid_grad")
    b, ext = _opts(interpolation, bound, extrapolate)
    x, g, (batch, channel, in_spatial, grid_spatial) = _prep(input.detach(), grid.detach())
    out = _grad_raw(x, g, b, ext)
    out_channel = [channel] if channel else ([1] if batch else [])
    return out.reshape(*batch, *out_channel, *grid_spatial, 3)


def grid_pull(input, grid, interpolation="linear", bound="zero", extrapolate=False, prefilter=False):
    """Sample `input` at the voxel coordinates in `grid`.

    input: (..., [channel], *spatial) ; grid: (..., *spatial_out, 3).  Returns (..., [channel], *spatial_out).
    Computes in fp32 like the reference's custom_fwd(cast_inputs=torch.float32)."""
    orders = interpolation if isinstance(interpolation, (list, tuple)) else [interpolation]
    if any(_INTER.get(o, o) != 1 for o in orders):
        raise NotImplementedError("only first-order (linear) interpolation is on the hot path")
    if prefilter:
        pass                      # order 1: the spline prefilter is the identity
    if grid.shape[-1] != 3:
        raise NotImplementedError("only 3-D grids are on the hot path")
    if input.device.type != "cuda":
        raise L.BfmError("grid_pull runs on a HIP device only; there is no CPU fallback in the product path")
    b, ext = _opts(interpolation, bound, extrapolate)
    x, g, (batch, channel, in_spatial, grid_spatial) = _prep(input, grid)
    out = _GridPull.apply(x, g, b, ext)                          # differentiable w.r.t. input and grid (first order)
    out_channel = [channel] if channel else ([1] if batch else [])
    return out.reshape(*batch, *out_channel, *grid_spatial)


# ----------------------------------------------------------------------------- resize (bspline_zooming)
_SPLINE_BOUND = {"nearest": 1, "replicate": 1, "repeat": 1, "border": 1, "dct2": 3, "reflect": 3, "reflection": 3,
                 "neumann": 3, 1: 1, 3: 3}


_PREFILTER_CACHE = {}


def _prefilter_scalars_dev(n, dev):
    """_prefilter_scalars with the fp32 table already on the device, cached per (n, device)."""
    key = (int(n), str(dev))
    if key not in _PREFILTER_CACHE:
        z, gain, pole_last, init_scale, final_scale, w = _prefilter_scalars(n)
        _PREFILTER_CACHE[key] = (z, gain, pole_last, init_scale, final_scale, w.to(dev) if n > 2 else None)
    return _PREFILTER_CACHE[key]


def _prefilter_scalars(n):
    """The host scalars of coeff.py's cubic DCT-II prefilter for a line of n samples (:55-60, :141-175, :218-226)."""
    import math
    z = math.sqrt(3.0) - 2.0
    gain = (1.0 - z) * (1.0 - 1.0 / z)
    polen = z ** n
    pole_last = polen * (1 + 1 / (z + polen * polen))
    init_scale = z / (1 - polen * polen)
    final_scale = z / (z - 1)
    zt = torch.as_tensor(z, dtype=torch.float32)
    w = (zt.pow(torch.arange(1, n - 1, dtype=torch.float32)) +
         zt.pow(torch.arange(2 * n - 2, n, -1, dtype=torch.float32)))            # fp32 table exactly as the reference builds it
    return z, gain, pole_last, init_scale, final_scale, w


_SPLINE_ORDER = {"nearest": 0, "linear": 1, "quadratic": 2, "cubic": 3, "fourth": 4, "fifth": 5, "sixth": 6, "seventh": 7}


def spline_coeff_nd(input, interpolation="linear", bound="dct2", dim=None, inplace=False):
    """utils/interpol/api.py:386-432 -> coeff.py:315-344 for cubic splines with DCT-II ('nearest' / 'dct2') conditions over
    the last three dimensions (`dim` None or 3); orders 0 / 1 return the input, as the reference does."""
    inp = input
    if inp.device.type != "cuda":
        raise L.BfmError("spline_coeff_nd runs on a HIP device only")
    order = interpolation[0] if isinstance(interpolation, (list, tuple)) else interpolation
    order = _SPLINE_ORDER.get(order.lower(), None) if isinstance(order, str) else int(order)
    dim = 3 if dim is None else dim
    if order in (0, 1):
        return inp if inplace else inp.clone()
    if order != 3 or dim != 3:
        raise NotImplementedError("only cubic 3-D prefiltering is implemented")
    b = bound if isinstance(bound, (str, int)) else bound[0]
    b = _SPLINE_BOUND.get(b.lower() if isinstance(b, str) else b)
    if b is None:
        raise NotImplementedError("spline prefilter: only 'nearest'/'dct2' boundary conditions")
    lead = inp.shape[:-3]
    vols = inp.to(torch.float32).reshape((-1,) + tuple(inp.shape[-3:])).contiguous()
    if not inplace or vols.data_ptr() != inp.data_ptr():
        vols = vols.clone()
    lib = L.load()
    nx, ny, nz = vols.shape[-3:]
    for v in vols:
        for axis, n in enumerate((nx, ny, nz)):
            if n == 1:
                continue
            z, gain, pole_last, init_scale, final_scale, wd = _prefilter_scalars_dev(n, inp.device)
            L.check(lib.bfm_bspline3_prefilter_axis(L.ptr(v), nx, ny, nz, axis, b, z, gain, L.ptr(wd), pole_last,
                                                    init_scale, final_scale, L.stream_ptr()), "bspline3_prefilter")
    return vols.reshape(tuple(lead) + (nx, ny, nz))


def _make_list(x, n=None):
    x = list(x) if isinstance(x, (list, tuple)) else [x]
    if n is not None and len(x) < n:
        x = x + [x[-1]] * (n - len(x))
    return x


def resize(image, factor=None, shape=None, anchor="c", interpolation=1, prefilter=True, **kwargs):
    """utils/interpol/resize.py:13-119.  image: (..., X, Y, Z).  Cubic (interpolation=3) runs as prefilter + three
    separable 4-tap passes; linear builds the grid and goes through grid_pull like the reference."""
    if image.device.type != "cuda":
        raise L.BfmError("resize runs on a HIP device only; there is no CPU fallback in the product path")
    factor = _make_list(factor) if factor else []
    shape = _make_list(shape) if shape else []
    anchor = _make_list(anchor)
    nb_dim = max(len(factor), len(shape), len(anchor)) or (image.dim() - 2)
    if nb_dim != 3:
        raise NotImplementedError("only 3-D resize is implemented")
    anchor = [a[0].lower() for a in _make_list(anchor, nb_dim)]
    inshape = image.shape[-nb_dim:]
    if factor:
        factor = _make_list(factor, nb_dim)
    elif not shape:
        raise ValueError("One of `factor` or `shape` must be provided")
    if shape:
        shape = _make_list(shape, nb_dim)
    else:
        shape = [int(i * f) for i, f in zip(inshape, factor)]
    if not factor:
        factor = [o / i for o, i in zip(shape, inshape)]
    lin = []
    for anch, f, inshp, outshp in zip(anchor, factor, inshape, shape):         # fp32 on the host: same values as torch CPU
        if anch == "c":
            lin.append(torch.linspace(0, inshp - 1, outshp, dtype=torch.float32))
        elif anch == "e":
            scale = inshp / outshp
            shift = 0.5 * (scale - 1)
            lin.append(torch.arange(0., outshp, dtype=torch.float32) * scale + shift)
        elif anch == "f":
            lin.append(torch.arange(0., outshp, dtype=torch.float32) / f)
        elif anch == "l":
            shift = (inshp - 1) - (outshp - 1) / f
            lin.append(torch.arange(0., outshp, dtype=torch.float32) / f + shift)
        else:
            raise ValueError("Unknown anchor {}".format(anch))
    bound = kwargs.get("bound", "nearest")
    order = kwargs.get("interpolation", interpolation)
    prefilter = kwargs.get("prefilter", prefilter)
    if order in (1, "linear"):
        grid = torch.stack(torch.meshgrid(*[v.to(image.device) for v in lin], indexing="ij"), dim=-1)
        return grid_pull(image, grid, interpolation=1, bound=bound, extrapolate=kwargs.get("extrapolate", True))
    if order not in (3, "cubic"):
        raise NotImplementedError("resize: interpolation orders 1 and 3 are implemented")
    if not kwargs.get("extrapolate", True):
        raise NotImplementedError("resize: cubic path assumes extrapolate=True (the reference default here)")
    b = bound if isinstance(bound, (str, int)) else bound[0]
    b = _SPLINE_BOUND.get(b.lower() if isinstance(b, str) else b)
    if b is None:
        raise NotImplementedError("resize: cubic path supports 'nearest'/'dct2' bounds")
    coeff = spline_coeff_nd(image, interpolation=3, bound=b, dim=3) if prefilter else image.to(torch.float32)
    lead = coeff.shape[:-3]
    vols = coeff.reshape((-1,) + tuple(coeff.shape[-3:])).contiguous()
    lib = L.load()
    coords = [v.to(image.device) for v in lin]
    outs = []
    for v in vols:
        cur = v
        dims = list(v.shape)
        for axis in range(3):
            nxt_dims = list(dims)
            nxt_dims[axis] = shape[axis]
            nxt = torch.empty(nxt_dims, dtype=torch.float32, device=image.device)
            L.check(lib.bfm_bspline3_resample_axis(L.ptr(cur), dims[0], dims[1], dims[2], axis, L.ptr(coords[axis]),
                                                   shape[axis], b, L.ptr(nxt), L.stream_ptr()), "bspline3_resample")
            cur, dims = nxt, nxt_dims
        outs.append(cur)
    out = torch.stack(outs, 0) if len(outs) > 1 else outs[0][None]
    return out.reshape(tuple(lead) + tuple(shape))
