"""Host-side mirror of the part of ``utils/interpol`` (vendored torch-interpol) that is on the hot
path: ``grid_pull`` with linear interpolation -> ``iso1.pull3d`` (utils/interpol/api.py:137-200,
autograd.py:125-152, pushpull.py:35-66, iso1.py:28-133, bounds.py:24-89, jit_utils.py:241-255).

Forward only: the generator never differentiates through it (SURVEY 'next' row N4 covers push/grad
and the B-spline prefilter).
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
    dim = 3
    # _preproc (api.py:81-118): broadcast batch dims, default channel of 1
    grid_spatial = tuple(grid.shape[-dim - 1:-1])
    grid_batch = tuple(grid.shape[:-dim - 1])
    in_spatial = tuple(input.shape[-dim:])
    channel = 0 if input.dim() == dim else input.shape[-dim - 1]
    in_batch = tuple(input.shape[:-dim - 1]) if input.dim() > dim else ()
    batch = torch.broadcast_shapes(grid_batch, in_batch)
    g = grid.to(torch.float32).expand(*batch, *grid_spatial, dim).reshape(-1, *grid_spatial, dim).contiguous()
    x = input.to(torch.float32).expand(*batch, channel or 1, *in_spatial).reshape(-1, channel or 1, *in_spatial).contiguous()
    B, Cc = x.shape[0], x.shape[1]
    ext = {False: 0, True: 1, "no": 0, "yes": 1, "hist": 2}.get(extrapolate, extrapolate)
    b = (C.c_int * 3)(*_bound_list(bound))
    out = torch.empty((B, Cc) + grid_spatial, dtype=torch.float32, device=x.device)
    L.check(L.load().bfm_grid_pull3d_linear(L.ptr(x), B, Cc, in_spatial[0], in_spatial[1], in_spatial[2], L.ptr(g), B,
                                            grid_spatial[0], grid_spatial[1], grid_spatial[2], b, int(ext), L.ptr(out),
                                            L.stream_ptr()), "grid_pull3d_linear")
    out_channel = [channel] if channel else ([1] if batch else [])
    return out.reshape(*batch, *out_channel, *grid_spatial)
