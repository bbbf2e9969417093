"""Device mirrors of the pre-processing helpers of utils/misc.py that prepare_image chains around the inference path
(SURVEY N1): torch_resize (Gaussian pre-blur + anisotropic linear zoom, utils/misc.py:1051-1187),
myzoom_torch_anisotropic (:1051-1116), align_volume_to_ref (:1207-1247), get_ras_axes (:226-235).
Same names, arguments and return values; volumes live on the HIP device, affines stay NumPy on the host.
There is no CPU fallback: the kernels come from libbrainfm_hip.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import generator_utils as GU


# ----------------------------------------------------------------------------- orientation
def get_ras_axes(aff, n_dims=3):
    """utils/misc.py:226-235."""
    aff_inverted = np.linalg.inv(aff)
    return np.argmax(np.absolute(aff_inverted[0:n_dims, 0:n_dims]), axis=0)


def align_plan(shape, aff, aff_ref=None, n_dims=3):
    """Host half of align_volume_to_ref: (perm, flip, aligned affine) such that
    aligned[i0,i1,i2] = volume[j] with j[perm[a]] = flip[a] ? n-1-ia : ia  (utils/misc.py:1207-1247)."""
    aff_flo = aff.copy()
    if aff_ref is None:
        aff_ref = np.eye(4)
    ras_axes_ref = get_ras_axes(aff_ref, n_dims=n_dims)
    ras_axes_flo = get_ras_axes(aff_flo, n_dims=n_dims)
    aff_flo[:, ras_axes_ref] = aff_flo[:, ras_axes_flo]
    axes = list(range(n_dims))                               # axes[k] = source axis now sitting at position k
    for i in range(n_dims):
        if ras_axes_flo[i] != ras_axes_ref[i]:
            a, b = int(ras_axes_flo[i]), int(ras_axes_ref[i])
            axes[a], axes[b] = axes[b], axes[a]              # torch.swapaxes(volume, a, b)
            j = int(np.where(ras_axes_flo == ras_axes_ref[i])[0][0])
            ras_axes_flo[j], ras_axes_flo[i] = ras_axes_flo[i], ras_axes_flo[j]
    new_shape = [shape[axes[k]] for k in range(n_dims)]
    dot_products = np.sum(aff_flo[:3, :3] * aff_ref[:3, :3], axis=0)
    flip = [0] * n_dims
    for i in range(n_dims):
        if dot_products[i] < 0:
            flip[i] = 1
            aff_flo[:, i] = -aff_flo[:, i]
            aff_flo[:3, 3] = aff_flo[:3, 3] - aff_flo[:3, i] * (new_shape[i] - 1)
    return axes, flip, aff_flo


def align_volume_to_ref(volume, aff, aff_ref=None, return_aff=False, n_dims=3):
    """utils/misc.py:1207-1247 -- the swaps and flips run as one gather kernel."""
    GU._require_cuda(volume, "align_volume_to_ref")
    if volume.dim() != 3 or n_dims != 3:
        raise L.BfmError("align_volume_to_ref: 3-D volumes only on the device path")
    perm, flip, aff_flo = align_plan(tuple(volume.shape), aff, aff_ref, n_dims)
    if perm == [0, 1, 2] and not any(flip):
        out = volume
    else:
        src = volume.to(torch.float32).contiguous()
        out = torch.empty([src.shape[p] for p in perm], dtype=torch.float32, device=src.device)
        L.check(L.load().bfm_permute_flip3d(L.ptr(src), src.shape[0], src.shape[1], src.shape[2], (C.c_int * 3)(*perm),
                                            (C.c_int * 3)(*flip), L.ptr(out), L.stream_ptr()), "permute_flip3d")
        out = out.to(volume.dtype) if volume.dtype != torch.float32 else out
    if return_aff:
        return out, aff_flo
    return out


# ----------------------------------------------------------------------------- resize
def aniso_zoom_tables(n, newsize):
    """Per-axis (floor, ceil, w_floor, w_ceil) of myzoom_torch_anisotropic, utils/misc.py:1060-1090
    (fp32 torch.arange as ATen's CPU kernel evaluates it)."""
    factor = float(newsize) / float(n)
    delta = (1.0 - factor) / (2.0 * factor)
    v = GU.torch_cpu_arange_f32(delta, delta + newsize / factor, 1.0 / factor)[:newsize]
    v = np.where(v < 0, np.float32(0), v)
    v = np.where(v > n - 1, np.float32(n - 1), v).astype(np.float32)
    f = np.floor(v).astype(np.int32)
    c = np.minimum(f + 1, n - 1).astype(np.int32)
    wc = (v - f.astype(np.float32)).astype(np.float32)
    wf = (np.float32(1) - wc).astype(np.float32)
    return f, c, wf, wc


_ANISO_CACHE = {}
_GAUSS_CACHE = {}


def _aniso_tables_dev(n, newsize, dev):
    key = (int(n), int(newsize), str(dev))
    if key not in _ANISO_CACHE:
        if len(_ANISO_CACHE) > 256:
            _ANISO_CACHE.clear()
        _ANISO_CACHE[key] = [torch.from_numpy(v).to(dev) for v in aniso_zoom_tables(n, newsize)]
    return _ANISO_CACHE[key]


def myzoom_torch_anisotropic(X, aff, newsize):
    """utils/misc.py:1051-1116: separable linear zoom to an explicit size (one fused kernel)."""
    GU._require_cuda(X, "myzoom_torch_anisotropic")
    lib = L.load()
    dev = X.device
    squeeze = X.dim() == 3
    Xc = (X[..., None] if squeeze else X).to(torch.float32).contiguous()
    nx, ny, nz, cc = Xc.shape
    newsize = [int(v) for v in newsize]
    factors = np.array(newsize) / np.array([nx, ny, nz])
    keep = []
    axes = (L.ZoomAxis * 3)()
    for a, n in enumerate((nx, ny, nz)):
        t = _aniso_tables_dev(n, newsize[a], dev)
        if len(t[0]) != newsize[a]:
            raise L.BfmError("zoom table length %d != requested size %d" % (len(t[0]), newsize[a]))
        keep.append(t)
        axes[a] = L.ZoomAxis(*[v.data_ptr() for v in t])
    out = torch.empty((newsize[0], newsize[1], newsize[2], cc), dtype=torch.float32, device=dev)
    L.check(lib.bfm_zoom_linear(L.ptr(Xc), nx, ny, nz, cc, axes, newsize[0], newsize[1], newsize[2], L.ptr(out),
                                L.stream_ptr()), "zoom_linear")
    Y = out[..., 0] if cc == 1 else out
    if aff is not None:
        aff_new = aff.copy()
        for c in range(3):
            aff_new[:-1, c] = aff_new[:-1, c] / factors[c]
        aff_new[:-1, -1] = aff_new[:-1, -1] - aff[:-1, :-1] @ (0.5 - 0.5 / factors)
        return Y, aff_new
    return Y


def resize_plan(shape, aff, resolution, power_factor_at_half_width=5):
    """Host half of torch_resize: (newsize, sigmas) -- utils/misc.py:1124-1130."""
    voxsize = np.sqrt(np.sum(aff[:-1, :-1] ** 2, axis=0))
    newsize = np.round(np.array(shape[0:3]) * (voxsize / resolution)).astype(int)
    factors = np.array(shape[0:3]) / np.array(newsize)
    k = np.log(power_factor_at_half_width) / np.pi
    sigmas = k * factors
    sigmas[sigmas <= k] = 0
    return newsize, sigmas


def torch_resize(I, aff, resolution, power_factor_at_half_width=5, dtype=torch.float32, slow=False):
    """utils/misc.py:1118-1187: per-axis zero-padded Gaussian (only where the volume is being down-sampled), then the
    anisotropic linear zoom.  I: (X,Y,Z) or (X,Y,Z,C) on the device."""
    GU._require_cuda(I, "torch_resize")
    if I.dim() not in (3, 4):
        raise Exception("torch_resize works with 3D or 3D+label volumes")
    lib = L.load()
    newsize, sigmas = resize_plan(tuple(I.shape), aff, resolution, power_factor_at_half_width)
    no_channels = I.dim() == 3
    chans = [I] if no_channels else [I[..., c] for c in range(I.shape[3])]
    outs, aff2 = [], None
    for It in chans:
        cur = It.to(torch.float32).contiguous()
        nx, ny, nz = cur.shape
        for d in range(3):
            if sigmas[d] > 0:
                gkey = (float(sigmas[d]), str(cur.device))
                if gkey not in _GAUSS_CACHE:
                    sl = np.ceil(sigmas[d] * 2.5).astype(int)
                    v = np.arange(-sl, sl + 1)
                    gauss = np.exp((-(v / sigmas[d]) ** 2 / 2))
                    _GAUSS_CACHE[gkey] = torch.tensor(gauss / np.sum(gauss), device=cur.device, dtype=torch.float32)
                kernel = _GAUSS_CACHE[gkey]
                nxt = torch.empty_like(cur)
                L.check(lib.bfm_conv1d_axis(L.ptr(cur), nx, ny, nz, d, L.ptr(kernel), kernel.numel(), L.ptr(nxt),
                                            L.stream_ptr()), "conv1d_axis")
                cur = nxt
        z, aff2 = myzoom_torch_anisotropic(cur, aff, newsize)
        outs.append(z)
    if no_channels:
        return outs[0], aff2
    return torch.stack(outs, dim=-1), aff2
