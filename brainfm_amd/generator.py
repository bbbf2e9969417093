"""Host-side mirror of ``Generator`` (``BaseGen`` / ``BrainIDGen.__getitem__``): on-the-fly synthesis of
training samples from label maps (Generator/datasets.py:28-757, Generator/__init__.py:18-21).

Same orchestration, RNG call order on ``np.random`` and return structure
``(datasets_num, dataset_name, input_mode, target, sample[s])`` as the reference; the per-voxel work
(deformation grid, trilinear / nearest resampling, contrast synthesis, bias field, blur, noise, Perlin
shapes, advection) runs in libbrainfm_hip.so through ``generator_utils`` / ``shapeid``.

Volume sources are objects exposing ``get_fdata()``, ``shape`` and ``affine`` (nibabel's interface) or
plain arrays: NIfTI/MGZ file I/O is SURVEY 'next' row N3, outside this path, so cases are passed in
memory as ``{name, 'Gen': labels, 'T1': vol, ...}`` dicts instead of being globbed from a data root.
"""
import ctypes as C
import os
import random
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import _lib as L
from . import generator_utils as GU
from . import interpol as IP
from .engine import LABELS_FULL, LABELS_LEFT
from .shapeid import AdvDiffPDE, generate_shape_3d, generate_shape_3d_dev

n_neutral_labels = 20

ct_brightness_group = {"darker": [4, 5, 14, 15, 24, 31, 72], "dark": [2, 7, 16, 77, 30],
                       "bright": [3, 8, 17, 18, 28, 10, 11, 12, 13, 26], "brighter": []}


class ArrayVolume:
    """Minimal nibabel-like wrapper around an in-memory array."""

    def __init__(self, data, affine=None):
        self._d = np.asarray(data)
        self.shape = self._d.shape
        self.affine = np.eye(4) if affine is None else affine

    def get_fdata(self):
        return self._d.astype(np.float64)


def _vol(v):
    return v if hasattr(v, "get_fdata") else ArrayVolume(v)


class DeviceVolumes:
    """Case volumes resident in HBM.  The reference opens every NIfTI file of the case for every item and crops on the
    host (Generator/utils.py:296-305: nib.load -> get_fdata()[box] -> torch.tensor -> device); a case is ~10 volumes of
    ~30 MB, so a few hundred cases fit in a fraction of the 288 GB and an item never touches the host copy again: the
    crop becomes a box inside the resident volume.  Least-recently-used volumes are dropped beyond `budget` bytes
    (BFM_GEN_CACHE_GB, default 64).  A volume is keyed by the identity of its source object AND a cheap content stamp
    (shape, dtype, data address and the sum of ~256 strided samples, taken at every request: a few microseconds): an array
    re-filled in place -- re-used buffers, on-the-fly pre-processing -- gets a fresh resident copy instead of silently
    serving the old one.  The stamp is a BEST-EFFORT guard, not a hash: it samples ~256 elements of C-contiguous
    arrays only, so an in-place edit that leaves the sampled voxels alone (a lesion mask touched locally, a label volume
    whose sampled voxels stay background) is NOT seen -- after editing a source array in place call `forget(vol)` (one
    volume) or `invalidate()` (all) before the next item (a full crc32 of a 160^3 volume costs 5-15 ms per
    request against 4 ms for the whole item, which is why it is not the default).  mean 0 / scale 1 'prep' copies
    of a volume without NaNs are the 'f32' copy itself."""

    def __init__(self, device, budget=None):
        self.device = device
        self.budget = int(float(os.environ.get("BFM_GEN_CACHE_GB", "64")) * 2 ** 30) if budget is None else budget
        self.items = OrderedDict()                        # (id(source), kind, ...) -> (tensor, source kept alive, stamp)
        self.bytes = 0

    @staticmethod
    def _stamp(src):
        """What changes when an in-memory source is re-filled: sampled content, not only identity (file-backed sources:
        their array proxy's identity and shape)."""
        arr = src if isinstance(src, np.ndarray) else getattr(src, "_d", None)
        if not isinstance(arr, np.ndarray) or arr.size == 0:
            return (getattr(src, "shape", None),)
        tot = 0.0
        if arr.flags.c_contiguous and arr.dtype.kind in "fiub":
            flat = arr.reshape(-1)
            tot = float(np.nansum(flat[::max(1, flat.size // 256)].astype(np.float64)))
        return (arr.shape, arr.dtype.str, arr.ctypes.data, tot)

    def invalidate(self):
        """Drop every resident copy (the sources changed in a way the sampled stamp may not see)."""
        self.forget(None)

    @staticmethod
    def _host_array(src):
        if isinstance(src, np.ndarray):
            arr = src
        elif hasattr(src, "dataobj"):
            arr = np.asarray(src.dataobj)
            if arr.dtype == object or arr.ndim < 3:             # a proxy NumPy cannot read as an array: take the float copy
                arr = src.get_fdata()
        else:
            arr = src.get_fdata()
        while arr.ndim > 3 and arr.shape[-1] == 1:
            arr = arr[..., 0]
        if not (np.issubdtype(arr.dtype, np.floating) or np.issubdtype(arr.dtype, np.integer)):
            arr = arr.astype(np.float64)
        return arr

    def forget(self, vol=None):
        """Drop the resident copies of one source (or of all of them)."""
        src = None if vol is None else (vol._d if isinstance(vol, ArrayVolume) else vol)
        for key in [k for k in self.items if src is None or k[0] == id(src)]:
            t = self.items.pop(key)[0]
            if not any(v[0] is t for v in self.items.values()):      # an aliased 'prep' / 'f32' pair is counted once
                self.bytes -= t.numel() * 4

    def get(self, vol, kind="f32", mean=0., scale=1.):
        """The whole volume as float32 (`kind` 'f32': what torch.tensor(get_fdata().astype(float), dtype=torch.float)
        holds, one rounding from the stored type), as int32 ('i32': .astype(int), truncation), or as read_and_deform
        prepares it ('prep': nan_to_num, then (I - mean) / scale, Generator/utils.py:303-308 -- elementwise, so applying it
        to the resident volume once gives every later crop the values the reference computes per item)."""
        src = vol._d if isinstance(vol, ArrayVolume) else vol
        key = (id(src), kind, float(mean), float(scale))
        stamp = self._stamp(src)
        hit = self.items.get(key)
        if hit is not None:
            if hit[2] == stamp:
                self.items.move_to_end(key)
                return hit[0]
            self.forget(src)                                        # re-filled in place since it was uploaded
        if kind == "prep" and mean == 0. and scale == 1.:           # identity transform: share the 'f32' copy when it has
            base = self.get(vol, "f32")                             # no NaN to scrub (one resident copy instead of two)
            if bool(torch.isfinite(base).all().item()):             # nan_to_num would change nothing
                self.items[key] = (base, src, stamp)
                return base
        arr = self._host_array(src)
        if kind in ("f32", "prep"):
            host = np.ascontiguousarray(arr, dtype=np.float32)
        else:
            host = np.ascontiguousarray(arr.astype(np.int64) if np.issubdtype(arr.dtype, np.floating) else arr,
                                        dtype=np.int32)
        t = torch.from_numpy(host).to(self.device)
        if kind == "prep":
            with torch.cuda.device(self.device):
                t = GU.ew_unary(L.EW_NAN_TO_NUM, t)
                if mean != 0. or scale != 1.:
                    t = GU.ew_unary(L.EW_SUB_DIV, t, float(mean), float(scale))
        nbytes = t.numel() * 4
        while self.items and self.bytes + nbytes > self.budget:
            old = self.items.popitem(last=False)[1][0]
            if not any(v[0] is old for v in self.items.values()):
                self.bytes -= old.numel() * 4
        self.items[key] = (t, src, stamp)
        self.bytes += nbytes
        return t


class DeviceDirection:
    """pathol_direction decided on the device: gm_mean > wm_mean from bfm_label_class_stats' four sums
    (datasets.py:392-404), read by bfm_pathology_encode_dev without a host round trip."""

    def __init__(self, stats):
        self.stats = stats

    def __bool__(self):
        st = self.stats.cpu().tolist()
        wm = st[0] / st[1] if st[1] else float("nan")
        gm = st[2] / st[3] if st[3] else float("nan")
        return gm > wm


class BaseGen(torch.utils.data.Dataset):
    """BaseGen, Generator/datasets.py:24-681."""

    def __init__(self, gen_args, device="cuda", cases=None):
        self.gen_args = gen_args
        self.synth_args = gen_args.generator
        self.shape_gen_args = gen_args.pathology_shape_generator
        self.real_image_args = getattr(gen_args, "real_image_generator", None)
        self.synth_image_args = getattr(gen_args, "synth_image_generator", None)
        self.augmentation_steps = vars(gen_args.augmentation_steps) if not isinstance(gen_args.augmentation_steps, (list, dict)) \
            else gen_args.augmentation_steps
        if isinstance(self.augmentation_steps, list):
            self.augmentation_steps = {"synth": self.augmentation_steps, "real": self.augmentation_steps}
        self.input_prob = getattr(gen_args, "modality_probs", None)
        self.device = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
        if self.device.type != "cuda":
            raise L.BfmError("the generator runs on a HIP device only; there is no CPU fallback in the product path")
        self.cases = cases or []
        self.volumes = DeviceVolumes(self.device)
        self._batch = None                                # pending gather jobs while _targets collects them
        self._psum = (None, 0.0)                          # (target['pathology'] tensor, its sum) as the host knows it; the
                                                          # tensor is HELD, so its address cannot be handed to another one
        self.datasets_num = 1
        self.pathology_type = None
        self.hemis_mask = None
        self.prepare_tasks()
        self.prepare_grid()
        self.prepare_one_hot()

    def __len__(self):
        return len(self.cases)

    # -------------------------------------------------------------- setup (datasets.py:123-184)
    def prepare_tasks(self):
        self.tasks = [k for (k, v) in vars(self.gen_args.task).items() if v]
        if "bias_field" in self.tasks and "segmentation" not in self.tasks:
            self.tasks += ["segmentation"]
        if "pathology" in self.tasks and self.synth_args.augment_pathology and \
                getattr(self.synth_args, "random_shape_prob", 1.) <= 1.:
            self.t = torch.from_numpy(np.arange(self.shape_gen_args.max_nt) * self.shape_gen_args.dt)
            self.adv_pde = AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict={},
                                      BC=self.shape_gen_args.bc, dt=self.shape_gen_args.dt, device=self.device)
        else:
            self.t, self.adv_pde = None, None

    def prepare_grid(self):
        self.size = list(self.synth_args.size)
        self.res_training_data = np.array([1.0, 1.0, 1.0])

    def prepare_one_hot(self):
        labels = LABELS_LEFT if self.synth_args.left_hemis_only else LABELS_FULL
        n_labels = len(labels)
        lut = np.zeros(10000, dtype=np.int32)
        for l in range(n_labels):
            lut[labels[l]] = l
        self.n_labels = n_labels
        self.lut = torch.from_numpy(lut).to(self.device)
        nlat = int((n_labels - n_neutral_labels) / 2.0)
        self.vflip = np.concatenate([np.arange(n_neutral_labels), np.arange(n_neutral_labels + nlat, n_labels),
                                     np.arange(n_neutral_labels, n_neutral_labels + nlat)])

    # -------------------------------------------------------------- deformation (datasets.py:187-303)
    def random_affine_transform(self, shp):
        s = self.synth_args
        rotations = (2 * s.max_rotation * np.random.rand(3) - s.max_rotation) / 180.0 * np.pi
        shears = (2 * s.max_shear * np.random.rand(3) - s.max_shear)
        scalings = 1 + (2 * s.max_scaling * np.random.rand(3) - s.max_scaling)
        scaling_factor_distances = np.prod(scalings) ** .33333333333
        A = GU.make_affine_matrix(rotations, shears, scalings).astype(np.float32)
        if s.random_shift:
            max_shift = np.maximum((np.array(shp[0:3]) - self.size) / 2, 0).astype(np.float32)
            c2 = ((np.array(shp[0:3]) - 1) / 2).astype(np.float32) + (2 * (max_shift * np.random.rand(3)) - max_shift)
        else:
            c2 = ((np.array(shp[0:3]) - 1) / 2).astype(np.float32)
        return scaling_factor_distances, A, c2.astype(np.float32)

    def _random_nonlinear_small(self, photo_mode, spac):
        """The draws of random_nonlinear_transform (datasets.py:209-226): the low-resolution field and its zoom factors."""
        s = self.synth_args
        nonlin_scale = s.nonlin_scale_min + np.random.rand(1) * (s.nonlin_scale_max - s.nonlin_scale_min)
        size_F_small = np.round(nonlin_scale * np.array(self.size)).astype(int).tolist()
        if photo_mode:
            size_F_small[1] = np.round(self.size[1] / spac).astype(int)
        nonlin_std = s.nonlin_std_max * np.random.rand()
        Fsmall = GU.ew_unary(L.EW_AFFINE, GU.draws.randn([*size_F_small, 3], self.device),
                             float(np.float32(nonlin_std)), 0.0)
        return Fsmall, np.array(self.size) / size_F_small

    def random_nonlinear_transform(self, photo_mode, spac):
        Fsmall, factor = self._random_nonlinear_small(photo_mode, spac)
        F = GU.myzoom_torch(Fsmall, factor)
        if photo_mode:
            F[:, :, :, 1] = 0
        return F, None

    def deform_grid(self, shp, A, c2, F):
        """datasets.py:264-303: one kernel for affine(+nonlinear) coordinates, clamp and the six extrema; the
        host reads the crop box (the reference synchronises here too, :296-301) and the offsets are subtracted."""
        lib = L.load()
        sx, sy, sz = self.size
        dev = self.device
        xx, yy, zz = (torch.empty((sx, sy, sz), dtype=torch.float32, device=dev) for _ in range(3))
        mm = torch.empty(6, dtype=torch.float32, device=dev)
        ws = torch.empty(max(lib.bfm_deform_grid_workspace(sx, sy, sz), 64), dtype=torch.uint8, device=dev)
        Ah = (C.c_float * 9)(*[float(v) for v in np.asarray(A, np.float32).reshape(-1)])
        ch = (C.c_float * 3)(*[float(v) for v in np.asarray(c2, np.float32)])
        sh = (C.c_int * 3)(*[int(v) for v in shp[:3]])
        Fc = F.contiguous() if F is not None else None
        L.check(lib.bfm_deform_grid(L.ptr(Fc), sx, sy, sz, Ah, ch, sh, L.ptr(xx), L.ptr(yy), L.ptr(zz), L.ptr(mm),
                                    L.ptr(ws), ws.numel(), L.stream_ptr()), "deform_grid")
        m = mm.cpu().numpy()
        lo = np.floor(m[:3])
        hi = 1 + np.ceil(m[3:])
        xx = GU.ew_unary(L.EW_AFFINE, xx, 1.0, -float(lo[0]))
        yy = GU.ew_unary(L.EW_AFFINE, yy, 1.0, -float(lo[1]))
        zz = GU.ew_unary(L.EW_AFFINE, zz, 1.0, -float(lo[2]))
        x1, y1, z1 = (int(v) for v in lo)
        x2, y2, z2 = (int(v) for v in hi)
        return xx, yy, zz, x1, y1, z1, x2, y2, z2

    def deform_grid_zoomed(self, shp, A, c2, Fsmall, factor, photo_mode=False):
        """deform_grid(shp, A, c2, myzoom_torch(Fsmall, factor)) without the detour through HBM: pass 1 evaluates the zoomed
        field per voxel and leaves only the six extrema (the one read-back the reference has here too, datasets.py:296-301),
        pass 2 evaluates it again and writes the coordinates already shifted by the crop origin, and F.  Same expressions,
        same order: the bits of the two-step form.  Returns (F, [xx2, yy2, zz2, x1, y1, z1, x2, y2, z2])."""
        lib = L.load()
        sx, sy, sz = self.size
        dev = self.device
        Fs = Fsmall.to(torch.float32).contiguous()
        fnx, fny, fnz = Fs.shape[:3]
        factor = np.asarray(factor, dtype=np.float64) * np.ones(3)
        axes = (L.ZoomAxis * 3)()
        keep = []
        for a, n in enumerate((fnx, fny, fnz)):
            t = GU._zoom_tables_dev(n, float(factor[a]), dev)
            if len(t[0]) != self.size[a]:
                raise L.BfmError("zoom of the deformation field gives %d voxels along axis %d, not %d"
                                 % (len(t[0]), a, self.size[a]))
            keep.append(t)
            axes[a] = L.ZoomAxis(*[v.data_ptr() for v in t])
        Ah = (C.c_float * 9)(*[float(v) for v in np.asarray(A, np.float32).reshape(-1)])
        ch = (C.c_float * 3)(*[float(v) for v in np.asarray(c2, np.float32)])
        sh = (C.c_int * 3)(*[int(v) for v in shp[:3]])
        mm = torch.empty(6, dtype=torch.float32, device=dev)
        ws = GU.workspace(dev, lib.bfm_deform_zoom_workspace())
        L.check(lib.bfm_deform_zoom_minmax(L.ptr(Fs), fnx, fny, fnz, axes, int(bool(photo_mode)), sx, sy, sz, Ah, ch, sh,
                                           L.ptr(mm), L.ptr(ws), ws.numel(), L.stream_ptr()), "deform_zoom_minmax")
        m = mm.cpu().numpy()
        lo = np.floor(m[:3])
        hi = 1 + np.ceil(m[3:])
        loh = (C.c_float * 3)(*[float(v) for v in lo])
        xx, yy, zz = (torch.empty((sx, sy, sz), dtype=torch.float32, device=dev) for _ in range(3))
        F = torch.empty((sx, sy, sz, 3), dtype=torch.float32, device=dev)
        L.check(lib.bfm_deform_zoom_write(L.ptr(Fs), fnx, fny, fnz, axes, int(bool(photo_mode)), sx, sy, sz, Ah, ch, sh, loh,
                                          L.ptr(xx), L.ptr(yy), L.ptr(zz), L.ptr(F), L.stream_ptr()), "deform_zoom_write")
        x1, y1, z1 = (int(v) for v in lo)
        x2, y2, z2 = (int(v) for v in hi)
        return F, [xx, yy, zz, x1, y1, z1, x2, y2, z2]

    def generate_deformation(self, setups, shp):
        scaling_factor_distances, A, c2 = self.random_affine_transform(shp)
        if self.synth_args.nonlinear_transform:
            Fsmall, factor = self._random_nonlinear_small(setups["photo_mode"], setups["spac"])
            F, grid = self.deform_grid_zoomed(shp, A, c2, Fsmall, factor, setups["photo_mode"])
            Fneg = None
        else:
            F, Fneg = None, None
            grid = list(self.deform_grid(shp, A, c2, F))
        return {"scaling_factor_distances": scaling_factor_distances, "A": A, "c2": c2, "F": F, "Fneg": Fneg,
                "grid": grid, "shape": tuple(int(v) for v in shp[:3])}

    # -------------------------------------------------------------- contrast / setup (datasets.py:430-493)
    def get_contrast(self, photo_mode):
        """256-entry mean / std tables (host arithmetic on 256 numbers), datasets.py:430-464."""
        mus = (25 + 200 * GU.draws.rand(256)).numpy()
        sigmas = (5 + 20 * GU.draws.rand(256)).numpy()
        if np.random.rand() < self.synth_args.ct_prob:
            for grp, (a, b) in (("darker", (25, 10)), ("dark", (90, 20)), ("bright", (110, 20)), ("brighter", (150, 50))):
                v = a + b * GU.draws.rand(1)[0].item()
                for l in ct_brightness_group[grp]:
                    mus[l] = v
        if photo_mode or np.random.rand(1) < 0.5:
            mus[0] = 0
        v = (0.02 * np.arange(50)).astype(np.float32)
        mus[100:150] = mus[1] * (1 - v) + mus[2] * v
        mus[150:200] = mus[2] * (1 - v) + mus[3] * v
        mus[200:250] = mus[3] * (1 - v) + mus[4] * v
        mus[250] = mus[4]
        sigmas[100:150] = np.sqrt(sigmas[1] ** 2 * (1 - v) + sigmas[2] ** 2 * v)
        sigmas[150:200] = np.sqrt(sigmas[2] ** 2 * (1 - v) + sigmas[3] ** 2 * v)
        sigmas[200:250] = np.sqrt(sigmas[3] ** 2 * (1 - v) + sigmas[4] ** 2 * v)
        sigmas[250] = sigmas[4]
        return torch.from_numpy(mus.astype(np.float32)).to(self.device), \
            torch.from_numpy(sigmas.astype(np.float32)).to(self.device)

    def get_setup_params(self):
        s = self.synth_args
        hemis = "left" if s.left_hemis_only else "both"
        if s.low_res_only:
            photo_mode = False
        elif s.left_hemis_only:
            photo_mode = True
        else:
            photo_mode = np.random.rand() < s.photo_prob
        pathol_mode = np.random.rand() < s.pathology_prob
        pathol_random_shape = np.random.rand() < getattr(s, "random_shape_prob", 1.0)
        spac = 2.5 + 10 * np.random.rand() if photo_mode else None
        flip = np.random.randn() < s.flip_prob if not s.left_hemis_only else False      # quirk Q13: normal draw
        if photo_mode:
            resolution = np.array([self.res_training_data[0], spac, self.res_training_data[2]])
            thickness = np.array([self.res_training_data[0], 0.1, self.res_training_data[2]])
        else:
            resolution, thickness = GU.resolution_sampler(s.low_res_only)
        return {"resolution": resolution, "thickness": thickness, "photo_mode": photo_mode, "pathol_mode": pathol_mode,
                "pathol_random_shape": pathol_random_shape, "spac": spac, "flip": flip, "hemis": hemis}

    # -------------------------------------------------------------- targets (Generator/utils.py:296-477)
    def _box(self, vol_shape, grid):
        """The crop the reference reads, `[x1:x2, y1:y2, z1:z2]` clipped like a NumPy slice: (origin, dims)."""
        [_, _, _, x1, y1, z1, x2, y2, z2] = grid
        x2, y2, z2 = min(x2, vol_shape[0]), min(y2, vol_shape[1]), min(z2, vol_shape[2])
        return (x1, y1, z1), (x2 - x1, y2 - y1, z2 - z1)

    def _crop(self, vol, grid, dtype=torch.float32, cache=None):
        """The crop of a case volume as a device tensor (float32 or int32), cut out of the resident copy on the device
        (round 3: NumPy slice + conversion + upload per crop, 17 crops = 44 ms of host time per item).  `cache`: a dict
        that keeps the crop for the item (generate_sample crops the label map once per sample in the reference)."""
        kind = "f32" if dtype == torch.float32 else "i32"
        full = self.volumes.get(_vol(vol), kind)
        key = (full.data_ptr(), kind)
        if cache is not None and key in cache:
            return cache[key]
        (x1, y1, z1), (cx, cy, cz) = self._box(full.shape, grid)
        out = torch.empty((cx, cy, cz), dtype=full.dtype, device=self.device)
        L.check(L.load().bfm_crop3d(L.ptr(full), full.shape[0], full.shape[1], full.shape[2], x1, y1, z1, cx, cy, cz,
                                    L.ptr(out), L.stream_ptr()), "crop3d")          # a bit copy of 4-byte elements
        out = torch.squeeze(out)
        if dtype not in (torch.float32, torch.int32):
            out = out.to(dtype)
        if cache is not None:
            cache[key] = out
        return out

    def _job(self, vol, out, mean=0., scale=1., default_max=False, post_div=0., clamp=None, sign=0.,
             want_minmax=False, post=None):
        """One volume of a bfm_gather_targets launch (the resident volume already holds nan_to_num((I - mean) / scale));
        `post(scalars, j)` runs after the launch."""
        return {"vol": self.volumes.get(_vol(vol), "prep", mean, scale), "out": out, "pre": 0, "mean": 0.,
                "scale": 1., "default_max": bool(default_max), "post_div": float(post_div), "clamp": clamp,
                "sign": float(sign), "want_minmax": bool(want_minmax), "post": post}

    def _submit(self, jobs, deform_dict, flip):
        """Queue the jobs while _targets is collecting (one launch for all float targets of the item), else run them."""
        if self._batch is not None:
            self._batch.extend(jobs)
        else:
            self._run_gather(jobs, deform_dict, flip)

    def _run_gather(self, jobs, deform_dict, flip):
        lib = L.load()
        [xx2, yy2, zz2, x1, y1, z1, x2, y2, z2] = deform_dict["grid"]
        sx, sy, sz = xx2.shape
        box = (C.c_int * 6)(x1, y1, z1, x2, y2, z2)
        ws = GU.workspace(self.device, lib.bfm_gather_targets_workspace())
        by_shape = defaultdict(list)
        for j in jobs:
            by_shape[tuple(j["vol"].shape)].append(j)
        for shape, group in by_shape.items():
            for c0 in range(0, len(group), L.GATHER_MAX_JOBS):
                chunk = group[c0:c0 + L.GATHER_MAX_JOBS]
                arr = (L.GatherJob * len(chunk))()
                for k, j in enumerate(chunk):
                    lo, hi = j["clamp"] if j["clamp"] is not None else (0., 0.)
                    arr[k] = L.GatherJob(j["vol"].data_ptr(), j["out"].data_ptr(), j["mean"], j["scale"], j["pre"],
                                         int(j["default_max"]), j["post_div"], int(j["clamp"] is not None), float(lo),
                                         float(hi), j["sign"], int(j["want_minmax"]))
                scal = torch.empty(3 * L.GATHER_MAX_JOBS, dtype=torch.float64, device=self.device)
                L.check(lib.bfm_gather_targets(arr, len(chunk), shape[0], shape[1], shape[2], box, L.ptr(xx2), L.ptr(yy2),
                                               L.ptr(zz2), sx, sy, sz, int(bool(flip)), L.ptr(scal), L.ptr(ws),
                                               ws.numel(), L.stream_ptr()), "gather_targets")
                for k, j in enumerate(chunk):
                    if j["post"] is not None:
                        j["post"](scal, k)

    def read_and_deform(self, vol, deform_dict, default_max=False, mean=0., scale=1., flip=False):
        """read_and_deform, Generator/utils.py:296-322 (nan_to_num, (I - mean) / scale, default value = the crop's
        maximum on request, trilinear sample), straight from the resident volume."""
        out = torch.empty(tuple(deform_dict["grid"][0].shape), dtype=torch.float32, device=self.device)
        self._submit([self._job(vol, out, mean=mean, scale=scale, default_max=default_max)], deform_dict, flip)
        return out

    def _flip0(self, t):
        """torch.flip(t, [0]) of a 3-D fp32 tensor (a mirrored gather)."""
        tc = t.to(torch.float32).contiguous()
        out = torch.empty_like(tc)
        perm = (C.c_int * 3)(0, 1, 2)
        fl = (C.c_int * 3)(1, 0, 0)
        L.check(L.load().bfm_permute_flip3d(L.ptr(tc), tc.shape[0], tc.shape[1], tc.shape[2], perm, fl, L.ptr(out),
                                            L.stream_ptr()), "permute_flip3d")
        return out

    def _flip_axis1(self, t):
        """torch.flip(t, [1]) of a (1, sx, sy, sz) target (datasets.py:676-679, 752-755) through the same mirrored gather:
        an fp64 volume is flipped as its fp32 pairs along the last axis, which the flip along x leaves together."""
        if not (isinstance(t, torch.Tensor) and t.dim() == 4 and t.shape[0] == 1 and t.is_cuda
                and t.dtype in (torch.float32, torch.float64)):
            return torch.flip(t, [1])
        v = t[0].contiguous()
        if v.dtype == torch.float64:
            out = self._flip0(v.view(torch.float32)).view(torch.float64)
        else:
            out = self._flip0(v)
        return out[None]

    def read_and_deform_image(self, task, vol, setups, deform_dict):
        """Generator/utils.py:331-345: sample, I -= min, I /= max, flip -- min / max never leave the device."""
        out = torch.empty(tuple(deform_dict["grid"][0].shape), dtype=torch.float32, device=self.device)

        def normalise(scal, j):
            L.check(L.load().bfm_minmax_normalise(L.ptr(out), out.numel(),
                                                  C.c_void_p(scal.data_ptr() + 8 * (L.GATHER_MAX_JOBS + 2 * j)),
                                                  L.stream_ptr()), "minmax_normalise")

        self._submit([self._job(vol, out, want_minmax=True, post=normalise)], deform_dict, setups["flip"])
        return {task: out[None]}

    def read_and_deform_segmentation(self, vol, setups, deform_dict):
        """Generator/utils.py:402-425: nearest sample of the label volume, lut, one-hot (+ flip with the left / right
        channel swap), one kernel."""
        [xx2, yy2, zz2, x1, y1, z1, x2, y2, z2] = deform_dict["grid"]
        S = self.volumes.get(_vol(vol), "i32")
        sx, sy, sz = xx2.shape
        # written class by class: the reference's one_hot(...).permute([3, 0, 1, 2]) is a strided view of a channels-last
        # tensor, and making it contiguous for the criterion cost 5 ms per 160^3 sample
        out = torch.empty((self.n_labels, sx, sy, sz), dtype=torch.float32, device=self.device)
        box = (C.c_int * 6)(x1, y1, z1, x2, y2, z2)
        vflip = None
        if setups["flip"]:
            if getattr(self, "_vflip_dev", None) is None:
                self._vflip_dev = torch.as_tensor(self.vflip, dtype=torch.int32, device=self.device)
            vflip = self._vflip_dev
        L.check(L.load().bfm_gather_onehot_rows(L.ptr(S), S.shape[0], S.shape[1], S.shape[2], box, L.ptr(xx2), L.ptr(yy2),
                                           L.ptr(zz2), sx, sy, sz, int(bool(setups["flip"])), L.ptr(self.lut),
                                           self.lut.numel(), self.n_labels, L.ptr(vflip), L.ptr(out), L.stream_ptr()),
                "gather_onehot_rows")
        return {"segmentation": out}

    def read_and_deform_distance(self, vols, setups, deform_dict):
        """Generator/utils.py:376-400: (I - 128) / 20 sampled with the crop's maximum outside, / scaling factor, clamp;
        the maps are written straight into the stacked target (left / right swapped under a flip)."""
        shape = tuple(deform_dict["grid"][0].shape)
        out = torch.empty((len(vols),) + shape, dtype=torch.float32, device=self.device)
        order = list(range(len(vols)))
        if len(vols) == 4 and setups["flip"]:
            order = [2, 3, 0, 1]                              # (lp, lw, rp, rw) -> (rp, rw, lp, lw)
        m = float(self.gen_args.max_surf_distance)
        sf = float(np.float32(float(deform_dict["scaling_factor_distances"])))
        jobs = [self._job(vols[src], out[dst], mean=128., scale=20, default_max=True, post_div=sf, clamp=(-m, m))
                for dst, src in enumerate(order)]
        self._submit(jobs, deform_dict, setups["flip"])
        return {"distance": out}

    def read_and_deform_registration(self, vols, setups, deform_dict):
        """Generator/utils.py:462-473: I / 10000 sampled; under a flip the x map changes sign."""
        shape = tuple(deform_dict["grid"][0].shape)
        out = torch.empty((len(vols),) + shape, dtype=torch.float32, device=self.device)
        jobs = [self._job(v, out[i], scale=10000, sign=(-1. if (setups["flip"] and i == 0) else 0.))
                for i, v in enumerate(vols)]
        self._submit(jobs, deform_dict, setups["flip"])
        return {"registration": out}

    def read_and_deform_pathology(self, source, setups, deform_dict, augment, thres):
        """Generator/utils.py:428-459.  One read-back: sum(P) for the `P.mean() <= pathol_tol` test."""
        shape = tuple(deform_dict["grid"][0].shape)

        def zeros():
            z = {"pathology": torch.zeros(shape, device=self.device)[None],
                 "pathology_prob": torch.zeros(shape, device=self.device)[None]}
            self._psum = (z["pathology"], 0.0)
            return z

        if source is None:
            return zeros()
        pmax = None
        if isinstance(source, str) and source == "random_shape":
            percentile = np.random.uniform(self.shape_gen_args.mask_percentile_min,
                                           self.shape_gen_args.mask_percentile_max)
            Pdef, pmax = generate_shape_3d_dev(shape, self.shape_gen_args.perlin_res, percentile, self.device)
        else:
            pending, self._batch = self._batch, None              # needed now: not part of the collected gather
            try:
                Pdef = self.read_and_deform(source, deform_dict)
            finally:
                self._batch = pending
        if augment:
            Pdef = GU.augment_pathology(Pdef, self.adv_pde, self.t, self.shape_gen_args, self.device)
            pmax = None
        P, psum = GU.binarize_dev(Pdef, thres, pmax)
        total = float(psum.item())
        if total / P.numel() <= self.shape_gen_args.pathol_tol:
            return zeros()
        self._psum = (P, total)
        return {"pathology": P[None], "pathology_prob": Pdef[None]}

    def _pathology_sum(self, target):
        """float(target['pathology'].sum()) for the tests of datasets.py:322,387; the host already knows it for the tensor
        read_and_deform_pathology / generate_sample left there (no kernel, no sync), anything else is summed."""
        P = target.get("pathology") if hasattr(target, "get") else None
        if not isinstance(P, torch.Tensor):
            return 0.0
        held = self._psum[0]
        if held is not None and held.data_ptr() == P.data_ptr() and held.numel() == P.numel():
            return self._psum[1]
        return float(P.sum().item())

    def read_and_deform_target(self, case, task_name, input_mode, setups, deform_dict):
        """datasets.py:592-633."""
        if task_name == "pathology":
            src, augment, thres = None, False, 0.1
            if self.pathology_type is None and setups["pathol_mode"]:
                if setups["pathol_random_shape"] or "pathology_prob" not in case:
                    src, augment, thres = "random_shape", False, self.shape_gen_args.pathol_thres
                else:
                    src = case["pathology_prob"]
                    augment, thres = self.synth_args.augment_pathology, self.shape_gen_args.pathol_thres
            return self.read_and_deform_pathology(src, setups, deform_dict, augment, thres)
        if task_name not in case:
            return {task_name: 0.}
        v = case[task_name]
        if task_name in ("T1", "T2", "FLAIR"):
            return self.read_and_deform_image(task_name, v, setups, deform_dict)
        if task_name == "CT":
            return {"CT": self.read_and_deform(v, deform_dict, scale=1000, flip=setups["flip"])[None]}
        if task_name == "segmentation":
            return self.read_and_deform_segmentation(v, setups, deform_dict)
        if task_name == "distance":
            return self.read_and_deform_distance(v, setups, deform_dict)
        if task_name == "registration":
            return self.read_and_deform_registration(v, setups, deform_dict)
        if task_name == "bias_field":
            return {"bias_field": self.read_and_deform(v, deform_dict, flip=setups["flip"])[None]}
        return {task_name: 0.}

    # -------------------------------------------------------------- samples (datasets.py:306-412,496-518)
    def encode_pathology(self, I, P, Pprob, pathol_direction=None):
        """datasets.py:496-518.  I_mu = sum(I*P) / sum(P), the mean / spread tables and the direction stay on the device
        (pathol_direction may be a DeviceDirection); P / Pprob are read in their own dtype (fp64 for Perlin shapes: the
        sum I + Pprob * (...) is then formed in fp64 and rounded once, as torch's promotion does)."""
        lib = L.load()
        if pathol_direction is None:
            pathol_direction = random.choice([True, False])
        P, Pprob = torch.squeeze(P), torch.squeeze(Pprob)
        f64 = P.dtype == torch.float64 and Pprob.dtype == torch.float64
        if not f64:
            P, Pprob = P.to(torch.float32), Pprob.to(torch.float32)
        P, Pprob = P.contiguous(), Pprob.contiguous()
        I = I.to(torch.float32).contiguous()
        u_mu = GU.draws.rand(10000)
        u_sig = GU.draws.rand(10000)
        u4 = (C.c_float * 4)(float(u_mu[0]), float(u_mu[1]), float(u_sig[0]), float(u_sig[1]))
        rn = GU.draws.randn(P.shape, self.device)
        out = torch.empty_like(I)
        dotsum = torch.empty(2, dtype=torch.float64, device=self.device)
        ws = GU.workspace(self.device, lib.bfm_pathology_encode_workspace())
        if isinstance(pathol_direction, DeviceDirection):
            direction, stats = -1, pathol_direction.stats
        else:
            direction, stats = int(bool(pathol_direction)), None
        L.check(lib.bfm_pathology_encode_dev(L.ptr(I), L.ptr(P), L.ptr(Pprob), int(f64), L.ptr(rn), u4, direction,
                                             L.ptr(stats), I.numel(), L.ptr(out), L.ptr(dotsum), L.ptr(ws), ws.numel(),
                                             L.stream_ptr()), "pathology_encode_dev")
        return out

    def augment_sample(self, name, I_def, setups, deform_dict, res, target, pathol_direction=None, input_mode="synth"):
        """datasets.py:306-352."""
        lib = L.load()
        sample = {}
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        if not isinstance(I_def, torch.Tensor):
            I_def = self._crop(I_def, deform_dict["grid"])
            I_def = GU.fast_3D_interp_torch(I_def, xx2, yy2, zz2, "linear")
        if input_mode == "CT":
            I_def = GU.ew_unary(L.EW_CLAMP, I_def, 0., 80.)
        if "pathology" in target and isinstance(target["pathology"], torch.Tensor) and self._pathology_sum(target) > 0:
            I_def = self.encode_pathology(I_def, target["pathology"], target["pathology_prob"], pathol_direction)
        else:
            target["pathology"] = 0.
            target["pathology_prob"] = 0.
        aux = {}
        steps = self.augmentation_steps["synth"] if input_mode == "synth" else self.augmentation_steps["real"]
        for fn in steps:
            I_def, aux = GU.augmentation_funcs[fn](I=I_def, aux_dict=aux, cfg=self.gen_args.generator,
                                                   input_mode=input_mode, setups=setups, size=self.size, res=res,
                                                   device=self.device)
        if getattr(self.synth_args, "bspline_zooming", False):          # Generator/datasets.py:337-338
            I_def = IP.resize(I_def, shape=list(self.size), anchor="edge", interpolation=3, bound="dct2", prefilter=True)
        else:
            I_def = GU.myzoom_torch(I_def, 1 / aux["factors"]) if "factors" in aux else I_def
        # maxi = max(I_def); I_final = I_def / maxi; residual = high_res / maxi - I_final: one reduction, one kernel, the
        # maximum stays on the device (round 3: a read-back plus four launches)
        I_def = I_def.to(torch.float32).contiguous()
        maxi = GU.reduce_dev(1, I_def)
        flip = bool(setups["flip"])
        want_res = "super_resolution" in self.tasks and "high_res" in aux
        hr = aux["high_res"].to(torch.float32).contiguous() if want_res else None
        I_final = torch.empty_like(I_def)
        resid = torch.empty_like(I_def) if want_res else None
        sx, sy, sz = I_def.shape
        L.check(lib.bfm_sample_finalize(L.ptr(I_def), L.ptr(hr), sx, sy, sz, L.ptr(maxi), int(flip), L.ptr(I_final),
                                        L.ptr(resid), L.stream_ptr()), "sample_finalize")
        if want_res:
            sample["high_res_residual"] = resid[None]
        sample["input"] = I_final[None]
        if "bias_field" in self.tasks and input_mode != "CT" and "BFlog" in aux:
            sample["bias_field_log"] = (self._flip0(aux["BFlog"]) if flip else aux["BFlog"])[None]
        return sample

    def get_pathology_direction(self, input_mode, pathol_direction=None):
        """datasets.py:414-427: True = T2/FLAIR-like (lesion brighter), False = T1/CT-like."""
        if pathol_direction is not None:
            return pathol_direction
        if input_mode in ("T1", "CT"):
            return False
        if input_mode in ("T2", "FLAIR"):
            return True
        return random.choice([True, False])

    def generate_sample(self, name, G, setups, deform_dict, res, target):
        """datasets.py:355-412."""
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        lib = L.load()
        mus, sigmas = self.get_contrast(setups["photo_mode"])
        Gc = self._crop(G, deform_dict["grid"], cache=deform_dict.setdefault("_crops", {})).contiguous()
        rn = GU.draws.randn(Gc.shape, self.device)
        SYN = torch.empty_like(Gc)
        L.check(lib.bfm_label_gauss(L.ptr(Gc), L.ptr(mus), L.ptr(sigmas), L.ptr(rn), Gc.numel(), 256, L.ptr(SYN),
                                    L.stream_ptr()), "label_gauss")
        SYN0 = SYN                                            # crop space, before the deformation
        SYN = GU.fast_3D_interp_torch(SYN, xx2, yy2, zz2)
        if np.random.rand() < getattr(self.gen_args, "mix_synth_prob", 0.):        # random linear combination, :377-386
            have = getattr(self, "modalities", None) or {}
            v = GU.draws.rand(4).clone()
            v[2] = 0 if "T2" not in have else v[2]
            v[3] = 0 if "FLAIR" not in have else v[3]
            v /= torch.sum(v)
            SYN = GU.ew_binary(L.EW_AXPY, GU.ew_unary(L.EW_AFFINE, SYN, float(v[0]), 0.0),
                               target["T1"][0].to(torch.float32), float(v[1]))
            if "T2" in have:
                SYN = GU.ew_binary(L.EW_AXPY, SYN, target["T2"][0].to(torch.float32), float(v[2]))
            if "FLAIR" in have:
                SYN = GU.ew_binary(L.EW_AXPY, SYN, target["FLAIR"][0].to(torch.float32), float(v[3]))
        if "pathology" in target and isinstance(target["pathology"], torch.Tensor) and self._pathology_sum(target) > 0:
            # :388-404.  The reference masks the DEFORMED image with the label crop (SYN_cerebral[Gr == 0] = 0) and deforms
            # the result a second time; torch accepts that only when the crop box has the generator's size (IndexError
            # otherwise: the reference cannot draw pathology on a synthetic input from a larger volume).  Same sizes: the
            # reference's arithmetic, pinned by gen_chain.npz case B.  Otherwise (an extension, not reference behaviour):
            # the mask and the white / grey matter means are taken in crop space, where labels and image line up, and
            # deformed once.
            same = tuple(Gc.shape) == tuple(SYN.shape)
            src = SYN.contiguous() if same else SYN0.contiguous()
            cer = torch.empty_like(src)
            stats = torch.empty(4, dtype=torch.float64, device=self.device)
            part = GU.workspace(self.device, 8 * 4 * L.CLASS_STATS_BLOCKS)
            L.check(lib.bfm_label_class_stats(L.ptr(Gc), L.ptr(src), src.numel(), L.ptr(cer), L.ptr(stats), L.ptr(part),
                                              L.stream_ptr()), "label_class_stats")
            cer = GU.fast_3D_interp_torch(cer, xx2, yy2, zz2)
            # target['pathology'][SYN_cerebral == 0] = 0 (and _prob), in place and in their own dtype like the reference's
            # indexed assignment; the masked sum comes back with the same launch pair and is the sample's one read-back
            P, Pprob = target["pathology"], target["pathology_prob"]
            if P.dtype != Pprob.dtype or P.dtype not in (torch.float32, torch.float64) or not P.is_contiguous() \
                    or not Pprob.is_contiguous():
                dt = torch.float64 if P.dtype == torch.float64 else torch.float32
                P, Pprob = P.to(dt).contiguous(), Pprob.to(dt).contiguous()
                target["pathology"], target["pathology_prob"] = P, Pprob
            psum = torch.empty(1, dtype=torch.float64, device=self.device)
            ws = GU.workspace(self.device, lib.bfm_shape_workspace())
            L.check(lib.bfm_pathology_mask(L.ptr(P), L.ptr(Pprob), int(P.dtype == torch.float64), L.ptr(cer), cer.numel(),
                                           L.ptr(psum), L.ptr(ws), ws.numel(), L.stream_ptr()), "pathology_mask")
            self._psum = (P, float(psum.item()))
            pathol_direction = self.get_pathology_direction("synth", DeviceDirection(stats))
        else:
            pathol_direction = None
            target["pathology"] = 0.
            target["pathology_prob"] = 0.
        SYN = GU.ew_unary(L.EW_CLAMP_MIN, SYN, 0.0)
        return target["pathology"], target["pathology_prob"], \
            self.augment_sample(name, SYN, setups, deform_dict, res, target, pathol_direction=pathol_direction)

    def update_gen_args(self, new_args):
        if new_args is None:
            return
        for k, v in vars(new_args).items():
            vars(self.gen_args.generator)[k] = v

    def _read_input(self, idx):
        case = self.cases[idx]
        # Generator/datasets.py:541-557 (get_info): the item's modalities = what exists for this case; generate_sample's
        # T2 / FLAIR mixing asks this list (:381-387), not the case dictionary it happens to be handed
        self.modalities = {k: case[k] for k in case if k in ("T1", "T2", "FLAIR", "CT", "Gen", "segmentation",
                                                              "T1_DM", "T2_DM", "FLAIR_DM", "CT_DM")}
        prob = np.random.rand()
        probs = self.input_prob
        mode = "synth"
        if probs is not None:
            p = vars(probs) if not isinstance(probs, dict) else probs
            p = p.get(case.get("dataset", "synth"), p) if isinstance(p, dict) else p
            p = vars(p) if hasattr(p, "__dict__") else p
            for m in ("T1", "T2", "FLAIR", "CT"):
                if isinstance(p, dict) and prob < p.get(m, 0.) and m in case:
                    mode = m
                    break
        img = _vol(case[mode if mode != "synth" else "Gen"])
        res = np.sqrt(np.sum(abs(img.affine[:-1, :-1]), axis=0))
        return case.get("dataset", "synth"), case.get("name", str(idx)), mode, img, img.affine, res, case

    def _targets(self, case, input_mode, setups, deform_dict, default):
        """datasets.py:592-633,660-668 / 722-730.  The float targets of the item share one coordinate field: their
        read_and_deform calls are collected and run as ONE gather over the resident volumes (coordinates and weights read
        and computed once for T1, the distance maps, the registration maps ...)."""
        target = defaultdict(default)
        target["name"] = case.get("name", "")
        self._psum = (None, 0.0)
        self._batch = []
        try:
            for t in ("T1", "T2", "FLAIR"):
                target.update(self.read_and_deform_target(case, t, input_mode, setups, deform_dict))
            for t in self.tasks:
                if t not in ("T1", "T2", "FLAIR", "surface", "super_resolution", "contrastive", "age"):
                    target.update(self.read_and_deform_target(case, t, input_mode, setups, deform_dict))
            jobs, self._batch = self._batch, None
            if jobs:
                self._run_gather(jobs, deform_dict, setups["flip"])
        finally:
            self._batch = None
        return target

    def __getitem__(self, idx):
        """datasets.py:638-681."""
        dataset_name, case_name, input_mode, img, aff, res, case = self._read_input(idx)
        setups = self.get_setup_params()
        deform_dict = self.generate_deformation(setups, img.shape)
        target = self._targets(case, input_mode, setups, deform_dict, lambda: None)
        if input_mode == "synth":
            self.update_gen_args(self.synth_image_args)
            target["pathology"], target["pathology_prob"], sample = \
                self.generate_sample(case_name, img, setups, deform_dict, res, target)
        else:
            self.update_gen_args(self.real_image_args)
            sample = self.augment_sample(case_name, img, setups, deform_dict, res, target,
                                         pathol_direction=self.get_pathology_direction(input_mode), input_mode=input_mode)
        if setups["flip"] and isinstance(target["pathology"], torch.Tensor):
            target["pathology"] = self._flip_axis1(target["pathology"])
            target["pathology_prob"] = self._flip_axis1(target["pathology_prob"])
        return self.datasets_num, dataset_name, input_mode, target, sample


class BrainIDGen(BaseGen):
    """BrainIDGen, datasets.py:686-757: one deformation, `all_samples` augmented inputs (mild first)."""

    def __init__(self, gen_args, device="cuda", cases=None):
        super().__init__(gen_args, device, cases)
        self.all_samples = gen_args.generator.all_samples
        self.mild_samples = gen_args.generator.mild_samples
        self.mild_generator_args = getattr(gen_args, "mild_generator", None)
        self.severe_generator_args = getattr(gen_args, "severe_generator", None)

    def __getitem__(self, idx):
        dataset_name, case_name, input_mode, img, aff, res, case = self._read_input(idx)
        setups = self.get_setup_params()
        deform_dict = self.generate_deformation(setups, img.shape)
        target = self._targets(case, input_mode, setups, deform_dict, lambda: 1.)
        samples = []
        for i in range(self.all_samples):
            self.update_gen_args(self.mild_generator_args if i < self.mild_samples else self.severe_generator_args)
            if input_mode == "synth":
                self.update_gen_args(self.synth_image_args)
                target["pathology"], target["pathology_prob"], sample = \
                    self.generate_sample(case_name, img, setups, deform_dict, res, target)
            else:
                self.update_gen_args(self.real_image_args)
                sample = self.augment_sample(case_name, img, setups, deform_dict, res, target,
                                             pathol_direction=self.get_pathology_direction(input_mode),
                                             input_mode=input_mode)
            samples.append(sample)
        if setups["flip"] and isinstance(target["pathology"], torch.Tensor):
            target["pathology"] = self._flip_axis1(target["pathology"])
            target["pathology_prob"] = self._flip_axis1(target["pathology_prob"])
        return self.datasets_num, dataset_name, input_mode, target, samples


dataset_options = {"default": BaseGen, "brain_id": BrainIDGen}


def build_datasets(gen_args, device, cases=None):
    """Generator/__init__.py:18-21: {'all': Dataset}."""
    return {"all": dataset_options[getattr(gen_args, "dataset_option", "brain_id")](gen_args, device, cases)}
