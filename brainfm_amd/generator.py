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
import random
from collections import defaultdict

import numpy as np
import torch

from . import _lib as L
from . import generator_utils as GU
from . import interpol as IP
from .engine import LABELS_FULL, LABELS_LEFT
from .shapeid import AdvDiffPDE, generate_shape_3d

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

    def random_nonlinear_transform(self, photo_mode, spac):
        s = self.synth_args
        nonlin_scale = s.nonlin_scale_min + np.random.rand(1) * (s.nonlin_scale_max - s.nonlin_scale_min)
        size_F_small = np.round(nonlin_scale * np.array(self.size)).astype(int).tolist()
        if photo_mode:
            size_F_small[1] = np.round(self.size[1] / spac).astype(int)
        nonlin_std = s.nonlin_std_max * np.random.rand()
        Fsmall = GU.ew_unary(L.EW_AFFINE, GU.draws.randn([*size_F_small, 3], self.device),
                             float(np.float32(nonlin_std)), 0.0)
        F = GU.myzoom_torch(Fsmall, np.array(self.size) / size_F_small)
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

    def generate_deformation(self, setups, shp):
        scaling_factor_distances, A, c2 = self.random_affine_transform(shp)
        if self.synth_args.nonlinear_transform:
            F, Fneg = self.random_nonlinear_transform(setups["photo_mode"], setups["spac"])
        else:
            F, Fneg = None, None
        xx2, yy2, zz2, x1, y1, z1, x2, y2, z2 = self.deform_grid(shp, A, c2, F)
        return {"scaling_factor_distances": scaling_factor_distances, "A": A, "c2": c2, "F": F, "Fneg": Fneg,
                "grid": [xx2, yy2, zz2, x1, y1, z1, x2, y2, z2]}

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
    def _crop(self, vol, grid, dtype=torch.float32):
        [_, _, _, x1, y1, z1, x2, y2, z2] = grid
        v = _vol(vol)
        # Crop first and convert the crop straight to the target type.  The reference goes array -> float64 (whole
        # volume, get_fdata) -> crop -> float / int -> tensor(dtype): for float32, float64 and integer sources that chain
        # rounds once, exactly like the direct conversion, so the values are the same; it cost 30 ms x 17 crops per item.
        if isinstance(v, ArrayVolume):
            src = v._d[x1:x2, y1:y2, z1:z2]
        elif hasattr(v, "dataobj"):
            src = np.asarray(v.dataobj[x1:x2, y1:y2, z1:z2])
        else:
            src = v.get_fdata()[x1:x2, y1:y2, z1:z2]
        np_t = np.float32 if dtype == torch.float32 else np.int64
        if not (np.issubdtype(src.dtype, np.floating) or np.issubdtype(src.dtype, np.integer)):
            src = src.astype(np.float64)
        arr = np.ascontiguousarray(src, dtype=np_t)
        return torch.squeeze(torch.from_numpy(arr).to(device=self.device, dtype=dtype))

    def read_and_deform(self, vol, deform_dict, default_max=False, mean=0., scale=1.):
        """read_and_deform, Generator/utils.py:296-322."""
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        I = torch.nan_to_num(self._crop(vol, deform_dict["grid"]))
        if mean != 0. or scale != 1.:
            I = GU.ew_unary(L.EW_SUB_DIV, I, mean, scale)
        dv = GU.tensor_max(I) if default_max else 0.
        return GU.fast_3D_interp_torch(I, xx2, yy2, zz2, "linear", dv)

    def _flip0(self, t):
        return torch.flip(t, [0])

    def read_and_deform_image(self, task, vol, setups, deform_dict):
        I = self.read_and_deform(vol, deform_dict)
        I = GU.ew_unary(L.EW_SUB_DIV, I, GU.tensor_min(I), 1.0)
        I = GU.ew_unary(L.EW_DIV, I, GU.tensor_max(I))
        return {task: (self._flip0(I) if setups["flip"] else I)[None]}

    def read_and_deform_segmentation(self, vol, setups, deform_dict):
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        S = self._crop(vol, deform_dict["grid"], torch.int32)
        Sdef = GU.fast_3D_interp_torch(S, xx2, yy2, zz2, "nearest").contiguous()
        n = Sdef.numel()
        out = torch.empty(tuple(Sdef.shape) + (self.n_labels,), dtype=torch.float32, device=self.device)
        L.check(L.load().bfm_onehot_lut(L.ptr(Sdef), L.ptr(self.lut), self.lut.numel(), self.n_labels, n, L.ptr(out),
                                        L.stream_ptr()), "onehot_lut")
        if setups["flip"]:
            out = torch.flip(out, [0])[:, :, :, torch.as_tensor(self.vflip, device=self.device)]
        return {"segmentation": out.permute([3, 0, 1, 2])}

    def read_and_deform_distance(self, vols, setups, deform_dict):
        maps = [self.read_and_deform(v, deform_dict, default_max=True, mean=128., scale=20) for v in vols]
        if len(maps) == 4 and setups["flip"]:
            lp, lw, rp, rw = maps
            maps = [self._flip0(rp), self._flip0(rw), self._flip0(lp), self._flip0(lw)]
        m = float(self.gen_args.max_surf_distance)
        sf = float(deform_dict["scaling_factor_distances"])
        maps = [GU.ew_unary(L.EW_CLAMP, GU.ew_unary(L.EW_DIV, t, sf), -m, m) for t in maps]
        return {"distance": torch.stack(maps, dim=0)}

    def read_and_deform_registration(self, vols, setups, deform_dict):
        r = [self.read_and_deform(v, deform_dict, scale=10000) for v in vols]
        if setups["flip"]:
            r = [GU.ew_unary(L.EW_AFFINE, self._flip0(r[0]).contiguous(), -1.0, 0.0), self._flip0(r[1]), self._flip0(r[2])]
        return {"registration": torch.stack(r, dim=0)}

    def read_and_deform_pathology(self, source, setups, deform_dict, augment, thres):
        """Generator/utils.py:428-459."""
        shape = tuple(deform_dict["grid"][0].shape)
        zeros = {"pathology": torch.zeros(shape, device=self.device)[None],
                 "pathology_prob": torch.zeros(shape, device=self.device)[None]}
        if source is None:
            return zeros
        if isinstance(source, str) and source == "random_shape":
            percentile = np.random.uniform(self.shape_gen_args.mask_percentile_min,
                                           self.shape_gen_args.mask_percentile_max)
            _, Pdef = generate_shape_3d(shape, self.shape_gen_args.perlin_res, percentile, self.device)
        else:
            Pdef = self.read_and_deform(source, deform_dict)
        if augment:
            Pdef = GU.augment_pathology(Pdef, self.adv_pde, self.t, self.shape_gen_args, self.device)
        P = GU.binarize(Pdef, thres)
        mean = (GU.tensor_sum(P) if P.dtype == torch.float32 else float(P.sum().item())) / P.numel()
        if mean <= self.shape_gen_args.pathol_tol:
            return zeros
        return {"pathology": P[None], "pathology_prob": Pdef[None]}

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
            I = self.read_and_deform(v, deform_dict, scale=1000)
            return {"CT": (self._flip0(I) if setups["flip"] else I)[None]}
        if task_name == "segmentation":
            return self.read_and_deform_segmentation(v, setups, deform_dict)
        if task_name == "distance":
            return self.read_and_deform_distance(v, setups, deform_dict)
        if task_name == "registration":
            return self.read_and_deform_registration(v, setups, deform_dict)
        if task_name == "bias_field":
            I = self.read_and_deform(v, deform_dict)
            return {"bias_field": (self._flip0(I) if setups["flip"] else I)[None]}
        return {task_name: 0.}

    # -------------------------------------------------------------- samples (datasets.py:306-412,496-518)
    def encode_pathology(self, I, P, Pprob, pathol_direction=None):
        if pathol_direction is None:
            pathol_direction = random.choice([True, False])
        P, Pprob = torch.squeeze(P).to(torch.float32).contiguous(), torch.squeeze(Pprob).to(torch.float32).contiguous()
        I = I.contiguous()
        I_mu = np.float32(GU.tensor_dot(I, P) / GU.tensor_sum(P))
        pth_mus = 3 * I_mu / 4 + I_mu / 4 * GU.draws.rand(10000).numpy()
        pth_mus = pth_mus if pathol_direction else -pth_mus
        pth_sigmas = I_mu / 4 * GU.draws.rand(10000).numpy()
        rn = GU.draws.randn(P.shape, self.device)
        out = torch.empty_like(I)
        L.check(L.load().bfm_pathology_encode(L.ptr(I), L.ptr(P), L.ptr(Pprob), L.ptr(rn), float(pth_mus[0]),
                                              float(pth_mus[1]), float(pth_sigmas[0]), float(pth_sigmas[1]), I.numel(),
                                              L.ptr(out), L.stream_ptr()), "pathology_encode")
        return out

    def augment_sample(self, name, I_def, setups, deform_dict, res, target, pathol_direction=None, input_mode="synth"):
        sample = {}
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        if not isinstance(I_def, torch.Tensor):
            I_def = self._crop(I_def, deform_dict["grid"])
            I_def = GU.fast_3D_interp_torch(I_def, xx2, yy2, zz2, "linear")
        if input_mode == "CT":
            I_def = GU.ew_unary(L.EW_CLAMP, I_def, 0., 80.)
        if "pathology" in target and isinstance(target["pathology"], torch.Tensor) and \
                float(target["pathology"].sum().item()) > 0:
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
        maxi = GU.tensor_max(I_def)
        I_final = GU.ew_unary(L.EW_DIV, I_def, maxi)
        fl = (lambda t: torch.flip(t, [0])) if setups["flip"] else (lambda t: t)
        if "super_resolution" in self.tasks and "high_res" in aux:
            hr = GU.ew_unary(L.EW_DIV, aux["high_res"], maxi)
            sample["high_res_residual"] = fl(GU.ew_binary(L.EW_AXPY, hr, I_final, -1.0))[None]
        sample["input"] = fl(I_final)[None]
        if "bias_field" in self.tasks and input_mode != "CT" and "BFlog" in aux:
            sample["bias_field_log"] = fl(aux["BFlog"])[None]
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

    def generate_sample(self, name, G, setups, deform_dict, res, target, case=None):
        """datasets.py:355-412."""
        [xx2, yy2, zz2] = deform_dict["grid"][:3]
        lib = L.load()
        mus, sigmas = self.get_contrast(setups["photo_mode"])
        Gc = self._crop(G, deform_dict["grid"]).contiguous()
        rn = GU.draws.randn(Gc.shape, self.device)
        SYN = torch.empty_like(Gc)
        L.check(lib.bfm_label_gauss(L.ptr(Gc), L.ptr(mus), L.ptr(sigmas), L.ptr(rn), Gc.numel(), 256, L.ptr(SYN),
                                    L.stream_ptr()), "label_gauss")
        SYN0 = SYN                                            # crop space, before the deformation
        SYN = GU.fast_3D_interp_torch(SYN, xx2, yy2, zz2)
        if np.random.rand() < getattr(self.gen_args, "mix_synth_prob", 0.):        # random linear combination, :377-386
            have = getattr(self, "modalities", None)
            if have is None:
                have = case if case is not None else {}
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
        if "pathology" in target and isinstance(target["pathology"], torch.Tensor) and \
                float(target["pathology"].sum().item()) > 0:
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
            part = torch.empty(4 * L.CLASS_STATS_BLOCKS, dtype=torch.float64, device=self.device)
            L.check(lib.bfm_label_class_stats(L.ptr(Gc), L.ptr(src), src.numel(), L.ptr(cer), L.ptr(stats), L.ptr(part),
                                              L.stream_ptr()), "label_class_stats")
            cer = GU.fast_3D_interp_torch(cer, xx2, yy2, zz2)[None]
            st = stats.cpu().tolist()
            wm_mean = st[0] / st[1] if st[1] else float("nan")
            gm_mean = st[2] / st[3] if st[3] else float("nan")
            for k in ("pathology", "pathology_prob"):
                t = target[k]
                target[k] = GU.ew_binary(L.EW_ZERO_WHERE_ZERO, t.to(torch.float32).contiguous(), cer).to(t.dtype)
            pathol_direction = self.get_pathology_direction("synth", gm_mean > wm_mean)
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
        target = defaultdict(default)
        target["name"] = case.get("name", "")
        for t in ("T1", "T2", "FLAIR"):
            target.update(self.read_and_deform_target(case, t, input_mode, setups, deform_dict))
        for t in self.tasks:
            if t not in ("T1", "T2", "FLAIR", "surface", "super_resolution", "contrastive", "age"):
                target.update(self.read_and_deform_target(case, t, input_mode, setups, deform_dict))
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
                self.generate_sample(case_name, img, setups, deform_dict, res, target, case)
        else:
            self.update_gen_args(self.real_image_args)
            sample = self.augment_sample(case_name, img, setups, deform_dict, res, target,
                                         pathol_direction=self.get_pathology_direction(input_mode), input_mode=input_mode)
        if setups["flip"] and isinstance(target["pathology"], torch.Tensor):
            target["pathology"] = torch.flip(target["pathology"], [1])
            target["pathology_prob"] = torch.flip(target["pathology_prob"], [1])
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
                    self.generate_sample(case_name, img, setups, deform_dict, res, target, case)
            else:
                self.update_gen_args(self.real_image_args)
                sample = self.augment_sample(case_name, img, setups, deform_dict, res, target,
                                             pathol_direction=self.get_pathology_direction(input_mode),
                                             input_mode=input_mode)
            samples.append(sample)
        if setups["flip"] and isinstance(target["pathology"], torch.Tensor):
            target["pathology"] = torch.flip(target["pathology"], [1])
            target["pathology_prob"] = torch.flip(target["pathology_prob"], [1])
        return self.datasets_num, dataset_name, input_mode, target, samples


dataset_options = {"default": BaseGen, "brain_id": BrainIDGen}


def build_datasets(gen_args, device, cases=None):
    """Generator/__init__.py:18-21: {'all': Dataset}."""
    return {"all": dataset_options[getattr(gen_args, "dataset_option", "brain_id")](gen_args, device, cases)}
