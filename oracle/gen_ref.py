"""CPU restatement (NumPy) of the reference's generator chains -- TEST INFRASTRUCTURE, not product code:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import anything under oracle/.

  BaseGen.get_setup_params / random_affine_transform / random_nonlinear_transform / generate_deformation
                                                                   Generator/datasets.py:187-303, 466-493
  read_and_deform* (image, segmentation, distance, registration, pathology from a random Perlin shape)
                                                                   Generator/utils.py:296-459
  get_contrast, generate_sample, encode_pathology, augment_sample   Generator/datasets.py:306-464, 496-518
  BaseGen.__getitem__ / BrainIDGen.__getitem__                      Generator/datasets.py:638-681, 700-757
  augmentation functions                                            Generator/utils.py:568-638

Pinned by tests/golden/gen_chain.npz, which tests/golden/make_golden_gen.py produced by running the reference's own
__getitem__ on in-memory cases.  Randomness: NumPy's and `random`'s global streams are consumed in the reference's call
order (the caller seeds them with the fixture's seed); every torch draw of the reference is replayed from the fixture
(`draws`: list of (kind, array) in call order).
"""
from collections import defaultdict

import numpy as np

from . import synth_ref as S

F32 = np.float32
LABELS_FULL = [0, 11, 12, 13, 16, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46,
               1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 17, 47, 49, 51, 53, 55,
               18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 48, 50, 52, 54, 56]
N_NEUTRAL = 20


class Draws:
    def __init__(self, seq):
        self.seq, self.pos = list(seq), 0

    def take(self, kind, shape):
        k, a = self.seq[self.pos]
        assert k == kind and tuple(a.shape) == tuple(shape), (self.pos, kind, tuple(shape), k, a.shape)
        self.pos += 1
        return np.asarray(a, dtype=F32)


def make_affine_matrix(rot, sh, s):
    """Generator/utils.py:102-116, restated: A = SHx SHy SHz Rx Ry Rz, then row r times s[r]."""
    cx, cy, cz = (np.cos(rot[k]) for k in range(3))
    sx, sy, sz = (np.sin(rot[k]) for k in range(3))
    factors = np.zeros((6, 3, 3))
    factors[:, [0, 1, 2], [0, 1, 2]] = 1.0
    factors[0, 1, 0], factors[0, 2, 0] = sh[1], sh[2]              # shear carried by column x
    factors[1, 0, 1], factors[1, 2, 1] = sh[0], sh[2]              # ... column y
    factors[2, 0, 2], factors[2, 1, 2] = sh[0], sh[1]              # ... column z
    factors[3, 1:, 1:] = [[cx, -sx], [sx, cx]]
    factors[4, ::2, ::2] = [[cy, sy], [-sy, cy]]
    factors[5, :2, :2] = [[cz, -sz], [sz, cz]]
    A = factors[0]
    for M in factors[1:]:
        A = A @ M
    return np.stack([A[r] * s[r] for r in range(3)])


def resolution_sampler(low_res_only=False):
    """Generator/utils.py:34-58."""
    r = (np.random.rand() * 0.5) + 0.5 if low_res_only else np.random.rand()
    if r < 0.25:
        resolution, thickness = np.array([1.0, 1.0, 1.0]), np.array([1.0, 1.0, 1.0])
    elif r < 0.5:
        resolution, thickness = np.array([1.0, 1.0, 1.0]), np.array([1.0, 1.0, 1.0])
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


class GenOracle:
    """cfg: the generator configuration as nested dicts (fixture 'cfg_json'); case: {'Gen','T1','segmentation',
    'distance': [4], 'registration': [3]} arrays."""

    def __init__(self, cfg, case, draws, t1_prob=0.0, brain_id=True):
        self.cfg = cfg
        self.g = dict(cfg["generator"])                     # update_gen_args edits it
        self.case = case
        self.draws = Draws(draws)
        self.t1_prob = t1_prob
        self.brain_id = brain_id
        self.size = list(self.g["size"])
        self.tasks = [k for k, v in cfg["task"].items() if v]
        if "bias_field" in self.tasks and "segmentation" not in self.tasks:
            self.tasks.append("segmentation")
        n = len(LABELS_FULL)
        self.lut = np.zeros(10000, dtype=np.int64)
        for l in range(n):
            self.lut[LABELS_FULL[l]] = l
        nlat = (n - N_NEUTRAL) // 2
        self.vflip = np.concatenate([np.arange(N_NEUTRAL), np.arange(N_NEUTRAL + nlat, n), np.arange(N_NEUTRAL, N_NEUTRAL + nlat)])
        self.n_labels = n

    # ------------------------------------------------------------------ datasets.py:466-493
    def get_setup_params(self):
        g = self.g
        photo_mode = np.random.rand() < g["photo_prob"]
        pathol_mode = np.random.rand() < g["pathology_prob"]
        pathol_random_shape = np.random.rand() < g["random_shape_prob"]
        spac = 2.5 + 10 * np.random.rand() if photo_mode else None
        flip = np.random.randn() < g["flip_prob"]
        if photo_mode:
            resolution, thickness = np.array([1.0, spac, 1.0]), np.array([1.0, 0.1, 1.0])
        else:
            resolution, thickness = resolution_sampler(g["low_res_only"])
        return dict(resolution=resolution, thickness=thickness, photo_mode=photo_mode, pathol_mode=pathol_mode,
                    pathol_random_shape=pathol_random_shape, spac=spac, flip=flip)

    # ------------------------------------------------------------------ datasets.py:187-303
    def generate_deformation(self, setups, shp):
        g = self.g
        rotations = (2 * g["max_rotation"] * np.random.rand(3) - g["max_rotation"]) / 180.0 * np.pi
        shears = (2 * g["max_shear"] * np.random.rand(3) - g["max_shear"])
        scalings = 1 + (2 * g["max_scaling"] * np.random.rand(3) - g["max_scaling"])
        sfd = np.prod(scalings) ** .33333333333
        A = make_affine_matrix(rotations, shears, scalings).astype(F32)
        c2 = ((np.array(shp[0:3]) - 1) / 2).astype(F32)
        F = None
        if g["nonlinear_transform"]:
            nonlin_scale = g["nonlin_scale_min"] + np.random.rand(1) * (g["nonlin_scale_max"] - g["nonlin_scale_min"])
            size_F_small = np.round(nonlin_scale * np.array(self.size)).astype(int).tolist()
            if setups["photo_mode"]:
                size_F_small[1] = np.round(self.size[1] / setups["spac"]).astype(int)
            nonlin_std = g["nonlin_std_max"] * np.random.rand()
            Fsmall = (F32(nonlin_std) * self.draws.take("randn", size_F_small + [3])).astype(F32)
            F = S.myzoom(Fsmall, np.array(self.size) / size_F_small)
            if setups["photo_mode"]:
                F[:, :, :, 1] = 0
        xx, yy, zz, lo, hi = S.deform_grid(self.size, shp, A, c2, F)
        return dict(sfd=sfd, grid=(xx, yy, zz), lo=lo, hi=hi)

    def _crop(self, vol, d):
        (x1, y1, z1), (x2, y2, z2) = d["lo"], d["hi"]
        return np.asarray(vol)[x1:x2, y1:y2, z1:z2]

    # ------------------------------------------------------------------ Generator/utils.py:296-459
    def read_and_deform(self, vol, d, default_max=False, mean=0., scale=1.):
        I = np.nan_to_num(self._crop(vol, d).astype(np.float64).astype(F32))
        I = ((I - F32(mean)) / F32(scale)).astype(F32)
        dv = float(I.max()) if default_max else 0.
        return S.interp3d_linear(I, *d["grid"], default_value=dv)

    def targets(self, setups, d):
        case, flip = self.case, setups["flip"]
        t = {}
        I = self.read_and_deform(case["T1"], d)
        I = I - I.min()
        I = (I / I.max()).astype(F32)
        t["T1"] = (I[::-1] if flip else I)[None]
        t["T2"], t["FLAIR"] = 0., 0.
        for task in self.tasks:
            if task == "segmentation":
                Sdef = S.interp3d_nearest(self._crop(case["segmentation"], d).astype(np.int32), *d["grid"])
                oh = S.onehot_lut(Sdef, self.lut, self.n_labels)
                if flip:
                    oh = oh[::-1][:, :, :, self.vflip]
                t["segmentation"] = np.ascontiguousarray(oh.transpose(3, 0, 1, 2))
            elif task == "distance":
                lp, lw, rp, rw = [self.read_and_deform(v, d, default_max=True, mean=128., scale=20) for v in case["distance"]]
                if flip:
                    lp, rp = rp[::-1], lp[::-1]
                    lw, rw = rw[::-1], lw[::-1]
                m = F32(self.cfg["max_surf_distance"])
                I = np.stack([lp, lw, rp, rw], 0)
                I = (I / F32(d["sfd"])).astype(F32)          # torch: float tensor /= python float
                t["distance"] = np.clip(I, -m, m)
            elif task == "registration":
                r = [self.read_and_deform(v, d, scale=10000) for v in case["registration"]]
                if flip:
                    r = [-r[0][::-1], r[1][::-1], r[2][::-1]]
                t["registration"] = np.stack(r, 0)
            elif task == "bias_field":
                t["bias_field"] = 0.
            elif task == "pathology":
                t.update(self.pathology_target(setups, d))
        return t

    def pathology_target(self, setups, d):
        """read_and_deform_pathology with file_name in {None, 'random_shape'} (utils.py:428-459)."""
        sg = self.cfg["pathology_shape_generator"]
        shape = tuple(self.size)
        zeros = {"pathology": np.zeros((1,) + shape, F32), "pathology_prob": np.zeros((1,) + shape, F32)}
        if not setups["pathol_mode"]:
            return zeros
        assert setups["pathol_random_shape"], "the file-based source cannot run in the reference (utils.py:442)"
        percentile = np.random.uniform(sg["mask_percentile_min"], sg["mask_percentile_max"])
        res = sg["perlin_res"]
        theta = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)      # perlin3d.py:43-44
        phi = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)
        noise = S.perlin_noise_3d(shape, res, S.perlin_gradients(theta, phi, tileable=(True, False, False)))
        masked, mask, _ = S.percentile_mask(noise, percentile)
        Pdef = masked                                                               # generate_shape_3d returns (mask, prob)
        thres = sg["pathol_thres"] * Pdef.max()
        P = np.where(Pdef >= thres, 1.0, 0.0).astype(Pdef.dtype)
        if P.mean() <= sg["pathol_tol"]:
            return zeros
        return {"pathology": P[None], "pathology_prob": Pdef[None]}

    # ------------------------------------------------------------------ datasets.py:430-464
    def get_contrast(self, photo_mode):
        mus = (25 + 200 * self.draws.take("rand", (256,))).astype(F32)
        sigmas = (5 + 20 * self.draws.take("rand", (256,))).astype(F32)
        assert not (np.random.rand() < self.g["ct_prob"]), "ct_prob > 0 is not part of the fixtures"
        if photo_mode or np.random.rand(1) < 0.5:
            mus[0] = 0
        v = (F32(0.02) * np.arange(50).astype(F32)).astype(F32)
        for a, k in ((100, 1), (150, 2), (200, 3)):
            mus[a:a + 50] = mus[k] * (1 - v) + mus[k + 1] * v
            sigmas[a:a + 50] = np.sqrt(sigmas[k] ** 2 * (1 - v) + sigmas[k + 1] ** 2 * v)
        mus[250], sigmas[250] = mus[4], sigmas[4]
        return mus.astype(F32), sigmas.astype(F32)

    # ------------------------------------------------------------------ datasets.py:496-518
    def encode_pathology(self, I, P, Pprob, direction):
        P, Pprob = np.squeeze(P).astype(F32), np.squeeze(Pprob).astype(F32)
        I_mu = F32((I * P).sum(dtype=np.float64) / P.sum(dtype=np.float64))
        pth_mus = (3 * I_mu / 4 + I_mu / 4 * self.draws.take("rand", (10000,))).astype(F32)
        pth_mus = pth_mus if direction else -pth_mus
        pth_sigmas = (I_mu / 4 * self.draws.take("rand", (10000,))).astype(F32)
        pm = np.round(P).astype(np.int64)
        rn = self.draws.take("randn", pm.shape)
        I = (I + Pprob * (pth_mus[pm] + pth_sigmas[pm] * rn)).astype(F32)
        I[I < 0] = 0
        return I

    # ------------------------------------------------------------------ datasets.py:306-353, utils.py:568-638
    def augment_sample(self, I, setups, d, target, direction, input_mode, res=np.array([1., 1., 1.])):
        g = self.g
        if input_mode != "synth":                               # a volume, not the synthesised image: crop and deform
            I = S.interp3d_linear(self._crop(I, d).astype(np.float64).astype(F32), *d["grid"])
        if isinstance(target.get("pathology"), np.ndarray) and target["pathology"].sum() > 0:
            I = self.encode_pathology(I, target["pathology"], target["pathology_prob"], direction)
        else:
            target["pathology"], target["pathology_prob"] = 0., 0.
        aux = {}
        for fn in self.cfg["augmentation_steps"]["synth" if input_mode == "synth" else "real"]:
            if fn == "gamma":
                gamma = np.exp(g["gamma_std"] * np.random.randn(1)[0])
                I = S.gamma_transform(I, gamma)
            elif fn == "bias_field":
                bf_scale = g["bf_scale_min"] + np.random.rand(1) * (g["bf_scale_max"] - g["bf_scale_min"])
                small = np.round(bf_scale * np.array(self.size)).astype(int).tolist()
                if setups["photo_mode"]:
                    small[1] = np.round(self.size[1] / setups["spac"]).astype(int)
                amp = F32(g["bf_std_min"] + (g["bf_std_max"] - g["bf_std_min"]) * np.random.rand(1))[0]
                BFsmall = (amp * self.draws.take("randn", small)).astype(F32)
                BFlog = S.myzoom(BFsmall, np.array(self.size) / small)
                I = S.apply_bias_field(I, BFlog)
                aux.update(BFlog=BFlog, high_res=I)
            elif fn == "resample":
                stds = (0.85 + 0.3 * np.random.rand()) * np.log(5) / np.pi * setups["thickness"] / res
                stds[setups["thickness"] <= res] = 0.0
                Ib = S.gaussian_blur_3d(I, stds)
                new_size = (np.array(self.size) * res / setups["resolution"]).astype(int)
                factors = np.array(new_size) / np.array(self.size)
                delta = (1.0 - factors) / (2.0 * factors)
                v = [np.arange(delta[a], delta[a] + new_size[a] / factors[a], 1 / factors[a])[:new_size[a]] for a in range(3)]
                II, JJ, KK = np.meshgrid(v[0], v[1], v[2], sparse=False, indexing="ij")
                I = S.interp3d_linear(Ib, II.astype(F32), JJ.astype(F32), KK.astype(F32))
                aux["factors"] = factors
            elif fn == "noise":
                std = F32(g["noise_std_min"] + (g["noise_std_max"] - g["noise_std_min"]) * np.random.rand(1))[0]
                I = S.add_noise(I, std, self.draws.take("randn", I.shape))
        I = S.myzoom(I, 1 / aux["factors"])
        maxi = I.max()
        I_final = (I / maxi).astype(F32)
        fl = (lambda a: a[::-1]) if setups["flip"] else (lambda a: a)
        sample = {}
        if "super_resolution" in self.tasks:
            sample["high_res_residual"] = fl((aux["high_res"] / maxi).astype(F32) - I_final)[None]
        sample["input"] = fl(I_final)[None]
        if "bias_field" in self.tasks and input_mode != "CT":
            sample["bias_field_log"] = fl(aux["BFlog"])[None]
        return sample

    # ------------------------------------------------------------------ datasets.py:355-412
    def generate_sample(self, setups, d, target):
        mus, sigmas = self.get_contrast(setups["photo_mode"])
        G = self._crop(self.case["Gen"], d).astype(np.float64).astype(F32).copy()
        G[G == 77] = 2
        Gr = np.round(G).astype(np.int64)
        SYN = (mus[Gr] + sigmas[Gr] * self.draws.take("randn", Gr.shape)).astype(F32)
        SYN[SYN < 0] = 0
        SYN = S.interp3d_linear(SYN, *d["grid"])
        if np.random.rand() < self.cfg["mix_synth_prob"]:
            v = self.draws.take("rand", (4,)).copy()
            v[2], v[3] = 0, 0                                   # no T2 / FLAIR file for the case
            v = (v / v.sum(dtype=F32)).astype(F32)
            SYN = (v[0] * SYN + v[1] * target["T1"][0]).astype(F32)
        if isinstance(target.get("pathology"), np.ndarray) and target["pathology"].sum() > 0:
            if Gr.shape != SYN.shape:
                raise IndexError("mask shape %s vs image %s (datasets.py:389-391)" % (Gr.shape, SYN.shape))
            cer = SYN.copy()
            cer[Gr == 0] = 0
            cer = S.interp3d_linear(cer, *d["grid"])[None]
            wm = (Gr == 2) | (Gr == 41)
            gm = (Gr != 0) & ~wm
            wm_mean = (SYN * wm).sum(dtype=np.float64) / wm.sum()
            gm_mean = (SYN * gm).sum(dtype=np.float64) / gm.sum()
            target["pathology"] = np.where(cer == 0, 0, target["pathology"]).astype(target["pathology"].dtype)
            target["pathology_prob"] = np.where(cer == 0, 0, target["pathology_prob"]).astype(target["pathology_prob"].dtype)
            direction = bool(gm_mean > wm_mean)
        else:
            direction = None
            target["pathology"], target["pathology_prob"] = 0., 0.
        SYN[SYN < 0] = 0
        return self.augment_sample(SYN, setups, d, target, direction, "synth")

    # ------------------------------------------------------------------ datasets.py:638-681, 700-757
    def getitem(self):
        prob = np.random.rand()
        input_mode = "T1" if prob < self.t1_prob else "synth"
        img = self.case["T1"] if input_mode == "T1" else self.case["Gen"]
        setups = self.get_setup_params()
        d = self.generate_deformation(setups, img.shape)
        target = defaultdict(lambda: 1.) if self.brain_id else defaultdict(lambda: None)
        target.update(self.targets(setups, d))
        samples = []
        n = self.g["all_samples"] if self.brain_id else 1
        for i in range(n):
            if self.brain_id:
                self.g.update(self.cfg["mild_generator"] if i < self.g["mild_samples"] else self.cfg["severe_generator"])
            if input_mode == "synth":
                self.g.update(self.cfg["synth_image_generator"])
                samples.append(self.generate_sample(setups, d, target))
            else:
                self.g.update(self.cfg["real_image_generator"])
                direction = False if input_mode in ("T1", "CT") else True
                samples.append(self.augment_sample(img, setups, d, target, direction, input_mode))
        if setups["flip"] and isinstance(target["pathology"], np.ndarray):
            target["pathology"] = target["pathology"][:, ::-1]
            target["pathology_prob"] = target["pathology_prob"][:, ::-1]
        return input_mode, target, samples, setups
