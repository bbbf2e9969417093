"""Golden vectors for the generator chains made by RUNNING the reference's own
``BrainIDGen.__getitem__`` / ``BaseGen.__getitem__`` (Generator/datasets.py:638-757) end to end:
generate_deformation -> read_and_deform_* -> generate_sample / augment_sample (get_contrast, encode_pathology,
the augmentation chain) on in-memory cases.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_gen.py
Writes tests/golden/gen_chain.npz.  Data only: the case volumes, the configuration values, every random draw
the reference made through torch (recorded in call order; NumPy's and `random`'s streams are reproduced by their
seeds, the mirror keeps the reference's call order on them), and the reference's outputs.

How the reference is driven without files: nibabel is absent from this image, so its stand-in module gets a
`load(path)` that serves in-memory volumes registered under the file names `BaseGen.get_info` derives
(datasets.py:520-560); `BaseGen.__init__` is bypassed only for `prepare_paths` (split files), whose results
(`names`, `datasets`, ...) are set directly.
"""
import copy
import json
import os
import random
import sys
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

R = ref_import.setup()
import torch  # noqa: E402

torch.set_num_threads(4)
import nibabel  # noqa: E402  (the stand-in module)


class MemVol:
    def __init__(self, data, affine=None):
        self._d = np.asarray(data)
        self.shape = self._d.shape
        self.affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)

    def get_fdata(self):
        return self._d.astype(np.float64)


REG = {}
nibabel.load = lambda path: REG[path]


def make_case(shp, seed, with_prob=False):
    rs = np.random.RandomState(seed)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
    c = [(s - 1) / 2. for s in shp]
    ell = (((zz - c[0]) / (0.44 * shp[0])) ** 2 + ((yy - c[1]) / (0.42 * shp[1])) ** 2 +
           ((xx - c[2]) / (0.46 * shp[2])) ** 2) <= 1
    seeds = rs.rand(24, 3) * np.array(shp)
    lab = np.argmin(((np.stack([zz, yy, xx], -1)[..., None, :] - seeds) ** 2).sum(-1), -1)
    ids = np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13, 77, 7])[lab % 12] * ell
    case = {"Gen": ids.astype(np.float32), "T1": (rs.rand(*shp) * ell).astype(np.float32),
            "segmentation": np.where(ids == 77, 2, ids).astype(np.int32),
            "distance": [(rs.rand(*shp) * 255).astype(np.float32) for _ in range(4)],
            "registration": [(rs.randn(*shp) * 500).astype(np.float32) for _ in range(3)]}
    if with_prob:
        blob = np.exp(-(((zz - c[0] - 3) / 5.) ** 2 + ((yy - c[1] + 2) / 4.) ** 2 + ((xx - c[2]) / 6.) ** 2))
        case["pathology_prob"] = (blob * ell).astype(np.float32)
    return case


def register(prefix, case):
    """File names of datasets.py:520-540 for t1 = prefix + 'T1w.nii'."""
    REG[prefix + "T1w.nii"] = MemVol(case["T1"])
    REG[prefix + "generation_labels.nii"] = MemVol(case["Gen"])
    REG[prefix + "brainseg_with_extracerebral.nii"] = MemVol(case["segmentation"])
    for nm, v in zip(("lp", "lw", "rp", "rw"), case["distance"]):
        REG[prefix + nm + "_dist_map.nii"] = MemVol(v)
    for nm, v in zip(("x", "y", "z"), case["registration"]):
        REG[prefix + "mni_reg." + nm + ".nii"] = MemVol(v)


def gen_args(size, t1_prob, overrides):
    import utils.misc as um
    g = um.preprocess_cfg([R + "/cfgs/generator/default.yaml", R + "/cfgs/generator/train/brain_id.yaml"], cfg_dir="")
    g.generator.size = list(size)
    g.generator.all_samples, g.generator.mild_samples = 2, 1
    g.task = Namespace(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
                       registration=True, super_resolution=True, age=False, surface=False, pathology=True,
                       contrastive=False)
    g.modality_probs = Namespace(MEM=Namespace(T1=t1_prob, T2=0., FLAIR=0., CT=0., synth=1.))
    g.pathology_shape_generator.max_nt = 4
    g.pathology_shape_generator.V_multiplier = 40
    g.mild_generator.bf_scale_min, g.mild_generator.bf_scale_max = 0.03, 0.05   # 0.01-0.02 x 32 voxels rounds to a 0-size field
    for k, v in overrides.items():
        tgt = g
        parts = k.split(".")
        for p in parts[:-1]:
            tgt = getattr(tgt, p)
        setattr(tgt, parts[-1], v)
    return g


def build(cls_name, g, prefix):
    import Generator.datasets as D
    cls = getattr(D, cls_name)
    ds = object.__new__(cls)
    # BaseGen.__init__ (datasets.py:28-44) minus prepare_paths
    ds.gen_args = g
    ds.split = g.split
    ds.synth_args = g.generator
    ds.shape_gen_args = g.pathology_shape_generator
    ds.real_image_args = g.real_image_generator
    ds.synth_image_args = g.synth_image_generator
    ds.augmentation_steps = vars(g.augmentation_steps)
    ds.input_prob = vars(g.modality_probs)
    ds.device = "cpu"
    ds.prepare_tasks()
    ds.ages, ds.names, ds.datasets, ds.datasets_num, ds.datasets_len = [], [[prefix + "T1w.nii"]], ["MEM"], 1, [1]
    ds.pathology_type = None
    ds.prepare_grid()
    ds.prepare_one_hot()
    if cls_name == "BrainIDGen":                               # BrainIDGen.__init__ (datasets.py:691-697)
        ds.all_samples, ds.mild_samples = g.generator.all_samples, g.generator.mild_samples
        ds.mild_generator_args, ds.severe_generator_args = g.mild_generator, g.severe_generator
    return ds, D


class Recorder:
    """torch.rand / torch.randn with every returned tensor kept, in call order."""

    def __init__(self):
        self.log = []
        self._randn, self._rand = torch.randn, torch.rand

    def __enter__(self):
        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.log.append(("randn", t.detach().cpu().numpy().copy()))
            return t

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.log.append(("rand", t.detach().cpu().numpy().copy()))
            return t
        torch.randn, torch.rand = randn, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand = self._randn, self._rand


def ns_to_dict(ns):
    if isinstance(ns, Namespace):
        return {k: ns_to_dict(v) for k, v in vars(ns).items()}
    if isinstance(ns, dict):
        return {k: ns_to_dict(v) for k, v in ns.items()}
    if isinstance(ns, (list, tuple)):
        return [ns_to_dict(v) for v in ns]
    if isinstance(ns, (np.integer,)):
        return int(ns)
    if isinstance(ns, (np.floating,)):
        return float(ns)
    return ns


def tonp(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def run_case(tag, out, cls_name, shp, size, seed, t1_prob=0., with_prob=False, overrides=None, need_crop_full=False):
    overrides = dict(overrides or {})
    case = make_case(shp, seed, with_prob)
    prefix = "/mem/%s." % tag
    register(prefix, case)
    g = gen_args(size, t1_prob, overrides)
    g0 = copy.deepcopy(g)
    ds, D = build(cls_name, g, prefix)
    if with_prob:
        REG["/mem/%s.pathology_prob.nii" % tag] = MemVol(case["pathology_prob"])
        D.pathology_prob_paths[:] = ["/mem/%s.pathology_prob.nii" % tag]
    for attempt in range(40):
        s = seed + 1000 * attempt
        np.random.seed(s)
        random.seed(s)
        torch.manual_seed(s)
        try:
            with Recorder() as rec, torch.no_grad():
                n, dname, mode, target, samples = ds[0]
        except IndexError:
            if not need_crop_full:
                raise
            continue                        # generate_sample's pathology branch needs crop box == size (see header of the test)
        break
    else:
        raise RuntimeError("no seed gave a full crop for " + tag)
    if not isinstance(samples, list):
        samples = [samples]
    out[tag + "/seed"] = np.array(s)
    out[tag + "/cls"] = np.array(cls_name)
    out[tag + "/mode"] = np.array(mode)
    out[tag + "/shape"] = np.array(shp)
    out[tag + "/size"] = np.array(size)
    out[tag + "/t1_prob"] = np.array(t1_prob)
    out[tag + "/overrides"] = np.array(repr(sorted(overrides.items())))
    out[tag + "/cfg_json"] = np.array(json.dumps(ns_to_dict(g0), sort_keys=True))   # the configuration as data (before __getitem__ edits it)
    for k in ("Gen", "T1", "segmentation"):
        out[tag + "/case/" + k] = case[k]
    for j, v in enumerate(case["distance"]):
        out[tag + "/case/distance%d" % j] = v
    for j, v in enumerate(case["registration"]):
        out[tag + "/case/registration%d" % j] = v
    if with_prob:
        out[tag + "/case/pathology_prob"] = case["pathology_prob"]
    out[tag + "/ndraws"] = np.array(len(rec.log))
    for i, (kind, arr) in enumerate(rec.log):
        out[tag + "/draw%03d_%s" % (i, kind)] = arr
    for k, v in target.items():
        if k == "name":
            continue
        out[tag + "/target/" + k] = tonp(v)
    for i, smp in enumerate(samples):
        for k, v in smp.items():
            out[tag + "/sample%d/%s" % (i, k)] = tonp(v)
    print(tag, "seed", s, "mode", mode, "draws", len(rec.log), "target", sorted(k for k in target if k != "name"),
          "pathology voxels", float(np.sum(tonp(target["pathology"]))), "samples", len(samples), flush=True)


if __name__ == "__main__":
    out = {}
    # A: BrainIDGen, synthetic input from the label map (crop smaller than the volume), no pathology drawn,
    #    random linear mix with the real T1 (mix_synth_prob 1), two samples (mild + severe)
    run_case("A", out, "BrainIDGen", (40, 36, 44), (32, 32, 32), 11,
             overrides={"generator.pathology_prob": 0., "mix_synth_prob": 1.0, "generator.flip_prob": -10.})
    # B: BrainIDGen, synthetic input, flipped (flip_prob 10: randn() < 10), random Perlin pathology shape; the volume
    #    has the generator's size so that the crop box is the whole volume (the reference's own pathology branch of
    #    generate_sample indexes the deformed image with a crop-shaped mask, datasets.py:389-391)
    run_case("B", out, "BrainIDGen", (32, 32, 32), (32, 32, 32), 23, need_crop_full=True,
             overrides={"generator.pathology_prob": 1.0, "generator.random_shape_prob": 1.0, "generator.flip_prob": 10.,
                        "generator.photo_prob": 0., "mix_synth_prob": 0.})
    # C: BaseGen (one sample, datasets.py:638-681), real T1 input (modality_probs T1 = 1), random Perlin pathology
    #    shape encoded into the image with the T1 direction (get_pathology_direction -> False), crop smaller than the
    #    volume.  (The file-based pathology source cannot run in the reference: read_and_deform_pathology calls
    #    read_and_deform without its `mask` argument, Generator/utils.py:442 -> TypeError; so the advected variant
    #    (augment_pathology) is pinned by its own fixtures, synth_perlin_pde.npz, not through __getitem__.)
    run_case("C", out, "BaseGen", (40, 36, 44), (32, 32, 32), 37, t1_prob=1.0,
             overrides={"generator.pathology_prob": 1.0, "generator.random_shape_prob": 1.0, "generator.flip_prob": -10.,
                        "generator.photo_prob": 0.})
    np.savez_compressed(os.path.join(HERE, "gen_chain.npz"), **out)
    print("gen_chain.npz", os.path.getsize(os.path.join(HERE, "gen_chain.npz")), "bytes,", len(out), "arrays")
