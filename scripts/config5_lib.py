"""BASELINE / SURVEY config 5 pieces shared by scripts/bench_config5.py and the -m gpu test: the synthetic label case
(a seeded 192^3 Voronoi volume of ~40 seeds mapped onto generation labels, background outside an ellipsoid,
standing in for the NIfTI read), the generator settings (cfgs/generator/train/brain_id.yaml:30-47 tasks + pathology with
the shape_id.yaml PDE settings, all_samples 4 / mild_samples 2 as cfgs/generator/test/demo_test.yaml:91-92), and the
consumer: one TrainStep of the full-width U-Net."""
from argparse import Namespace

import numpy as np
import torch

from brainfm_amd import generator as G
from brainfm_amd import test_utils as TU
from brainfm_amd import train as TR

LOSS_NAMES = ["T1", "T1_grad", "seg_ce", "seg_dice", "distance", "bias_field_log", "registration", "registration_grad", "SR",
              "SR_grad"]


def voronoi_case(seed, n=192, nseeds=40, pathology_prob=False):
    rs = np.random.RandomState(seed)
    ax = np.arange(n, dtype=np.float32)
    zz, yy, xx = np.meshgrid(ax, ax, ax, indexing="ij")
    c = (n - 1) / 2.0
    ell = ((zz - c) / (0.45 * n)) ** 2 + ((yy - c) / (0.42 * n)) ** 2 + ((xx - c) / (0.40 * n)) ** 2 <= 1
    pts = rs.rand(nseeds, 3).astype(np.float32) * n
    best = np.full((n, n, n), np.inf, dtype=np.float32)
    lab = np.zeros((n, n, n), dtype=np.int32)
    for i, p in enumerate(pts):                       # running nearest seed: no (n^3, nseeds) array
        d = (zz - p[0]) ** 2 + (yy - p[1]) ** 2 + (xx - p[2]) ** 2
        m = d < best
        best[m] = d[m]
        lab[m] = i
    ids = np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13, 7, 8, 16, 18, 26, 28])[lab % 16] * ell
    shp = (n, n, n)
    case = {"name": "voronoi%d" % seed, "Gen": ids.astype(np.float32), "T1": rs.rand(*shp).astype(np.float32) * ell,
            "segmentation": ids.astype(np.int32),
            "distance": [rs.rand(*shp).astype(np.float32) * 255 for _ in range(4)],
            "registration": [rs.randn(*shp).astype(np.float32) * 500 for _ in range(3)]}
    if pathology_prob:
        # a resident lesion-probability volume (what a dataset's pathology_prob file holds): three smooth blobs inside the
        # head.  With random_shape_prob < 1 the item then takes the file-based branch of read_and_deform_pathology, i.e.
        # trilinear warp -> Perlin velocity field -> dopri5 advection (augment_pathology) -> binarize: the chain BASELINE
        # config 5 names (Generator/utils.py:428-459, 542-560)
        pp = np.zeros(shp, dtype=np.float32)
        for cz, cy, cx, r in ((0.40, 0.45, 0.55, 0.09), (0.58, 0.52, 0.40, 0.07), (0.50, 0.60, 0.62, 0.05)):
            d2 = (zz - cz * n) ** 2 + (yy - cy * n) ** 2 + (xx - cx * n) ** 2
            pp = np.maximum(pp, np.exp(-d2 / (2.0 * (r * n) ** 2)).astype(np.float32))
        case["pathology_prob"] = pp * ell
    return case


def gen_args(size, random_shape_prob=1.0):
    g = Namespace(size=[size] * 3, photo_prob=0.2, max_rotation=15, max_shear=0.2, max_scaling=0.2, nonlin_scale_min=0.03,
                  nonlin_scale_max=0.06, nonlin_std_max=4, bf_scale_min=0.02, bf_scale_max=0.04, bf_std_min=0.1,
                  bf_std_max=0.6, gamma_std=0.1, noise_std_min=0.05, noise_std_max=1., random_shift=False,
                  nonlinear_transform=True, left_hemis_only=False, low_res_only=False, ct_prob=0, flip_prob=0.,
                  pathology_prob=1.0, random_shape_prob=random_shape_prob, augment_pathology=True, bspline_zooming=False,
                  mild_samples=2, all_samples=4)
    shp = Namespace(perlin_res=[2, 2, 2], integ_method="dopri5", bc="neumann", V_multiplier=500, dt=0.1, max_nt=10,
                    pathol_thres=0.2, pathol_tol=1e-5, mask_percentile_min=85., mask_percentile_max=99.)
    task = Namespace(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
                     registration=True, super_resolution=True, surface=False, pathology=True, contrastive=False)
    return Namespace(generator=g, pathology_shape_generator=shp, task=task, max_surf_distance=3.0,
                     augmentation_steps=["gamma", "bias_field", "resample", "noise"], dataset_option="brain_id",
                     mix_synth_prob=0.)


def build(dev, size, rank=0):
    """(dataset, TrainStep, gen_args) on `dev`; seeds: NumPy / torch 100 + rank for the generator, 1 for the weights."""
    np.random.seed(100 + rank)
    torch.manual_seed(100 + rank)
    ga = gen_args(size)
    ds = G.build_datasets(ga, str(dev), cases=[voronoi_case(7 + rank)])["all"]
    tasks = dict(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
                 registration=True, super_resolution=True, surface=False, pathology=False, contrastive=False)
    gi, ti = TU.default_inference_args(f_maps=64, num_levels=6, tasks=tasks, size=(size, size, size))
    gstate = torch.random.get_rng_state()
    torch.manual_seed(1)
    sess = TU.InferenceSession(gi, ti, dev, passes=3)
    torch.random.set_rng_state(gstate)
    tail = sess.model.head.tail(sess.engine)
    nseg = tail.desc.n_seg
    step = TR.TrainStep(sess.engine, tail, LOSS_NAMES, {"loss_" + n: 1.0 for n in LOSS_NAMES}, torch.full((nseg,), 1.0 / nseg),
                        all_samples=ga.generator.all_samples, lr=1e-4, scaler=TR.LossScaler())
    step._session = sess                                  # keeps the parameter-carrying modules alive
    return ds, step, ga


def collate(target, samples):
    """DataLoader collation of one item: a batch dimension on every tensor."""
    target = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in target.items()}
    samples = [{k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in s.items()} for s in samples]
    return target, samples
