"""BASELINE / SURVEY config 5: the on-device synthesis pipeline feeding a training step, one item per GPU.

    python scripts/bench_config5.py [items=4] [size=160]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_config5.py ...

Per item: a seeded 192^3 Voronoi label volume (~40 seeds mapped onto generation labels, background outside
an ellipsoid) stands in for the NIfTI read; the generator (brainfm_amd.generator, BrainIDGen counterpart: ShapeID Perlin
shape + upwind advection integrated with dopri5 for the pathology, affine + non-linear deformation, label-to-image
synthesis, augmentation chain, targets) draws all_samples = 4 augmented inputs (2 mild) of size^3; the consumer is ONE
training iteration of the full-width U-Net on the build's own kernels (forward, losses, backward, AdamW; under
torch.distributed the gradients are averaged by one flat all-reduce: DDP, weak scaling, batch = N items).
Reports generator and step times, items/s and generated voxels/s per GPU.  Weights: default nn init under seed 1.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from argparse import Namespace
from brainfm_amd import generator as G
from brainfm_amd import test_utils as TU
from brainfm_amd import train as TR

items = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda:%d" % local)
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)


def voronoi_case(seed, n=192, nseeds=40):
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
    return {"name": "voronoi%d" % seed, "Gen": ids.astype(np.float32), "T1": rs.rand(*shp).astype(np.float32) * ell,
            "segmentation": ids.astype(np.int32),
            "distance": [rs.rand(*shp).astype(np.float32) * 255 for _ in range(4)],
            "registration": [rs.randn(*shp).astype(np.float32) * 500 for _ in range(3)]}


def gen_args(size):
    g = Namespace(size=[size] * 3, photo_prob=0.2, max_rotation=15, max_shear=0.2, max_scaling=0.2, nonlin_scale_min=0.03,
                  nonlin_scale_max=0.06, nonlin_std_max=4, bf_scale_min=0.02, bf_scale_max=0.04, bf_std_min=0.1,
                  bf_std_max=0.6, gamma_std=0.1, noise_std_min=0.05, noise_std_max=1., random_shift=False,
                  nonlinear_transform=True, left_hemis_only=False, low_res_only=False, ct_prob=0, flip_prob=0.,
                  pathology_prob=1.0, random_shape_prob=1.0, augment_pathology=True, bspline_zooming=False,
                  mild_samples=2, all_samples=4)
    shp = Namespace(perlin_res=[2, 2, 2], integ_method="dopri5", bc="neumann", V_multiplier=500, dt=0.1, max_nt=10,
                    pathol_thres=0.2, pathol_tol=1e-5, mask_percentile_min=85., mask_percentile_max=99.)
    task = Namespace(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
                     registration=True, super_resolution=True, surface=False, pathology=True, contrastive=False)
    return Namespace(generator=g, pathology_shape_generator=shp, task=task, max_surf_distance=3.0,
                     augmentation_steps=["gamma", "bias_field", "resample", "noise"], dataset_option="brain_id",
                     mix_synth_prob=0.)


np.random.seed(100 + rank)
torch.manual_seed(100 + rank)
ga = gen_args(N)
ds = G.build_datasets(ga, str(dev), cases=[voronoi_case(7 + rank)])["all"]
tasks = dict(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
             registration=True, super_resolution=True, surface=False, pathology=False, contrastive=False)
gi, ti = TU.default_inference_args(f_maps=64, num_levels=6, tasks=tasks, size=(N, N, N))
torch.manual_seed(1)
sess = TU.InferenceSession(gi, ti, dev, passes=3)
tail = sess.model.head.tail(sess.engine)
names = ["T1", "T1_grad", "seg_ce", "seg_dice", "distance", "bias_field_log", "registration", "registration_grad", "SR",
         "SR_grad"]
nseg = tail.desc.n_seg
step = TR.TrainStep(sess.engine, tail, names, {"loss_" + n: 1.0 for n in names}, torch.full((nseg,), 1.0 / nseg),
                    all_samples=ga.generator.all_samples, lr=1e-4, scaler=TR.LossScaler())


def make_item():
    _, _, _, target, samples = ds[0]
    target = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in target.items()}      # collate: batch dim
    samples = [{k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in s.items()} for s in samples]
    return target, samples


def train_on(target, samples):
    ld, total, ok = step.step([s["input"] for s in samples], target, samples)
    return total, ok


# generate, then train (a producer thread synthesising item i+1 on a side stream while the GPU trains on item i was
# tried: 1.36 vs 1.37 items/s -- both halves are bound by the one Python thread that submits their launches)
t, s_ = make_item()
train_on(t, s_)                                       # warm-up: conv autotune, optimiser state
torch.cuda.synchronize()
tg = tt = 0.0
if world > 1:
    dist.barrier()
t_all = time.perf_counter()
for _ in range(items):
    t0 = time.perf_counter()
    t, s_ = make_item()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    total, ok = train_on(t, s_)
    torch.cuda.synchronize()
    tg += t1 - t0
    tt += time.perf_counter() - t1
if world > 1:
    dist.barrier()
t_all = time.perf_counter() - t_all

if rank == 0:
    nv = ga.generator.all_samples * N ** 3
    print("config 5, %d rank(s), %d item(s) per rank of %d x %d^3: generator %.0f ms + training iteration %.0f ms per item "
          "(last loss %.4f, stepped %s)" % (world, items, ga.generator.all_samples, N, 1e3 * tg / items, 1e3 * tt / items,
                                            total, ok))
    print("%.2f items/s = %.1f Mvoxel/s generated and trained on, whole job; %.2f items/s per GPU; peak memory %.1f GB"
          % (world * items / t_all, world * items * nv / t_all / 1e6, items / t_all,
             torch.cuda.max_memory_allocated() / 2 ** 30))
if world > 1:
    dist.destroy_process_group()
