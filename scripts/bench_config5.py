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


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config5_lib as C5

ds, step, ga = C5.build(dev, N, rank)


def make_item():
    _, _, _, target, samples = ds[0]
    return C5.collate(target, samples)


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
