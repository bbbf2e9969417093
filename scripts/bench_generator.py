"""End-to-end time of one on-device training item (BrainIDGen.__getitem__ counterpart: ShapeID pathology + affine /
non-linear deformation + label-to-image synthesis + augmentation chain + targets) at the reference's 128^3 training size
(cfgs/generator/default.yaml:63), from an in-memory case.  usage: python scripts/bench_generator.py [size=128] [reps=5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_gpu_synth as SY
from brainfm_amd import generator as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rs = np.random.RandomState(0)
shp = (N + 32, N + 24, N + 40)
zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
c = [s / 2 for s in shp]
ell = (((zz - c[0]) / (0.42 * shp[0])) ** 2 + ((yy - c[1]) / (0.42 * shp[1])) ** 2 + ((xx - c[2]) / (0.42 * shp[2])) ** 2) <= 1
lab = ((zz // 16) * 7 + (yy // 16) * 3 + (xx // 16)) % 10
ids = np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13])[lab] * ell
case = {"name": "toy", "Gen": ids.astype(np.float32), "T1": rs.rand(*shp).astype(np.float32) * ell,
        "segmentation": ids.astype(np.int32),
        "distance": [rs.rand(*shp).astype(np.float32) * 255 for _ in range(4)],
        "registration": [rs.randn(*shp).astype(np.float32) * 500 for _ in range(3)]}
np.random.seed(3)
torch.manual_seed(3)
ga = SY._gen_args(size=(N, N, N))
ds = G.build_datasets(ga, "cuda:0", cases=[case])["all"]
ds[0]
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    _, _, _, target, samples = ds[0]
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("generator item %d^3 (%d augmented samples, pathology on): median %.1f ms, min %.1f ms over %d items = %.1f Mvoxel/s"
      % (N, len(samples), 1e3 * float(np.median(ts)), 1e3 * min(ts), reps, N ** 3 * len(samples) / float(np.median(ts)) / 1e6))
