"""One BrainIDGen.__getitem__ at BASELINE config-5 settings (192^3 Voronoi case, 4 x 160^3 samples, pathology on):
per-phase wall breakdown, launch-free wall vs device time, and (under rocprofv3 --kernel-trace) a marker kernel
(`bbox_nonzero` on 64 voxels) before every timed item so that scripts/prof_summary.py can cut exact item windows.

    python scripts/prof_synth_item.py [items=5] [size=160] [phases=1]
    rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_synth_item.py 5 160 0
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

import config5_lib as C5
from brainfm_amd import generator as G
from brainfm_amd import test_utils as TU

items = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
phases = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
np.random.seed(100)
torch.manual_seed(100)
ga = C5.gen_args(N)
ds = G.build_datasets(ga, str(dev), cases=[C5.voronoi_case(7)])["all"]
marker_src = torch.ones(4, 4, 4, device=dev)


def marker():
    TU.zero_crop(marker_src)


for _ in range(2):
    ds[0]
torch.cuda.synchronize()

ts = []
for _ in range(items):
    marker()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ds[0]
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
marker()
torch.cuda.synchronize()
nv = ga.generator.all_samples * N ** 3
med = float(np.median(ts))
print("item %d^3 x %d samples: median %.1f ms, min %.1f, max %.1f over %d items = %.1f Mvoxel/s generated"
      % (N, ga.generator.all_samples, 1e3 * med, 1e3 * min(ts), 1e3 * max(ts), items, nv / med / 1e6))

if phases:
    acc = {}

    def timed(obj, name, label=None):
        fn = getattr(obj, name)
        label = label or name

        def w(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize()
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, w)

    for nm in ("generate_deformation", "_targets", "generate_sample", "_read_input", "get_setup_params"):
        timed(ds, nm)
    # inside the phases (these overlap with the ones above: nested)
    timed(ds, "read_and_deform_pathology", "  (in _targets) read_and_deform_pathology")
    timed(ds, "augment_sample", "  (in generate_sample) augment_sample")
    timed(ds, "encode_pathology", "  (in augment_sample) encode_pathology")
    timed(ds, "get_contrast", "  (in generate_sample) get_contrast")
    t0 = time.perf_counter()
    for _ in range(items):
        ds[0]
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("phase breakdown (synchronised around every phase, so the sum exceeds the unsynchronised item): %.1f ms per item"
          % (1e3 * tot / items))
    for k, v in acc.items():
        print("  %-48s %8.2f ms per item" % (k, 1e3 * v / items))
