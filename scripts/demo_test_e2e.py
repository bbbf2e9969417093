"""scripts/demo_test.py end to end on the build's own stack, timed per phase: a NIfTI file on disk -> MRIread (volio) ->
prepare_image on the device (nan_to_num, min-max, resample to 1 mm, axis alignment) -> tiled multi-task inference
(win 160 / stride 80, hipGraph replay, two tiles in flight) -> every stitched map written as NIfTI.

    python scripts/demo_test_e2e.py [size=256] [ext=.nii | .nii.gz]

The input is the synthetic ellipsoid volume of bench.py with 1.2 x 1.0 x 1.1 mm voxels and a rotated axis order, so that
the pre-processing has real work to do.  Weights: default nn init under seed 1 (no checkpoint in this image).
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from brainfm_amd import test_utils as TU
from brainfm_amd import volio

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ext = sys.argv[2] if len(sys.argv) > 2 else ".nii"
dev = torch.device("cuda:0")
tmp = tempfile.mkdtemp()
vol = bench.make_volume(int(round(N / 1.1)), "cpu")[0, 0].numpy().astype(np.float32)
aff = np.array([[0, 0, 1.1, -60.0], [-1.2, 0, 0, 90.0], [0, 1.0, 0, -70.0], [0, 0, 0, 1]], dtype=np.float64)   # permuted + flipped
src = os.path.join(tmp, "input" + ext)
volio.MRIwrite(vol, aff, src)

ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
sess = TU.InferenceSession(ga, ta, dev, passes=3)
sess.use_graphs = True


def run(write=True):
    t = [time.perf_counter()]
    final, orig, high_res, bf, aff_out, crop_start, orig_shp = TU.prepare_image(src, device=dev)
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    full = final if final.dim() == 5 else final[None, None]
    acc, ranges, cnt = TU.tiled_inference(full, sess, [80] * 3, [160] * 3)
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    nbytes = 0
    if write:
        volio.write_device_volumes(acc, aff_out, tmp, ext=ext)
        nbytes = sum(v.numel() * 4 for v in acc.values())
    t.append(time.perf_counter())
    return tuple(full.shape[2:]), len(ranges), len(acc), nbytes, [b - a for a, b in zip(t[:-1], t[1:])]


run(write=False)                                        # tunes the conv variants, captures the graphs
run(write=False)
shape, ntiles, nmaps, nbytes, (tp, ti, tw) = run()
nv = shape[0] * shape[1] * shape[2]
print("input %s on disk (%s voxels of %.1fx%.1fx%.1f mm) -> %s at 1 mm, %d tiles, %d output maps"
      % (ext, "x".join(str(s) for s in vol.shape), 1.2, 1.0, 1.1, "x".join(str(s) for s in shape), ntiles, nmaps))
print("read + prepare_image %.0f ms | tiled inference %.0f ms (%.1f Mvoxel/s) | device->host + write %d maps (%.2f GB, %s) %.0f ms"
      % (1e3 * tp, 1e3 * ti, nv / ti / 1e6, nmaps, nbytes / 1e9, ext, 1e3 * tw))
print("end to end %.2f s = %.1f Mvoxel/s" % (tp + ti + tw, nv / (tp + ti + tw) / 1e6))
