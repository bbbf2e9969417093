"""The atlas-gather hazard inside the real tile flow, with the texel loads switched back to plain ones
(BFM_ATLAS_PLAIN_LOADS=1), under the switches that could tell WHAT in the flow it depends on (round 3):

    BFM_ATLAS_PLAIN_LOADS=1 [BFM_MASK_SKIP=0] [BFM_LANES=1] [BFM_COMPACT=0] python tests/diag/diag_atlas_flow.py [volumes]

Reference = the eager, tile-by-tile flow with sc0 sc1 loads is not available in the same process (the switch is read
once), so the check is the property the constant atlas gives: with atlas == 100 everywhere, every stitched
deformed_atlas voxel must be 0 (outside the mask, or sampled outside the atlas) or exactly 100.
Prints per volume the number of voxels that are neither, where they sit relative to the 4x4x16 boxes of the masked last
convolution and the 64-voxel runs of the tail, and whether regx / regy / regz changed from volume to volume.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from brainfm_amd import test_utils as TU

dev = torch.device("cuda:0")
nvol = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
_, aff = bench.make_atlas()
s.set_atlas(torch.full((256, 256, 256), 100.0), aff)
full = bench.make_volume(256, dev)
TU.prepare_tile_graphs(full, s, [80] * 3, [160] * 3)
ref_reg = None
total = 0
for v in range(nvol):
    acc, ranges, cnt = TU.tiled_inference(full, s, [80] * 3, [160] * 3, graphs=True)
    da = acc["deformed_atlas"]
    # where one tile covers a voxel the stitched value is the tile's; elsewhere a mean of 0 / 100 values of the covering tiles
    once = cnt == 1
    bad = once & (da != 0) & (da != 100.0)
    nb = int(bad.sum())
    total += nb
    reg = torch.stack([acc["regx"], acc["regy"], acc["regz"]])
    same = True if ref_reg is None else bool(torch.equal(reg, ref_reg))
    if ref_reg is None:
        ref_reg = reg.clone()
    line = "volume %2d: %d voxels neither 0 nor 100 (cnt == 1 region); reg maps identical to volume 0: %s" % (v, nb, same)
    if nb:
        idx = torch.nonzero(bad)
        z, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
        lin = (z * 256 + y) * 256 + x
        runs = torch.unique(lin // 16).numel()
        line += "; %d runs of 16, x %% 16 of the first: %s, values %s" % (runs, (x[:8] % 16).tolist(), da[bad][:4].tolist())
    print(line, flush=True)
print("switches: plain=%s mask_skip=%s lanes=%s compact=%s -> %d bad voxels in %d volumes" % (
    os.environ.get("BFM_ATLAS_PLAIN_LOADS", "0"), os.environ.get("BFM_MASK_SKIP", "1"), os.environ.get("BFM_LANES", "2"),
    os.environ.get("BFM_COMPACT", "1"), total, nvol), flush=True)
