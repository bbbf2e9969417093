"""Where does the uniform-box shortcut differ?  eager on/off per tile of the bench volume, then graph replay vs eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brainfm_amd import test_utils as TU

dev = torch.device("cuda", 0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
full = bench.make_volume(256, dev)
ranges = TU.tiling_ranges((256,) * 3, [80] * 3, [160] * 3)
eng = s.engine
for i in (0, 1, 4, 13, 26):
    r = ranges[i]
    im = full[:, :, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]].contiguous()
    dims = tuple(im.shape[2:])
    x_cl = eng.to_cl(im)
    res = {}
    for on in (False, True, True):
        eng.uniform_skip = on
        feats = eng.backbone_cl(x_cl, dims)
        torch.cuda.synchronize()
        res.setdefault(on, []).append([f.clone() for f, _ in feats])
    a, b, c = res[False][0], res[True][0], res[True][1]
    print("tile", i, dims, "on vs off differing:", [int((x != y).sum()) for x, y in zip(a, b)],
          "on vs on:", [int((x != y).sum()) for x, y in zip(b, c)], flush=True)

eng.uniform_skip = True
eager, _, _ = TU.tiled_inference(full, s, [80] * 3, [160] * 3, graphs=False)
eager = {k: v.clone() for k, v in eager.items()}
for lanes in (1, 2):
    torch.manual_seed(1)
    s2 = TU.InferenceSession(ga, ta, dev, passes=3)
    s2.lanes = lanes
    TU.prepare_tile_graphs(full, s2, [80] * 3, [160] * 3)
    for rep in range(3):
        acc, _, _ = TU.tiled_inference(full, s2, [80] * 3, [160] * 3, graphs=True)
        torch.cuda.synchronize()
        bad = {k: int((acc[k] != eager[k]).sum()) for k in eager}
        print("lanes", lanes, "rep", rep, {k: v for k, v in bad.items() if v}, flush=True)
