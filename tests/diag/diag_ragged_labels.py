import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import unet_ref as O
from brainfm_amd import test_utils as TU
dev = torch.device("cuda:0")
sd = O.random_state_dict(1, 16, 3, seed=23)
g = torch.Generator().manual_seed(5)
full = torch.rand(1, 1, 45, 38, 51, generator=g)
full[:, :, :, :5] = 0
stride, win = [14, 14, 14], [27, 27, 27]
ref, ranges_ref, cnt_ref = O.tiled_inference(full, sd, stride, win, f_maps=16, num_levels=3)
ga, ta = TU.default_inference_args(f_maps=16, num_levels=3)
s = TU.InferenceSession(ga, ta, dev, state_dict=sd, passes=3)
acc, ranges, cnt = TU.tiled_inference(full.to(dev), s, stride, win)
for k, v in ref.items():
    a = acc[k].cpu().numpy().astype(np.float64); b = np.asarray(v).astype(np.float64)
    d = np.abs(a - b)
    print(k, "max abs diff %.3e  rel %.3e  n(diff>1e-3*max) %d" % (d.max(), d.max() / max(1e-6, np.abs(b).max()), int((d > 1e-3 * np.abs(b).max()).sum())))
# per-tile label check on the tile(s) covering the worst voxel
lab = acc["label"].cpu().numpy(); lr = np.asarray(ref["label"])
idx = np.argwhere(np.abs(lab - lr) > 1e-6)
print("label voxels differing:", len(idx), idx[:5])
kinds = [(ly.name, ly.kind) for blk in s.engine.enc + s.engine.dec for ly in blk]
print(kinds)
print({k: int(c[6]) for k, c in s.engine._plan_cache.items()})
# softmax margin at the differing voxels: the oracle in float64 on every tile that covers the voxel
sd64 = {k: v.double() for k, v in sd.items()}
for (z, y, x) in idx[:3]:
    for r in ranges_ref:
        if all(r[a][0] <= v < r[a][1] for a, v in zip(range(3), (z, y, x))):
            t = full[:, :, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]]
            o = O.forward_all(t.double(), sd64, f_maps=16, num_levels=3)
            o32 = O.forward_all(t, sd, f_maps=16, num_levels=3)
            seg = o["segmentation"][0, :, z - r[0][0], y - r[1][0], x - r[2][0]]
            top = torch.topk(seg, 2)
            out, _ = s.forward_fused(t.to(dev))
            print("voxel", (int(z), int(y), int(x)), "tile", [tuple(q) for q in r], "fp64 top-2 gap %.3e (classes %s)  labels: fp64 %d fp32 oracle %d hip %d"
                  % (float(top.values[0] - top.values[1]), top.indices.tolist(), int(o["label"][0, 0, z - r[0][0], y - r[1][0], x - r[2][0]]),
                     int(o32["label"][0, 0, z - r[0][0], y - r[1][0], x - r[2][0]]), int(out["label"][0, 0, z - r[0][0], y - r[1][0], x - r[2][0]])))
