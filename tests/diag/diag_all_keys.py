"""Stress: the batched / graph / two-lane flow against the eager tile-by-tile flow, every stitched key, many volumes.
   python tests/diag/diag_all_keys.py [size=256] [reps=40] [pipelined=0]
   pipelined = K: K volumes are submitted back to back WITHOUT a synchronisation in between (the stitch of one volume then
   runs beside the first tiles of the next, as in bench.py's timed region) and compared afterwards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from brainfm_amd import test_utils as TU
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
s.set_atlas(*bench.make_atlas())
vols = [bench.make_volume(n, dev)]
vols.append(torch.flip(vols[0], dims=[3]) * 0.7 + 0.05 * (vols[0] != 0))
refs = []
for v in vols:
    e, ranges, cnt = TU.tiled_inference(v, s, [80] * 3, [160] * 3, graphs=False, batched=False)
    refs.append({k: t.clone() for k, t in e.items()})
TU.prepare_tile_graphs(vols[0], s, [80] * 3, [160] * 3)
bad_total = 0
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if pipe:
    for r0 in range(0, reps, pipe):
        outs = []
        for rep in range(r0, min(reps, r0 + pipe)):
            outs.append((rep, TU.tiled_inference(vols[rep % 2], s, [80] * 3, [160] * 3, graphs=True)[0]))
        torch.cuda.synchronize()
        for rep, g in outs:
            bad = {k: int((refs[rep % 2][k] != g[k]).sum()) for k in g}
            if sum(bad.values()):
                print("rep", rep, {k: b for k, b in bad.items() if b}, flush=True)
            bad_total += sum(bad.values())
        del outs
    print("volumes %d x %d^3 in pipelines of %d, %d keys each: differing voxels in total %d" % (reps, n, pipe, len(refs[0]), bad_total))
    sys.exit(0)
for rep in range(reps):
    v = vols[rep % 2]
    g, _, _ = TU.tiled_inference(v, s, [80] * 3, [160] * 3, graphs=True)
    torch.cuda.synchronize()
    bad = {k: int((refs[rep % 2][k] != g[k]).sum()) for k in g}
    nb = sum(bad.values())
    bad_total += nb
    if nb:
        print("rep", rep, {k: b for k, b in bad.items() if b}, flush=True)
    del g
print("volumes %d x %d^3, %d keys each: differing voxels in total %d" % (reps, n, len(refs[0]), bad_total))
