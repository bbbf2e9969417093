"""Write brainfm_amd/conv_tune_gfx950.json: the conv variant per layer shape for the BASELINE configurations, timed
once on an MI355X (UNetEngine._autotune) and then shipped with the library, so that every process launches the same
variants and the same volume gives the same bits run to run (the variants agree to ~1e-6, not bit for bit).

    BFM_CONV_TUNE=retune BFM_CONV_TUNE_SAVE=1 python scripts/make_tune_table.py        (on the GPU box; ~1 min)

Shapes covered: every tile shape of the reference tiling of 256^3 / 512^3 volumes (160/80 mixes) and a 128^3 tile
(inference: forward layers, skip halves in accumulate mode), and one training iteration at 128^3 and 160^3 (the
transposed data-gradient layers).  Anything else is timed in-process on first use, as before."""
import os
import sys

os.environ.setdefault("BFM_CONV_TUNE", "retune")
os.environ.setdefault("BFM_CONV_TUNE_SAVE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch

import bench
from brainfm_amd import engine as E
from brainfm_amd import test_utils as TU

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
sess = TU.InferenceSession(ga, ta, dev, passes=3)
full = bench.make_volume(256, dev)
shapes = sorted({tuple(b - a for a, b in r) for r in TU.tiling_ranges((256,) * 3, [80] * 3, [160] * 3)}) + [(128, 128, 128)]
for s in shapes:
    TU._run_tile(sess, full[:, :, :s[0], :s[1], :s[2]], raw=True)
    torch.cuda.synchronize()
    print("inference tile", s, "->", len(sess.engine.conv_choices()), "shapes timed", flush=True)
del sess
import config5_lib as C5
for size in (128, 160):
    ds, step, _ = C5.build(dev, size)
    _, _, _, target, samples = ds[0]
    t, sm = C5.collate(target, samples)
    step.step([x["input"] for x in sm[:1]], t, sm[:1])
    torch.cuda.synchronize()
    print("training iteration at", size, "->", len(step.eng.conv_choices()), "shapes timed", flush=True)
    del ds, step
print("wrote", E.TUNE_FILE)
