import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_npz, sd_from_npz
from oracle import unet_ref as O
from brainfm_amd import test_utils as TU

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))

d = load_npz("infer_small.npz")
sd = sd_from_npz(d)
x = torch.from_numpy(d["x"])
sd64 = {k: v.double() for k, v in sd.items()}
f64 = O.get_feature(x.double(), sd64, 1, 8, 4, 8, True)
f32 = O.get_feature(x, sd, 1, 8, 4, 8, True)
ga, ta = TU.default_inference_args(f_maps=8, num_levels=4)
s = TU.InferenceSession(ga, ta, "cuda:0", state_dict=sd)
out, _ = s.forward_fused(x.cuda())
for i in range(4):
    g = out["feat"][i].cpu().numpy()
    print("feat%d: hip-vs-golden %.2e  hip-vs-fp64 %.2e  torch32-vs-fp64 %.2e  golden-vs-fp64 %.2e" % (
        i, rel(g, d["feat%d" % i]), rel(g, f64[i].numpy()), rel(f32[i].numpy(), f64[i].numpy()), rel(d["feat%d" % i], f64[i].numpy())))
