"""Full architecture (64 maps, 6 levels, 9 heads) on one tile against the CPU oracle: label mismatches and the
worst relative error per output.  python tests/diag/full_arch_parity.py D H W [passes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import unet_ref as O  # noqa: E402
from brainfm_amd import test_utils as TU  # noqa: E402

D, H, W = [int(v) for v in sys.argv[1:4]]
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sd = O.random_state_dict(1, 64, 6, seed=5)
g = torch.Generator().manual_seed(9)
zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
ell = (((zz - (D - 1) / 2) / (0.45 * D)) ** 2 + ((yy - (H - 1) / 2) / (0.42 * H)) ** 2 +
       ((xx - (W - 1) / 2) / (0.44 * W)) ** 2) <= 1
x = torch.rand(1, 1, D, H, W, generator=g) * ell[None, None]
t0 = time.time()
with torch.no_grad():
    ref = O.forward_all(x, sd, f_maps=64, num_levels=6)
t_cpu = time.time() - t0
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
s = TU.InferenceSession(ga, ta, torch.device("cuda:0"), state_dict=sd, passes=passes)
out, _ = s.forward_fused(x.cuda())


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) / max(1e-6, float(np.abs(b).max()))


res = {}
for k, v in ref.items():
    if k == "feat":
        for i, f in enumerate(v):
            res["feat%d" % i] = rel(out["feat"][i].cpu().numpy(), f.numpy())
    elif k == "label":
        res["label_mismatch"] = int((out[k].cpu() != v).sum())
    else:
        res[k] = rel(out[k].cpu().numpy(), v.numpy())
print("dims", (D, H, W), "passes", passes, "cpu %.1f s" % t_cpu, "threads", torch.get_num_threads())
print({k: (v if isinstance(v, int) else float("%.2e" % v)) for k, v in res.items()})

# who is right at the mismatching voxels?  the same oracle evaluated in float64
if os.environ.get("FP64", "1") == "1":
    t0 = time.time()
    with torch.no_grad():
        ref64 = O.forward_all(x.double(), {k: v.double() for k, v in sd.items()}, f_maps=64, num_levels=6)
    lab64 = ref64["label"]
    p = ref64["segmentation"]
    top2 = torch.topk(p, 2, dim=1).values
    gap = ((top2[:, 0] - top2[:, 1]) / top2[:, 0])[:, None]           # relative gap of the two best classes, fp64
    m_cpu = ref["label"] != lab64
    m_hip = out["label"].cpu() != lab64
    m_both = out["label"].cpu() != ref["label"]
    print("fp64 oracle %.1f s; label mismatches vs fp64: torch-CPU fp32 %d, HIP %d; HIP vs torch-CPU fp32 %d" % (
        time.time() - t0, int(m_cpu.sum()), int(m_hip.sum()), int(m_both.sum())))
    for name, m in (("cpu32-vs-64", m_cpu), ("hip-vs-64", m_hip), ("hip-vs-cpu32", m_both)):
        if int(m.sum()):
            print("  %s: top-2 relative gap at those voxels: max %.3e median %.3e" % (name, float(gap[m].max()),
                                                                                   float(gap[m].median())))
    print("  voxels with gap < 1e-5: %d of %d" % (int((gap < 1e-5).sum()), gap.numel()))
    e32 = rel(ref["segmentation"].numpy(), p.numpy())
    eh = rel(out["segmentation"].cpu().numpy(), p.numpy())
    print("  softmax max rel err vs fp64: torch-CPU fp32 %.2e, HIP %.2e" % (e32, eh))
