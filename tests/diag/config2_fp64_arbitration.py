"""BASELINE config 2 (one 160^3 volume, seed 0, all heads, the bench weights: default nn init under seed 1) with the oracle
evaluated in float64 as the arbiter: how many labels does each fp32 evaluation -- the HIP path, and the
torch-CPU fp32 oracle the suite compares with -- flip against float64, and how close were the two best classes there?
Opt-in (minutes of host time, ~40 GB of host memory): not part of the driver-run suite; the result is committed under
profiles/.     python tests/diag/config2_fp64_arbitration.py [size=160]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle import unet_ref as O  # noqa: E402
from brainfm_amd import test_utils as TU  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda:0")
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
sd = {k: v.detach().cpu() for k, v in s.model.state_dict().items()}
torch.manual_seed(0)
x = torch.rand(1, 1, N, N, N)
out, _ = s.forward_fused(x.to(dev), want_feat=False, want_seg=True)
lab_hip = out["label"].cpu()
seg_hip = out["segmentation"].cpu()
floats_hip = {k: v.cpu() for k, v in out.items() if k not in ("label", "segmentation", "feat") and torch.is_tensor(v)}
del s, out
torch.cuda.empty_cache()
threads = max(1, (os.cpu_count() or 2) // 2)
torch.set_num_threads(threads)
t0 = time.time()
with torch.no_grad():
    ref32 = O.forward_all(x, sd, f_maps=64, num_levels=6)
t32 = time.time() - t0
lab32, seg32 = ref32["label"], ref32["segmentation"]
f32 = {k: v for k, v in ref32.items() if k not in ("label", "segmentation", "feat")}
del ref32
t0 = time.time()
with torch.no_grad():
    ref64 = O.forward_all(x.double(), {k: v.double() for k, v in sd.items()}, f_maps=64, num_levels=6)
t64 = time.time() - t0
lab64, p = ref64["label"], ref64["segmentation"]
top2 = torch.topk(p, 2, dim=1).values
gap = ((top2[:, 0] - top2[:, 1]) / top2[:, 0])[:, None]
m_cpu, m_hip, m_both = lab32 != lab64, lab_hip != lab64, lab_hip != lab32
print("config 2, %d^3 volume (seed 0), bench weights (seed 1), %d host threads: fp32 oracle %.0f s, float64 oracle %.0f s"
      % (N, threads, t32, t64))
print("labels that differ from the float64 oracle: HIP path %d, torch-CPU fp32 oracle %d; HIP vs fp32 oracle %d (of %d)"
      % (int(m_hip.sum()), int(m_cpu.sum()), int(m_both.sum()), lab64.numel()))
for name, m in (("HIP vs float64", m_hip), ("fp32 oracle vs float64", m_cpu), ("HIP vs fp32 oracle", m_both)):
    if int(m.sum()):
        print("  %-24s float64 top-2 relative gap at those voxels: max %.3e, median %.3e"
              % (name + ":", float(gap[m].max()), float(gap[m].median())))
print("voxels whose float64 top-2 gap is below 1e-5: %d; below 2.5e-5: %d" % (int((gap < 1e-5).sum()), int((gap < 2.5e-5).sum())))


def rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(1e-6, float(b.double().abs().max()))


print("softmax max rel err vs float64: HIP %.2e, fp32 oracle %.2e" % (rel(seg_hip, p), rel(seg32, p)))
worst_h = max(rel(floats_hip[k], ref64[k]) for k in f32 if k in floats_hip)
worst_c = max(rel(f32[k], ref64[k]) for k in f32)
print("15 float maps, worst rel err vs float64: HIP %.2e, fp32 oracle %.2e" % (worst_h, worst_c))
