import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
from oracle import unet_ref as O
from brainfm_amd import test_utils as TU, backward as BW
f_maps, levels, dims = 64, 3, (16, 12, 20)
sd = O.random_state_dict(1, f_maps, levels, seed=31)
g = torch.Generator().manual_seed(12)
x = torch.rand((1, 1) + dims, generator=g)
def grads(dtype):
    P = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if k.startswith("backbone.")}
    feats = O.get_feature(x.to(dtype), P, f_maps=f_maps, num_levels=levels, unit_feat=False)
    gg = torch.Generator().manual_seed(99)
    R = [torch.randn(f.shape, generator=gg).to(dtype) for f in feats]
    sum((f * r).sum() for f, r in zip(feats, R)).backward()
    return {k: p.grad.double() for k, p in P.items()}, R
g64, R = grads(torch.float64)
g32, _ = grads(torch.float32)
ga, ta = TU.default_inference_args(f_maps=f_maps, num_levels=levels)
s = TU.InferenceSession(ga, ta, "cuda:0", state_dict=sd); eng = s.engine
feats_d, tape = BW.backbone_forward_train(eng, eng.to_cl(x.cuda()), dims)
mine = BW.backbone_backward(eng, tape, [r[0].permute(1, 2, 3, 0).contiguous().float().cuda() for r in R])
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
for k in g64:
    print("%-62s torch32-vs-64 %.1e   hip-vs-64 %.1e" % (k[9:], rel(g32[k], g64[k]), rel(mine[k].reshape(g64[k].shape).cpu().double(), g64[k])))
