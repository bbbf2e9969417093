"""HISTORY.md section 3.3, one more cut (round 4): is it OUR gather, or anything with L1-cached data-dependent loads, that
reads wrong values while the library's conv_mfma (weights into LDS by LDS-DMA) runs on another stream?  The gather here
is torch's own index kernel (torch.take: ordinary global loads, no code of this library), on two streams, beside six
conv_mfma launches on a third; every result is compared with a serial run.  The library-independent pair of kernels
(scripts/micro/l1_ldsdma_hazard.hip: a checking gather beside pure LDS-DMA / LDS-DMA + MFMA co-runners) showed nothing.
    python tests/diag/diag_torch_gather_beside_conv.py [rounds=200]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brainfm_amd import _lib as L
from brainfm_amd import test_utils as TU
from brainfm_amd.engine import _Layer

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
lib = L.load()
g = torch.Generator().manual_seed(0)
X = torch.rand(256 ** 3, generator=g).to(dev)
n = 160 * 160 * 80
ax = torch.arange(n, dtype=torch.float32)
# a smooth, rotated path through the volume: neighbouring elements re-use lines
idx = ((40 + (ax // (160 * 80)) * 0.98).long() * 256 * 256 + (30 + ((ax // 80) % 160) * 0.97).long() * 256
       + (50 + (ax % 80) * 0.99).long()).to(dev)
idxs = [idx, (idx + 256 * 256 + 1).contiguous()]
want = [torch.take(X, i) for i in idxs]
torch.cuda.synchronize()

ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
cin = cout = 128
cd = (40, 40, 40)
cA = torch.randn(*cd, cin, device=dev)
cscale, cshift, cbound = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1, torch.full((8,), 6.0, device=dev)
cout_t, cws = torch.empty(*cd, cout, device=dev), torch.empty(1 << 26, dtype=torch.uint8, device=dev)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "corunner", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
ccfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, cd[0], cd[1], cd[2], ccfg), "plan")


def conv_beside(ver, k=6):
    ccfg[6] = ver
    for _ in range(k):
        eng._conv_launch(ly, cA, cin, None, 0, cd, None, cscale, cshift, cbound, 8, ccfg, cout_t, cws)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
side = torch.cuda.Stream()
for name, ver in (("no co-runner", None), ("conv_mfma (LDS-DMA weight ring)", 0), ("conv_wino (weights L2 -> VGPR)", 3)):
    if ver is not None:
        conv_beside(ver, 1)
    torch.cuda.synchronize()
    bad = 0
    for r in range(rounds):
        if ver is not None:
            with torch.cuda.stream(side):
                conv_beside(ver)
        outs = []
        for lane in range(2):
            with torch.cuda.stream(streams[lane]):
                outs.append([torch.take(X, idxs[lane]) for _ in range(3)])
        torch.cuda.synchronize()
        for lane in range(2):
            for o in outs[lane]:
                bad += int((o != want[lane]).sum())
    print("torch.take beside %-34s wrong elements: %d of %d" % (name + ":", bad, rounds * 6 * n))
