"""Per-shape timing of the MFMA convolution for the layers of one tile (tuning harness, GPU box).
usage: python scripts/conv_shapes.py [tile=160] [reps=3] [passes=3]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brainfm_amd import _lib as L
from brainfm_amd.engine import UNetEngine

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
only = sys.argv[4] if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
fm = [64, 128, 256, 512, 1024, 2048]
shapes = []
d = tile
for i, co in enumerate(fm):
    if i > 0:
        d //= 2
    ci = 1 if i == 0 else fm[i - 1]
    c1 = max(co // 2, ci)
    shapes += [("enc%d.1" % i, ci, 0, c1, d, 0), ("enc%d.2" % i, c1, 0, co, d, 0)]
sizes = [tile // 2 ** i for i in range(6)]
for i in range(5):
    lo, hi = sizes[5 - i], sizes[4 - i]
    cs, cx = fm[4 - i], fm[5 - i]
    shapes += [("dec%d.1" % i, cs, cx, cs, hi, lo), ("dec%d.2" % i, cs, 0, cs, hi, 0)]

eng = UNetEngine.__new__(UNetEngine)
eng.lib = L.load(); eng.device = dev; eng.num_groups = 8; eng.passes = passes; eng.eps = 1e-5; eng.slope = 0.01
eng._up_cache = {}; eng._ws = None; eng._plan_cache = {}; eng._tuned = set(); eng.force_direct = False
tot_ms, tot_fl = 0.0, 0.0
print("%-8s %5s %5s %5s %4s | %-22s | %9s %9s %8s" % ("layer", "CA", "CB", "Cout", "D", "plan WMxWN box splitk", "ms", "TFLOP/s", "GB/s"))
for name, ca, cb, cout, dd, lo in shapes:
    if (ca + cb) % 16 or cout % 64:
        continue
    if only and name != only:
        continue
    g = torch.Generator(device="cpu").manual_seed(0)
    sd = {"x.groupnorm.weight": torch.ones(ca + cb), "x.groupnorm.bias": torch.zeros(ca + cb),
          "x.conv.weight": (torch.rand(cout, ca + cb, 3, 3, 3, generator=g) - .5) * 0.05}
    ly = eng._make_layer(sd, "x", ca + cb, cout)
    A = torch.randn(dd, dd, dd, ca, device=dev)
    B = torch.randn(lo, lo, lo, cb, device=dev) if cb else None
    eng.single_conv(ly, A, (dd,) * 3, B=B, lo_dims=(lo,) * 3 if cb else None)   # warm, packs
    eng.prof = []
    for _ in range(reps):
        eng.single_conv(ly, A, (dd,) * 3, B=B, lo_dims=(lo,) * 3 if cb else None)
    torch.cuda.synchronize()
    ms = min(a.elapsed_time(b) for a, b, _, _ in eng.prof)
    fl, by = eng.prof[0][2], eng.prof[0][3]
    eng.prof = None
    cfg = eng._plan(ca + cb, cout, (dd,) * 3)
    plan = "%dx%d (%d,%d,%d) k%d v%d" % (cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], cfg[6])
    print("%-8s %5d %5d %5d %4d | %-22s | %9.3f %9.1f %8.0f" % (name, ca, cb, cout, dd, plan, ms, fl / ms / 1e9, by / ms / 1e6))
    tot_ms += ms; tot_fl += fl
    del A, B, ly
print("total %.3f ms, %.1f TFLOP/s algorithmic (x%d in f16 MFMA)" % (tot_ms, tot_fl / tot_ms / 1e9, passes))
