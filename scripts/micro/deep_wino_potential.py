"""What would Winograd buy on the deep single-source layers if the Winograd kernels took a batch of same-shape tiles?
A batch of S samples is emulated by ONE volume of S x the voxels (samples stacked along z: the boundary effects do not
matter for a timing), every variant timed on it: 0 conv_mfma, 2 conv_mfma16 (what the batched deep levels run today),
3 Winograd F(2,3), 4 Winograd F(4,3).     python scripts/micro/deep_wino_potential.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer

dev = torch.device("cuda:0")
torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
lib = L.load()
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
cases = [("enc3.1 x1 (20^3)", 256, 256, (20, 20, 20)), ("enc3.2 x1 (20^3)", 256, 512, (20, 20, 20)),
         ("dec1.2 x1 (20^3)", 512, 512, (20, 20, 20)),
         ("enc3.2 x8 (8 x 10^3)", 256, 512, (80, 10, 10)), ("dec1.2 x8 (8 x 10^3)", 512, 512, (80, 10, 10)),
         ("dec1.2 x4 (4 x 20x10x10)", 512, 512, (80, 10, 10)),
         ("enc4.2 x1 (10^3)", 512, 1024, (10, 10, 10)), ("dec0.2 x1 (10^3)", 1024, 1024, (10, 10, 10)),
         ("dec0.2 x8 (8 x 5^3)", 1024, 1024, (40, 5, 5)), ("enc2.2 x8 (8 x 20^3)", 128, 256, (160, 20, 20))]
for name, cin, cout, dims in cases:
    D, H, W = dims
    A = torch.randn(D, H, W, cin, device=dev)
    scale = torch.rand(cin, device=dev) + 0.5
    shift = torch.randn(cin, device=dev) * 0.1
    bound = torch.full((8,), 6.0, device=dev)
    out = torch.empty(D, H, W, cout, device=dev)
    ly = _Layer()
    ly.name, ly.cin, ly.cout, ly.groups = "bench", cin, cout, 8
    ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
    ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
    flops = 2.0 * 27 * cin * cout * D * H * W
    res = []
    for ver in (0, 2, 3, 4):
        cfg = (C.c_int * 8)()
        L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan")
        cfg[6] = ver
        try:
            run = lambda: eng._conv_launch(ly, A, cin, None, 0, dims, None, scale, shift, bound, 8, cfg, out, ws)
            run()
            torch.cuda.synchronize()
        except L.BfmError as e:
            res.append("v%d: %s" % (ver, str(e)[:30]))
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append("v%d %6.1f us %5.0f TF/s (splitk %d)" % (ver, ms * 1e3, flops / ms / 1e9, cfg[5]))
    print("%-28s %4d->%4d %-12s | %s" % (name, cin, cout, "x".join(map(str, dims)), " | ".join(res)))
