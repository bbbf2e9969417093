"""What a masked Winograd launch costs when only some of its boxes hold input (profiles/r02_sparse_launch_micro.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer
dev = torch.device("cuda:0"); torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
lib = L.load()
for D in (160, 80):
    H = W = D
    A = torch.randn(D, H, W, 64, device=dev); scale = torch.rand(64, device=dev) + 0.5
    shift = torch.randn(64, device=dev) * 0.1; bound = torch.full((8,), 6.0, device=dev)
    out = torch.empty(D, H, W, 64, device=dev); ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
    ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "bench", 64, 64, 8
    ly.w_raw = (torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05).contiguous()
    ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
    cfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(64, 64, D, H, W, cfg), "plan"); cfg[6] = 3
    for frac in (0.0, 0.25, 1.0):
        img = (torch.rand(D // 4, H // 4, W // 16, device=dev) < frac).float()
        img = img.repeat_interleave(4, 0).repeat_interleave(4, 1).repeat_interleave(16, 2).contiguous()
        def run(): eng._conv_launch(ly, A, 64, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out, ws, mask_img=img)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print("masked %d^3 active fraction %.2f (random boxes): %.1f us" % (D, float(img.mean()), e0.elapsed_time(e1) * 100))
    for name, sl in (("first quarter of the slabs", (slice(0, D // 4), slice(None), slice(None))),
                     ("a centred cube of half the side", (slice(D // 4, 3 * D // 4),) * 3),
                     ("quarter of the rows of every slab", (slice(None), slice(0, H // 4), slice(None)))):
        img = torch.zeros(D, H, W, device=dev)
        img[sl] = 1.0
        def run(): eng._conv_launch(ly, A, 64, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out, ws, mask_img=img)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print("masked %d^3 %s (fraction %.3f): %.1f us" % (D, name, float(img.mean()), e0.elapsed_time(e1) * 100))
