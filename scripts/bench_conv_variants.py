"""One single-source conv layer, every variant (0 conv_mfma, 1 ws, 2 mfma16, 3 wino), passes 3 and 1.
usage: python scripts/bench_conv_variants.py [size=160] [cin=64] [cout=64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer

dev = torch.device("cuda:0")
torch.manual_seed(0)
D = H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
lib = L.load()
A = torch.randn(D, H, W, cin, device=dev)
scale = torch.rand(cin, device=dev) + 0.5
shift = torch.randn(cin, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
out = torch.empty(D, H, W, cout, device=dev)
ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
flops = 2.0 * 27 * cin * cout * D * H * W
for passes in (3, 1):
    sess = TU.InferenceSession(ga, ta, dev, passes=passes)
    eng = sess.engine
    ly = _Layer()
    ly.name, ly.cin, ly.cout, ly.groups = "bench", cin, cout, 8
    ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
    ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
    for ver in (0, 1, 2, 3, 4, 5):
        cfg = (C.c_int * 8)()
        L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan")
        cfg[6] = ver
        try:
            run = lambda: eng._conv_launch(ly, A, cin, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out, ws)
            run(); torch.cuda.synchronize()
        except L.BfmError as e:
            print("passes %d ver %d: %s" % (passes, ver, e)); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("passes %d ver %d  %8.1f us  %7.1f TF/s algorithmic" % (passes, ver, ms * 1e3, flops / ms / 1e9))
