"""One conv launch sequence for rocprofv3 --pmc: python scripts/run_one_conv.py <ver> [size cin cout reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer
ver = int(sys.argv[1]); D = H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 64; cout = int(sys.argv[4]) if len(sys.argv) > 4 else 64
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0"); torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
lib = L.load()
A = torch.randn(D, H, W, cin, device=dev); scale = torch.rand(cin, device=dev) + 0.5
shift = torch.randn(cin, device=dev) * 0.1; bound = torch.full((8,), 6.0, device=dev)
out = torch.empty(D, H, W, cout, device=dev); ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "bench", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
cfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan"); cfg[6] = ver
eng._conv_launch(ly, A, cin, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out, ws)      # warm (packs weights)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng._conv_launch(ly, A, cin, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out, ws)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("ver %d %dx%dx%d %d->%d: %.3f ms  %.1f TFLOP/s algorithmic" % (ver, D, H, W, cin, cout, ms, 2.0 * 27 * cin * cout * D * H * W / ms / 1e9))
