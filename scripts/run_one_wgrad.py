"""One weight-gradient launch sequence for rocprofv3 --pmc / timing: python scripts/run_one_wgrad.py <passes 0|3> [size cin cout reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brainfm_amd import _lib as L
passes = int(sys.argv[1]); D = H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 64; cout = int(sys.argv[4]) if len(sys.argv) > 4 else 64
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0"); torch.manual_seed(0)
lib = L.load()
A = torch.randn(D, H, W, cin, device=dev); dP = torch.randn(D, H, W, cout, device=dev) * 1e-3
scale = torch.rand(cin, device=dev) + 0.5; shift = torch.randn(cin, device=dev) * 0.1
bnd = dP.abs().max().reshape(1); xb = torch.full((8,), 6.0, device=dev)
ws = torch.empty(lib.bfm_conv3x3x3_wgrad_workspace(cin, cout, D, H, W), dtype=torch.uint8, device=dev)
dW = torch.empty(cout, cin, 27, device=dev)
def run():
    L.check(lib.bfm_conv3x3x3_wgrad_ex(L.ptr(dP), cout, L.ptr(A), cin, None, 0, D, H, W, None, L.ptr(scale), L.ptr(shift),
                                       L.ptr(bnd), L.ptr(xb), 8, passes, L.ptr(dW), L.ptr(ws), ws.numel(), L.stream_ptr()), "wgrad")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("wgrad passes %d %d^3 %d->%d: %.3f ms  %.1f TFLOP/s algorithmic" % (passes, D, cin, cout, ms, 2.0 * 27 * cin * cout * D * H * W / ms / 1e9))
