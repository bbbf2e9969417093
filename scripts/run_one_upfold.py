"""One conv_upfold launch sequence for rocprofv3 --pmc: python scripts/run_one_upfold.py [lo cb cout reps]
(default: the last decoder's low-res half, 128 -> 64 channels from 80^3 to 160^3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from brainfm_amd import _lib as L
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0"); torch.manual_seed(0)
lib = L.load()
B = torch.randn(lo, lo, lo, cb, device=dev); scale = torch.rand(cb, device=dev) + 0.5
shift = torch.randn(cb, device=dev) * 0.1; bound = torch.full((8,), 6.0, device=dev)
w = (torch.randn(cout, cb, 3, 3, 3, device=dev) * 0.05).contiguous()
out = torch.empty(2 * lo, 2 * lo, 2 * lo, cout, device=dev)
wp = torch.empty(lib.bfm_pack_conv_weights_upfold_bytes(cb, cout, 3), dtype=torch.uint8, device=dev)
wexp = C.c_int(0)
L.check(lib.bfm_pack_conv_weights_upfold(L.ptr(w), 0, cb, cout, float(w.abs().max().item()), 3, L.ptr(wp), C.byref(wexp),
                                         L.stream_ptr()), "pack")
wsb = lib.bfm_conv3x3x3_upfold_workspace(cb, lo, lo, lo, cout)
ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)


def run():
    L.check(lib.bfm_conv3x3x3_upfold_ex(L.ptr(B), cb, lo, lo, lo, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8, L.ptr(wp),
                                        wexp.value, cout, 3, L.ptr(out), L.ptr(ws) if wsb else None, ws.numel(),
                                        L.stream_ptr()), "upfold")


run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nv = (2 * lo) ** 3
print("conv_upfold %d^3 -> %d^3, %d -> %d: %.3f ms  %.1f TFLOP/s algorithmic (27 taps), %.1f issued f16 (8 taps x 3 passes)" % (
    lo, 2 * lo, cb, cout, ms, 2.0 * 27 * cb * cout * nv / ms / 1e9, 2.0 * 8 * 3 * cb * cout * nv / ms / 1e9))
