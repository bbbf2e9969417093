"""Phase ablation of conv_upfold on one decoder join (needs a -DBFM_UP_ABLATE build; ablated launches compute wrong results).
   python tests/diag/diag_upfold_ablate.py [low-res size=80] [cb=128] [cout=64] [ca=64] [reps=5] [abl list, comma separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ca = int(sys.argv[4]) if len(sys.argv) > 4 else 64
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
abls = [int(a) for a in (sys.argv[6] if len(sys.argv) > 6 else "0,1,2,4,8,12,16,64,3,7,23,68,71,87").split(",")]
NAMES = {1: "no weight loads", 2: "no LDS operand reads", 4: "no staging", 8: "staging without global loads", 16: "no epilogue",
         64: "no MFMAs"}
B = torch.randn(S, S, S, cb, device=dev)
w = (torch.randn(cout, ca + cb, 3, 3, 3, device=dev) * 0.05).contiguous()
scale = torch.rand(cb, device=dev) + 0.5
shift = torch.randn(cb, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
wp = torch.empty(lib.bfm_pack_conv_weights_upfold_bytes(cb, cout, 3), dtype=torch.uint8, device=dev)
wexp = C.c_int(0)
L.check(lib.bfm_pack_conv_weights_upfold(L.ptr(w), ca, cb, cout, float(w[:, ca:].abs().max()), 3, L.ptr(wp), C.byref(wexp),
                                         L.stream_ptr()), "pack")
out = torch.empty(2 * S, 2 * S, 2 * S, cout, device=dev)
nws = lib.bfm_conv3x3x3_upfold_workspace(cb, S, S, S, cout)
ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=dev)
print("low-res %d^3 x %d -> %d^3 x %d; split-K workspace %d bytes" % (S, cb, 2 * S, cout, nws), flush=True)


def go():
    L.check(lib.bfm_conv3x3x3_upfold_ex(L.ptr(B), cb, S, S, S, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8, L.ptr(wp), wexp.value,
                                        cout, 3, L.ptr(out), L.ptr(ws) if nws else None, ws.numel(), L.stream_ptr()), "upfold")


def timed():
    go(); go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


os.environ.pop("BFM_UP_ABL", None)
for _ in range(20):          # settle the clock
    go()
torch.cuda.synchronize()
t0 = timed()
fl = 2.0 * 27 * cb * cout * (2 * S) ** 3
print("conv_upfold (shipped): %.3f ms = %.0f TFLOP/s algorithmic (x 64/216 x 3 issued)" % (t0, fl / t0 * 1e-9), flush=True)
for a in abls:
    os.environ["BFM_UP_ABL"] = str(a)
    what = " + ".join(v for k, v in NAMES.items() if a & k) or "everything kept (ablation build's copy)"
    print("ABL %3d: %.3f ms   %s" % (a, timed(), what), flush=True)
os.environ["BFM_UP_ABL"] = "0"
for sl in (0, 2, 5, 10, 20, 40, 60):
    os.environ["BFM_UP_SLEEP"] = str(sl)
    print("first-round stagger of %2d kilocycles: %.3f ms" % (sl, timed()), flush=True)
os.environ.pop("BFM_UP_SLEEP", None)
os.environ.pop("BFM_UP_ABL", None)
print("shipped kernel again: %.3f ms" % timed(), flush=True)
