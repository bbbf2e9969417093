"""Phase ablation of conv_wino4 on one layer (needs a -DBFM_W4_ABLATE build; results of the ablated launches are wrong).
   python tests/diag/diag_wino4_ablate.py [size=160] [cin=64] [cout=64] [reps=5] [abl list, comma separated]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
abls = [int(a) for a in (sys.argv[5] if len(sys.argv) > 5 else "0,1,2,3,4,8,12,15,16,31,28,64").split(",")]
NAMES = {1: "no weight loads", 2: "no LDS operand reads", 4: "no transform stage", 8: "no LDS-DMA", 16: "no epilogue", 64: "no MFMAs", 128: "epilogue without its global stores",
         256: "epilogue without the LDS exchange", 512: "epilogue stores into an 8 MB window (no HBM writes)"}
A = torch.randn(S, S, S, cin, device=dev)
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
scale = torch.rand(cin, device=dev) + 0.5
shift = torch.randn(cin, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
wp = torch.empty(lib.bfm_pack_conv_weights_wino4_bytes(cin, cout, 3), dtype=torch.uint8, device=dev)
wexp = C.c_int(0)
L.check(lib.bfm_pack_conv_weights_wino4(L.ptr(w), cin, cout, float(w.abs().max()), 3, L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
out = torch.empty(S, S, S, cout, device=dev)


nrows = lib.bfm_conv3x3x3_wino4_rows(S, S, S, 3)
rows = torch.empty(lib.bfm_moment_rows_bytes(nrows, cout), dtype=torch.uint8, device=dev)
WITH_ROWS = [False]


def go():
    L.check(lib.bfm_conv3x3x3_wino4(L.ptr(A), cin, S, S, S, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8, L.ptr(wp), wexp.value, cout,
                                    0.01, 3, 0, L.ptr(out), L.ptr(rows) if WITH_ROWS[0] else None, L.stream_ptr()), "wino4")


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


os.environ.pop("BFM_W4_ABL", None)
for _ in range(20):          # settle the clock
    go()
torch.cuda.synchronize()
new = out.clone()
os.environ["BFM_W4_OLD"] = "1"
out.fill_(float("nan"))
go()
torch.cuda.synchronize()
print("conv_wino4d vs conv_wino4 (lo halves rounded to nearest instead of truncated): equal %s, max |diff| / max|y| %.3e" % (bool(torch.equal(new, out)), float((new - out).abs().max() / out.abs().max())), flush=True)
print("conv_wino4 (round 3's kernel): %.3f ms" % timed(), flush=True)
os.environ.pop("BFM_W4_OLD", None)
print("conv_wino4d (shipped): %.3f ms" % timed(), flush=True)
WITH_ROWS[0] = True
print("conv_wino4d (shipped) with moment rows: %.3f ms" % timed(), flush=True)
WITH_ROWS[0] = False
print("conv_wino4d (shipped): %.3f ms" % timed(), flush=True)
for a in abls:
    os.environ["BFM_W4_ABL"] = str(a)
    what = " + ".join(v for k, v in NAMES.items() if a & k) or "everything kept (ablation build's copy)"
    print("ABL %3d: %.3f ms   %s" % (a, timed(), what), flush=True)
os.environ["BFM_W4_ABL"] = "0"
for sl in (0, 20):
    os.environ["BFM_W4_SLEEP"] = str(sl)
    print("first-round stagger of %2d kilocycles: %.3f ms" % (sl, timed()), flush=True)
os.environ.pop("BFM_W4_SLEEP", None)
os.environ.pop("BFM_W4_ABL", None)
print("shipped kernel again: %.3f ms" % timed(), flush=True)
