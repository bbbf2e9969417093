"""conv_updiff (three-product up-folded conv) against conv_upfold and a float64 reference; timing of both.
   python tests/diag/diag_updiff.py [lowres=80] [cb=128] [cout=64] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import torch.nn.functional as F
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(kind, Bt, w, ca, scale, shift, bound):
    d, h, ww, cb = Bt.shape
    cout = w.shape[0]
    if kind == "updiff":
        nb = lib.bfm_pack_conv_weights_updiff_bytes(cb, cout, 3)
        wp = torch.empty(nb, dtype=torch.uint8, device=dev); wexp = C.c_int(0)
        L.check(lib.bfm_pack_conv_weights_updiff(L.ptr(w), ca, cb, cout, float(w[:, ca:].abs().max()), 3, L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
        out = torch.full((2 * d, 2 * h, 2 * ww, cout), float("nan"), device=dev)
        go = lambda: L.check(lib.bfm_conv3x3x3_updiff(L.ptr(Bt), cb, d, h, ww, L.ptr(scale), L.ptr(shift), L.ptr(bound), bound.numel(), L.ptr(wp), wexp.value, cout, 3, L.ptr(out), L.stream_ptr()), "updiff")
    else:
        nb = lib.bfm_pack_conv_weights_upfold_bytes(cb, cout, 3)
        wp = torch.empty(nb, dtype=torch.uint8, device=dev); wexp = C.c_int(0)
        L.check(lib.bfm_pack_conv_weights_upfold(L.ptr(w), ca, cb, cout, float(w[:, ca:].abs().max()), 3, L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
        out = torch.full((2 * d, 2 * h, 2 * ww, cout), float("nan"), device=dev)
        go = lambda: L.check(lib.bfm_conv3x3x3_upfold_ex(L.ptr(Bt), cb, d, h, ww, L.ptr(scale), L.ptr(shift), L.ptr(bound), bound.numel(), L.ptr(wp), wexp.value, cout, 3, L.ptr(out), None, 0, L.stream_ptr()), "upfold")
    go(); torch.cuda.synchronize()
    return out, go

def ref64(Bt, w, ca, scale, shift):
    x = (Bt.double().cpu() * scale.double().cpu() + shift.double().cpu()).permute(3, 0, 1, 2)[None]
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv3d(x, w[:, ca:].double().cpu(), padding=1)[0].permute(1, 2, 3, 0)

for dims, ca, cb, cout in (((4, 4, 4), 16, 16, 64), ((5, 7, 9), 32, 32, 64), ((6, 4, 11), 16, 48, 128)):
    Bt = torch.randn(*dims, cb, device=dev)
    w = (torch.randn(cout, ca + cb, 3, 3, 3, device=dev) * 0.05).contiguous()
    scale = torch.rand(cb, device=dev) + 0.5; shift = torch.randn(cb, device=dev) * 0.1
    bound = torch.full((8,), float((Bt.abs().amax((0, 1, 2)) * scale + shift.abs()).max()), device=dev)
    want = ref64(Bt, w, ca, scale, shift)
    for kind in ("upfold", "updiff"):
        if kind == "updiff" and not lib.bfm_conv3x3x3_updiff_ok(cb, *dims, cout, 3):
            print(dims, "updiff cannot run"); continue
        out, _ = run(kind, Bt, w, ca, scale, shift, bound)
        print(dims, ca, cb, cout, kind, "max rel err %.2e" % float((out.double().cpu() - want).abs().max() / want.abs().max()), flush=True)

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ca = cout
Bt = torch.randn(lo, lo, lo, cb, device=dev)
w = (torch.randn(cout, ca + cb, 3, 3, 3, device=dev) * 0.05).contiguous()
scale = torch.rand(cb, device=dev) + 0.5; shift = torch.randn(cb, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
res = {}
for kind in ("upfold", "updiff"):
    out, go = run(kind, Bt, w, ca, scale, shift, bound)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    res[kind] = (e0.elapsed_time(e1) / reps, out)
fl = 2.0 * 27 * cb * cout * (2 * lo) ** 3
print("%d^3 -> %d^3, %d -> %d: upfold %.3f ms (%.0f TFLOP/s alg)   updiff %.3f ms (%.0f)   max |diff| / max %.2e"
      % (lo, 2 * lo, cb, cout, res["upfold"][0], fl / res["upfold"][0] / 1e9, res["updiff"][0], fl / res["updiff"][0] / 1e9,
         float((res["upfold"][1] - res["updiff"][1]).abs().max() / res["upfold"][1].abs().max())))
