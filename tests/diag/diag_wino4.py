"""Winograd F(4,3) kernel (bfm_conv3x3x3_wino4) against a float64 convolution (small shapes) and against conv_wino's time.
   python tests/diag/diag_wino4.py [size=160] [cin=64] [cout=64] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
import torch.nn.functional as F
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def run_w4(A, w, scale, shift, bound, slope, accumulate=None, rows=False):
    D, H, W, cin = A.shape
    cout = w.shape[0]
    nb = lib.bfm_pack_conv_weights_wino4_bytes(cin, cout, 3)
    wp = torch.empty(nb, dtype=torch.uint8, device=dev)
    wexp = C.c_int(0)
    L.check(lib.bfm_pack_conv_weights_wino4(L.ptr(w), cin, cout, float(w.abs().max()), 3, L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
    out = accumulate.clone() if accumulate is not None else torch.full((D, H, W, cout), float("nan"), device=dev)
    mr = None
    if rows:
        n = lib.bfm_conv3x3x3_wino4_rows(D, H, W, 3)
        mr = torch.zeros(lib.bfm_moment_rows_bytes(n, cout), dtype=torch.uint8, device=dev)
    def go():
        L.check(lib.bfm_conv3x3x3_wino4(L.ptr(A), cin, D, H, W, L.ptr(scale), L.ptr(shift), L.ptr(bound), bound.numel(), L.ptr(wp),
                                        wexp.value, cout, slope, 3, 1 if accumulate is not None else 0, L.ptr(out),
                                        L.ptr(mr) if mr is not None else None, L.stream_ptr()), "wino4")
    go()
    torch.cuda.synchronize()
    return out, mr, go


def ref64(A, w, scale, shift, slope, accumulate=None):
    x = (A.double().cpu() * scale.double().cpu() + shift.double().cpu()).permute(3, 0, 1, 2)[None]
    y = F.conv3d(x, w.double().cpu(), padding=1)[0].permute(1, 2, 3, 0)
    if accumulate is not None:
        y = y + accumulate.double().cpu()
    return torch.where(y >= 0, y, y * slope)


for dims, cin, cout in (((8, 8, 16), 16, 64), ((9, 11, 21), 32, 64), ((12, 8, 30), 64, 128), ((5, 9, 20), 32, 64), ((20, 20, 20), 32, 64),
                         ((17, 24, 7), 16, 128)):
    A = torch.randn(*dims, cin, device=dev)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
    scale = torch.rand(cin, device=dev) + 0.5
    shift = torch.randn(cin, device=dev) * 0.1
    bound = torch.full((8,), float((A.abs().amax((0, 1, 2)) * scale + shift.abs()).max()), device=dev)
    for acc in (None, torch.randn(*dims, cout, device=dev)):
        out, mr, _ = run_w4(A, w, scale, shift, bound, 0.01, accumulate=acc, rows=True)
        want = ref64(A, w, scale, shift, 0.01, accumulate=acc)
        err = float((out.double().cpu() - want).abs().max() / want.abs().max())
        n = mr.numel() // (cout * 24)
        k = n * cout
        rs = mr[:k * 8].view(torch.float64).view(n, cout).sum(0).cpu()
        rmx = mr[k * 20:k * 24].view(torch.float32).view(n, cout).max(0)[0].cpu()
        e_s = float((rs - out.double().cpu().reshape(-1, cout).sum(0)).abs().max())
        print(dims, cin, cout, "accumulate" if acc is not None else "plain", "max rel err %.2e" % err, "rows: sum err %.1e max equal %s"
              % (e_s, bool(torch.equal(rmx, out.cpu().reshape(-1, cout).max(0)[0]))), flush=True)

size = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
D = H = W = size
A = torch.randn(D, H, W, cin, device=dev)
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
scale = torch.rand(cin, device=dev) + 0.5
shift = torch.randn(cin, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
out4, _, go = run_w4(A, w, scale, shift, bound, 0.01, rows=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    go()
e1.record(); torch.cuda.synchronize()
ms4 = e0.elapsed_time(e1) / reps
# conv_wino (F(2,3)) on the same operands through the engine
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "bench", cin, cout, 8
ly.w_raw = w; ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
cfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan"); cfg[6] = 3
out2 = torch.empty(D, H, W, cout, device=dev); ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
eng._conv_launch(ly, A, cin, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out2, ws)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    eng._conv_launch(ly, A, cin, None, 0, (D, H, W), None, scale, shift, bound, 8, cfg, out2, ws)
e1.record(); torch.cuda.synchronize()
ms2 = e0.elapsed_time(e1) / reps
fl = 2.0 * 27 * cin * cout * D * H * W
print("%d^3 %d->%d: F(4,3) %.3f ms (%.0f TFLOP/s alg)   F(2,3) %.3f ms (%.0f)   max |diff| / max %.2e"
      % (size, cin, cout, ms4, fl / ms4 / 1e9, ms2, fl / ms2 / 1e9, float((out4 - out2).abs().max() / out2.abs().max())))
