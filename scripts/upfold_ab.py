"""A/B of the conv_upfold workgroup forms (BFM_UPFOLD_WAVES=8 | 4): timing per shape and a checksum of the output, so that two
processes can be compared bit for bit.  python scripts/upfold_ab.py"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from brainfm_amd import _lib as L
dev = torch.device("cuda:0"); torch.manual_seed(0)
lib = L.load()
shapes = [((80, 80, 80), 128, 64), ((40, 80, 80), 128, 64), ((40, 40, 40), 128, 64), ((40, 40, 40), 256, 128), ((20, 20, 20), 256, 128),
          ((20, 20, 20), 512, 256), ((10, 10, 10), 512, 256), ((10, 10, 10), 1024, 512), ((5, 5, 5), 2048, 1024)]
for (d, h, w), cb, cout in shapes:
    B = torch.randn(d, h, w, cb, device=dev); scale = torch.rand(cb, device=dev) + 0.5
    shift = torch.randn(cb, device=dev) * 0.1; bound = torch.full((8,), 6.0, device=dev)
    wt = (torch.randn(cout, cb, 3, 3, 3, device=dev) * 0.05).contiguous()
    out = torch.empty(2 * d, 2 * h, 2 * w, cout, device=dev)
    wp = torch.empty(lib.bfm_pack_conv_weights_upfold_bytes(cb, cout, 3), dtype=torch.uint8, device=dev)
    wexp = C.c_int(0)
    L.check(lib.bfm_pack_conv_weights_upfold(L.ptr(wt), 0, cb, cout, float(wt.abs().max().item()), 3, L.ptr(wp), C.byref(wexp),
                                             L.stream_ptr()), "pack")
    wsb = lib.bfm_conv3x3x3_upfold_workspace(cb, d, h, w, cout)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)

    def run():
        L.check(lib.bfm_conv3x3x3_upfold_ex(L.ptr(B), cb, d, h, w, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8, L.ptr(wp),
                                            wexp.value, cout, 3, L.ptr(out), L.ptr(ws) if wsb else None, ws.numel(),
                                            L.stream_ptr()), "upfold")
    out.fill_(float("nan"))
    run(); torch.cuda.synchronize()
    digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    reps = 20
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nv = 8 * d * h * w
    print("waves %s  %dx%dx%d %4d -> %4d: %7.3f ms  %7.1f TFLOP/s algorithmic  sha1 %s  nan %d" % (
        os.environ.get("BFM_UPFOLD_WAVES", "8"), d, h, w, cb, cout, ms, 2.0 * 27 * cb * cout * nv / ms / 1e9, digest,
        int(torch.isnan(out).sum())), flush=True)
