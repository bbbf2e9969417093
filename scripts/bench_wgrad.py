"""Weight-gradient kernels alone on the decoder joins of the full-width net (skip + 2x-upsampled channels):
bfm_conv3x3x3_wgrad_ex timed with HIP events, against float64 on a small case first.

    python scripts/bench_wgrad.py [size=128] [reps=5]          BFM_WGRAD_UPFOLD=0: the 27-tap kernel on all channels
                                                               BFM_WGRAD_WS=0: without the wave-specialised kernel
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brainfm_amd import _lib as L  # noqa: E402
from brainfm_amd.engine import nearest_index_map  # noqa: E402


def run(lib, dev, ca, cb, cout, dims, reps, check=False):
    D, H, W = dims
    lo = (D // 2, H // 2, W // 2)
    g = torch.Generator().manual_seed(ca + cb)
    A = torch.randn(dims + (ca,), generator=g).to(dev)
    B = torch.randn(lo + (cb,), generator=g).to(dev)
    dP = (torch.randn(dims + (cout,), generator=g) * 0.01).to(dev)
    cin = ca + cb
    scale = (torch.rand(cin, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cin, generator=g) * 0.1).to(dev)
    maps = [nearest_index_map(lo[a], dims[a]) for a in range(3)]
    reps_ = [np.bincount(maps[a], minlength=lo[a]).astype(np.int32) for a in range(3)]
    tens = [torch.from_numpy(m).to(dev) for m in maps + reps_]
    up = L.Upsample(lo[0], lo[1], lo[2], *[t.data_ptr() for t in tens])
    bnd = dP.abs().max().reshape(1)
    U = B[torch.from_numpy(maps[0]).long().to(dev)][:, torch.from_numpy(maps[1]).long().to(dev)][:, :, torch.from_numpy(maps[2]).long().to(dev)]
    X = torch.cat([A, U], dim=-1) * scale + shift
    xb = X.abs().max().reshape(1)
    wsb = lib.bfm_conv3x3x3_wgrad_workspace(cin, cout, D, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dW = torch.empty((cout, cin, 27), dtype=torch.float32, device=dev)
    st = L.stream_ptr()

    def call():
        L.check(lib.bfm_conv3x3x3_wgrad_ex(L.ptr(dP), cout, L.ptr(A), ca, L.ptr(B), cb, D, H, W, C.byref(up), L.ptr(scale),
                                           L.ptr(shift), L.ptr(bnd), L.ptr(xb), 1, 3, L.ptr(dW), L.ptr(ws), ws.numel(), st),
                "wgrad")
    call()
    torch.cuda.synchronize()
    if check:
        Xp = torch.nn.functional.pad(X.double().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 1, 1))[0]       # [cin][D+2][H+2][W+2]
        d64 = dP.double().reshape(-1, cout)
        ref = torch.empty((cout, cin, 27), dtype=torch.float64, device=dev)
        for t in range(27):
            kd, kh, kw = t // 9, (t // 3) % 3, t % 3
            win = Xp[:, kd:kd + D, kh:kh + H, kw:kw + W].reshape(cin, -1)
            ref[:, :, t] = (win @ d64).t()
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        print("  check %s ca %d cb %d cout %d: max rel err %.2e" % (dims, ca, cb, cout, err))
        assert err < 2e-5, err
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = 2.0 * 27 * cin * cout * D * H * W / 1e9
    print("  %3d + %4d -> %3d at %s: %.3f ms = %.0f TFLOP/s (27-tap flops)" % (ca, cb, cout, dims, ms, gf / ms))
    return ms


def run_plain(lib, dev, cin, cout, dims, reps):
    D, H, W = dims
    g = torch.Generator().manual_seed(cin)
    A = torch.randn(dims + (cin,), generator=g).to(dev)
    dP = (torch.randn(dims + (cout,), generator=g) * 0.01).to(dev)
    scale, shift = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    bnd, xb = dP.abs().max().reshape(1), A.abs().max().reshape(1)
    ws = torch.empty(lib.bfm_conv3x3x3_wgrad_workspace(cin, cout, D, H, W), dtype=torch.uint8, device=dev)
    dW = torch.empty((cout, cin, 27), dtype=torch.float32, device=dev)
    st = L.stream_ptr()

    def call():
        L.check(lib.bfm_conv3x3x3_wgrad_ex(L.ptr(dP), cout, L.ptr(A), cin, None, 0, D, H, W, None, L.ptr(scale), L.ptr(shift),
                                           L.ptr(bnd), L.ptr(xb), 1, 3, L.ptr(dW), L.ptr(ws), ws.numel(), st), "wgrad")
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("  %4d -> %3d at %s: %.3f ms = %.0f TFLOP/s" % (cin, cout, dims, ms, 2.0 * 27 * cin * cout * D * H * W / 1e9 / ms))
    return ms


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = L.load()
    print("BFM_WGRAD_UPFOLD=%s" % os.environ.get("BFM_WGRAD_UPFOLD", "1"))
    run(lib, dev, 32, 64, 64, (8, 16, 32), 1, check=True)
    run(lib, dev, 32, 32, 64, (8, 12, 40), 1, check=True)             # not a multiple of the folded tile: 27-tap kernel
    run(lib, dev, 64, 32, 128, (6, 8, 96), 1, check=True)
    tot = 0.0
    for ca, cb, cout, div in ((64, 128, 64, 1), (128, 256, 128, 2), (256, 512, 256, 4), (512, 1024, 512, 8)):
        tot += run(lib, dev, ca, cb, cout, (n // div,) * 3, reps)
    print("decoder joins together: %.2f ms" % tot)
    tot = 0.0
    for cin, cout, div in ((32, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 512, 8),
                           (512, 512, 8)):
        tot += run_plain(lib, dev, cin, cout, (n // div,) * 3, reps)
    print("plain layers together: %.2f ms" % tot)


if __name__ == "__main__":
    main()
