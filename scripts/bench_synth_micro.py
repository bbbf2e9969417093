"""Kernel-only times of the small synthesis kernels at 160^3 (20 launches in a replayed hipGraph between two HIP events):
the loop used while tuning them.  usage: python scripts/bench_synth_micro.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from brainfm_amd import _lib as L
from brainfm_amd import generator_utils as GU
from brainfm_amd import shapeid as SH

dev = torch.device("cuda:0")
N = 160
nv = N ** 3
lib = L.load()


def timed(fn, launches=20, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(launches):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (launches * replays) * 1e3


def show(name, us, nbytes):
    print("%-58s %8.2f us  %6.3f TB/s  %.3f of 8 TB/s" % (name, us, nbytes / us / 1e6, nbytes / us / 1e6 / 8))


vol = torch.rand(N, N, N, device=dev)
out = torch.empty_like(vol)
for sigma in (0.7, 1.5, 4.0):
    k = GU.make_gaussian_kernel(sigma, dev)
    for ax in range(3):
        show("conv1d_axis axis %d sigma %.1f (%d taps)" % (ax, sigma, k.numel()),
             timed(lambda: L.check(lib.bfm_conv1d_axis(L.ptr(vol), N, N, N, ax, L.ptr(k), k.numel(), L.ptr(out), L.stream_ptr()), "c")),
             nv * 8)
for shp in ((5, 5, 5), (6, 6, 6, 3), (80, 80, 27)):
    small = torch.rand(*shp, device=dev)
    f = np.array([N / shp[0], N / shp[1], N / shp[2]])
    c = shp[3] if len(shp) == 4 else 1
    show("zoom_linear %s -> 160^3 x %d" % (shp, c), timed(lambda: GU.myzoom_torch(small, f)), nv * 4 * c)
show("reduce max", timed(lambda: GU.reduce_dev(1, vol)), nv * 4)
show("reduce sum(x*y)", timed(lambda: GU.reduce_dev(3, vol, vol)), nv * 8)
show("randn_philox", timed(lambda: GU.draws.randn((N, N, N), dev)), nv * 4)
np.random.seed(0)
grads = torch.from_numpy(np.ascontiguousarray(SH.perlin_gradients((2, 2, 2), (True, False, False)))).to(dev)
noise = torch.empty((N, N, N), dtype=torch.float64, device=dev)
show("perlin3d", timed(lambda: L.check(lib.bfm_perlin3d(L.ptr(grads), N, N, N, 2, 2, 2, L.ptr(noise), L.stream_ptr()), "p")), nv * 8)
show("percentile_f64", timed(lambda: SH.percentile_dev(noise, 91.0), launches=5), nv * 8 * 7)
show("ew_unary gamma", timed(lambda: GU.ew_unary(L.EW_GAMMA, vol, 300.0, 1.1)), nv * 8)
show("ew_binary axpy_clamp0 (noise)", timed(lambda: GU.ew_binary(L.EW_AXPY_CLAMP0, vol, out, 0.3)), nv * 12)
