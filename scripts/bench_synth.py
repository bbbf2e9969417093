"""Synthesis / pre-processing kernels one at a time on 160^3 volumes against the HBM
roofline.  Each call is issued `reps` times back to back inside one HIP event pair through the host mirror (so a
number includes the mirror's torch.empty and table uploads where it has them); GB/s = the kernel's algorithmic bytes
(coordinates + one touch of the source + the output) / average duration.
usage: python scripts/bench_synth.py [reps]  -> one line per kernel (committed as profiles/r01_synth_kernels.txt)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from brainfm_amd import generator_utils as GU
from brainfm_amd import interpol as IP
from brainfm_amd import misc as MI
from brainfm_amd import shapeid as SH
from brainfm_amd import test_utils as TU
from brainfm_amd import _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
np.random.seed(0)
N = 160
nv = N ** 3
PEAK = 8000.0


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, nbytes):
    gbs = nbytes / ms / 1e6
    print("%-58s %9.1f us  %8.1f MB  %7.0f GB/s  %5.1f %% of 8 TB/s" % (name, ms * 1e3, nbytes / 1e6, gbs, 100 * gbs / PEAK))


vol = torch.rand(N, N, N, device=dev)
src = torch.rand(200, 200, 200, device=dev)
ii = (torch.rand(N, N, N, device=dev) * 198).contiguous()
jj = (torch.rand(N, N, N, device=dev) * 198).contiguous()
kk = (torch.rand(N, N, N, device=dev) * 198).contiguous()
# smooth coordinates (what a deformation field produces): identity + small perturbation
zz, yy, xx = torch.meshgrid(*[torch.arange(N, device=dev, dtype=torch.float32)] * 3, indexing="ij")
si, sj, sk = (zz * 1.2 + 3).contiguous(), (yy * 1.2 + 3).contiguous(), (xx * 1.2 + 3).contiguous()
report("fast_3D_interp linear, smooth coords 200^3 -> 160^3", timeit(lambda: GU.fast_3D_interp_torch(src, si, sj, sk, "linear")), nv * 20)
report("fast_3D_interp linear, random coords", timeit(lambda: GU.fast_3D_interp_torch(src, ii, jj, kk, "linear")), nv * 20)
lab = (torch.rand(200, 200, 200, device=dev) * 30).to(torch.int32)
report("fast_3D_interp nearest (labels), smooth coords", timeit(lambda: GU.fast_3D_interp_torch(lab, si, sj, sk, "nearest")), nv * 20)
small = torch.rand(10, 10, 10, 3, device=dev)
report("myzoom_torch 10^3x3 -> 160^3x3 (deformation field)", timeit(lambda: GU.myzoom_torch(small, 16.0)), nv * 12)
report("gaussian_blur_3d sigma 1.5 (3 axis passes)", timeit(lambda: GU.gaussian_blur_3d(vol, [1.5, 1.5, 1.5], dev)), nv * 24)
report("elementwise gamma (ew_unary)", timeit(lambda: GU.ew_unary(L.EW_GAMMA, vol, 300.0, 1.1)), nv * 8)
report("bias field multiply-exp (ew_binary)", timeit(lambda: GU.ew_binary(L.EW_MUL_EXP, vol, vol)), nv * 12)
report("reduction max (bfm_reduce_f32)", timeit(lambda: GU.tensor_max(vol)), nv * 4)
grid = torch.stack([si, sj, sk], -1)[None].contiguous()
src5 = src[None, None].contiguous()
report("interpol.grid_pull linear, 1 ch", timeit(lambda: IP.grid_pull(src5, grid, 1, "dct2", True)), nv * 20)
v5 = vol[None, None].contiguous()
g160 = (torch.stack([zz, yy, xx], -1) + 0.37)[None].contiguous()
report("interpol.grid_push linear, 1 ch (atomics)", timeit(lambda: IP.grid_push(v5, g160, [N, N, N], 1, "dct2", True)), nv * 24)
report("interpol.grid_grad linear, 1 ch", timeit(lambda: IP.grid_grad(v5, g160, 1, "dct2", True)), nv * 28)
report("bspline3 prefilter, 3 axes (spline_coeff_nd)", timeit(lambda: IP.spline_coeff_nd(vol, 3, "dct2", 3)), nv * 24)
low = torch.rand(80, 80, 53, device=dev)
report("interpol.resize cubic 80x80x53 -> 160^3 (prefilter + 3 passes)",
       timeit(lambda: IP.resize(low, shape=[N, N, N], anchor="edge", interpolation=3, bound="dct2", prefilter=True)), nv * 4 * 2)
aff = np.array([[0., -1.2, 0., 30.], [1.0, 0., 0., -20.], [0., 0., 2.0, 5.], [0., 0., 0., 1.]])
report("align_volume_to_ref (permute + flip gather)", timeit(lambda: MI.align_volume_to_ref(vol, aff, np.eye(4))), nv * 8)
report("torch_resize 1.2x1.0x2.0 mm -> 1 mm (blur + aniso zoom)", timeit(lambda: MI.torch_resize(vol, aff, 1.0)), nv * 4 * (1 + 2.4))
report("zero_crop bounding box (bbox reduction + slice)", timeit(lambda: TU.zero_crop(vol)), nv * 4)
# ShapeID
shp = (160, 160, 160)
report("generate_perlin_noise_3d res 2 (fp64 out)", timeit(lambda: SH.generate_perlin_noise_3d(shp, (2, 2, 2))), nv * 8)
Vx, Vy, Vz = (torch.rand(N, N, N, device=dev) for _ in range(3))
Cc = torch.rand(N, N, N, device=dev)
dC = torch.empty_like(Cc)
lib = L.load()
try:
    report("advect_upwind_rhs (AdvDiffPDE.forward body, fp32 state)",
           timeit(lambda: L.check(lib.bfm_advect_upwind_rhs(L.ptr(Cc), 0, L.ptr(Vx), L.ptr(Vy), L.ptr(Vz), N, N, N, 1,
                                                            L.ptr(dC), L.stream_ptr()), "rhs")), nv * 20)
except Exception as e:
    print("advect_upwind_rhs: skipped (%s)" % str(e)[:100])
