"""Every synthesis kernel of bench.py's `synthesis` block, eagerly, each launch on its own buffers (cold: the sets of one
kernel add up to >= 1.2 GB, far more than the 256 MB Infinity Cache), then two generator items (plain, and with the
file-based lesion map that runs the dopri5 chain).  Meant to run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
(scripts/pmc_synth.sh).  Sections are delimited in the dispatch stream by a marker kernel nothing else here launches
(bfm_bbox_nonzero on a 2^3 volume: `bbox_nonzero_kernel`): the dispatches between the (2k+1)-th and the (2k+2)-th marker
are section k; the section names and call counts go to <out> as JSON, in order."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import torch

import config5_lib as C5
from brainfm_amd import _lib as L
from brainfm_amd import generator as G
from brainfm_amd import generator_utils as GU
from brainfm_amd import shapeid as SH

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/synth_sections.json"
dev = torch.device("cuda:0")
N = 160
nv = N ** 3
lib = L.load()
np.random.seed(0)
torch.manual_seed(0)
rnd = lambda *shape: torch.rand(*shape, device=dev)
ax = torch.arange(N, device=dev, dtype=torch.float32)
zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
c_i, c_j, c_k = (zz * 1.1 + 5 + 0.3 * torch.sin(yy / 9)).contiguous(), (yy * 1.1 + 6).contiguous(), (xx * 1.1 + 4).contiguous()
_mk_src, _mk_box = torch.ones(2, 2, 2, device=dev), torch.zeros(6, dtype=torch.int32, device=dev)


def marker():
    L.check(lib.bfm_bbox_nonzero(L.ptr(_mk_src), 2, 2, 2, 0.0, L.ptr(_mk_box), L.stream_ptr()), "marker")


grads = SH.perlin_gradients((2, 2, 2), (True, False, False))
gdev = torch.from_numpy(np.ascontiguousarray(grads, dtype=np.float64)).to(dev)


def perlin(buf):
    L.check(lib.bfm_perlin3d(L.ptr(gdev), N, N, N, 2, 2, 2, L.ptr(buf), L.stream_ptr()), "perlin3d")
    return buf


CASES = [
    ("interp_linear", lambda: (rnd(192, 192, 192), c_i.clone(), c_j.clone(), c_k.clone()),
     lambda b: GU.fast_3D_interp_torch(b[0], b[1], b[2], b[3]), 4 * (192 ** 3 + 4 * nv), nv * 20),
    ("zoom_linear_field", lambda: rnd(6, 6, 6, 3), lambda b: GU.myzoom_torch(b, N / 6.0), 12 * nv, nv * 12),
    ("zoom_linear_bias", lambda: rnd(5, 5, 5), lambda b: GU.myzoom_torch(b, N / 5.0), 4 * nv, nv * 4),
    ("gaussian_blur_3d", lambda: rnd(N, N, N), lambda b: GU.gaussian_blur_3d(b, [1.5, 1.5, 1.5], dev), 8 * nv, nv * 24),
    ("ew_unary_gamma", lambda: rnd(N, N, N), lambda b: GU.ew_unary(L.EW_GAMMA, b, 300.0, 1.1), 8 * nv, nv * 8),
    ("ew_binary_mul_exp", lambda: (rnd(N, N, N), rnd(N, N, N)), lambda b: GU.ew_binary(L.EW_MUL_EXP, b[0], b[1]), 12 * nv, nv * 12),
    ("reduce_max", lambda: rnd(N, N, N), lambda b: GU.reduce_dev(1, b), 4 * nv, nv * 4),
    ("randn_philox", lambda: None, lambda b: GU.draws.randn((N, N, N), dev), 4 * nv, nv * 4),
    ("perlin3d", lambda: torch.empty((N, N, N), dtype=torch.float64, device=dev), perlin, 8 * nv, nv * 8),
    ("percentile_f64", lambda: perlin(torch.empty((N, N, N), dtype=torch.float64, device=dev)), lambda b: SH.percentile_dev(b, 91.0),
     8 * nv, nv * 8 * 7),
]
sections = []
for name, make, call, touched, alg in CASES:
    k = int(min(96, max(8, -(-1.2e9 // touched))))
    pool = [make() for _ in range(k)]
    keep = [call(pool[0])]                                   # warm (code objects, workspaces)
    torch.cuda.synchronize()
    sections.append({"section": name, "calls": k, "algorithmic_bytes_per_call": alg, "begin": True})
    marker()
    keep = [call(pool[j]) for j in range(k)]
    marker()
    torch.cuda.synchronize()
    del pool, keep
    torch.cuda.empty_cache()
# the pathology augmentation alone, nt = max_nt (the bench line's augment_pathology_160)
t = torch.from_numpy(np.arange(10) * 0.1)
pde = SH.AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict={}, BC="neumann", dt=0.1, device=dev)
_, P0 = SH.generate_shape_3d((N, N, N), [2, 2, 2], 90.0, dev)
shp_args = C5.gen_args(N).pathology_shape_generator
orig = np.random.randint
np.random.randint = lambda a, b=None: shp_args.max_nt
np.random.seed(11)
GU.augment_pathology(P0, pde, t, shp_args, dev)
torch.cuda.synchronize()
sections.append({"section": "augment_pathology", "calls": 1})
marker()
pde.nfe = 0
GU.augment_pathology(P0, pde, t, shp_args, dev)
marker()
torch.cuda.synchronize()
sections[-1]["steps"] = (pde.nfe - 2) // 6
np.random.randint = orig
# generator items
for tag, rsp, pp in (("item", 1.0, False), ("item_with_pde", 0.0, True)):
    np.random.seed(100)
    torch.manual_seed(100)
    ga = C5.gen_args(N, random_shape_prob=rsp)
    ds = G.build_datasets(ga, str(dev), cases=[C5.voronoi_case(7, pathology_prob=pp)])["all"]
    ds[0]; ds[0]
    torch.cuda.synchronize()
    sections.append({"section": tag, "calls": 1})
    marker()
    ds[0]
    marker()
    torch.cuda.synchronize()
    del ds
json.dump(sections, open(out_path, "w"), indent=1)
print("sections:", [s["section"] for s in sections])
