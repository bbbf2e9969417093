"""Which gather goes wrong beside conv_wino4d, where and how (round 5 follow-up of tests/test_gpu_synth.py's co-runner test)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer
from brainfm_amd.generator_utils import fast_3D_interp_torch
from brainfm_amd.interpol import grid_pull
DEV = "cuda:0"
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "synth_interp.npz")))
d2 = dict(np.load(os.path.join(ROOT, "tests", "golden", "synth_grid_pull.npz")))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
X = T(d["X1"]); vol, grid = T(d2["vol"]), T(d2["grid"])
print("X", tuple(X.shape), "points", d["II"].shape, "vol", tuple(vol.shape), "grid", tuple(grid.shape))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
side = torch.cuda.Stream()
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, torch.device(DEV)).engine
cin = cout = 128
cd = (40, 40, 40)
cA = torch.randn(*cd, cin, device=DEV)
csc, csh, cbd = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1, torch.full((8,), 6.0, device=DEV)
cout_t, cws = torch.empty(*cd, cout, device=DEV), torch.empty(1 << 26, dtype=torch.uint8, device=DEV)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "corunner", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
ver = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
ccfg = (C.c_int * 8)()
L.check(eng.lib.bfm_conv3x3x3_mfma_plan(cin, cout, cd[0], cd[1], cd[2], ccfg), "plan")
ccfg[6] = ver


def conv_beside():
    for _ in range(6):
        eng._conv_launch(ly, cA, cin, None, 0, cd, None, csc, csh, cbd, 8, ccfg, cout_t, cws)


conv_beside(); torch.cuda.synchronize()
ref_conv = cout_t.clone()
lanes = []
for lane in range(2):
    ii, jj, kk = T(d["II"]), T(d["JJ"]), T(d["KK"])
    gcopy = grid.clone()
    outs = {}

    def body(ii=ii, jj=jj, kk=kk, gcopy=gcopy, outs=outs):
        outs["interp"] = fast_3D_interp_torch(X, ii, jj, kk, "linear")
        outs["pull_zero"] = grid_pull(vol, gcopy, interpolation="linear", bound="zero", extrapolate=False, prefilter=False)

    g = None
    with torch.cuda.stream(streams[lane]):
        body(); streams[lane].synchronize()
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[lane]):
                body()
    lanes.append((g, outs, body))
bad = {"interp": 0, "pull_zero": 0, "conv": 0}
for it in range(int(os.environ.get("BFM_DIAG_ROUNDS", "60"))):
    for _, outs, _ in lanes:
        for v in outs.values():
            v.fill_(float("nan"))
    cout_t.fill_(float("nan"))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        conv_beside()
    for lane, (g, _, body) in enumerate(lanes):
        with torch.cuda.stream(streams[lane]):
            if g is not None:
                g.replay()
            else:
                body()
    torch.cuda.synchronize()
    if not torch.equal(cout_t, ref_conv):
        bad["conv"] += int((cout_t != ref_conv).sum())
    for lane, (_, outs, _) in enumerate(lanes):
        a = outs["interp"].cpu().numpy(); w = d["lin1"]
        m = np.flatnonzero(a != w)
        if m.size:
            bad["interp"] += m.size
            if bad["interp"] <= 3 * m.size:
                print("it %d lane %d interp: %d wrong at %s got %s want %s II %s" % (it, lane, m.size, m[:6], a[m[:6]], w[m[:6]], d["II"].reshape(-1)[m[:3]]), flush=True)
        a = outs["pull_zero"].cpu().numpy(); w = d2["out_zero_0"]
        m = np.flatnonzero(np.abs(a - w).reshape(-1) > 1e-6 * np.abs(w).max())
        if m.size:
            bad["pull_zero"] += m.size
            if bad["pull_zero"] <= 3 * m.size:
                print("it %d lane %d pull_zero: %d wrong at %s got %s want %s" % (it, lane, m.size, m[:6], a.reshape(-1)[m[:6]], w.reshape(-1)[m[:6]]), flush=True)
print("co-runner ver %d, %s: wrong elements in 60 rounds x 2 lanes:" % (ver, mode), bad)
