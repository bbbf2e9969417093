"""Stand-alone reproducer attempt for the atlas-gather hazard of HISTORY.md section 3.3 (round 3, time-boxed).

    BFM_ATLAS_PLAIN_LOADS=1 python tests/diag/diag_atlas_repro.py     # texels by plain global loads (the form that failed)
    python tests/diag/diag_atlas_repro.py                              # texels by sc0 sc1 loads (what ships)

No network: per lane a hipGraph of [a writer kernel that fills the three registration rows the way the tail does (values
from a fixed table, so the expected output is known) -> bfm_deformed_atlas_tile], private row buffers per lane, ONE atlas
shared by both, the two graphs replayed concurrently on two streams many times.  Variations: the atlas constant (any
wrong texel shows) or smooth; a third stream streaming 1 GB through the caches meanwhile; eager instead of graph replay.
Prints the number of wrong voxels per variation.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brainfm_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.load()
n = 160 * 160 * 80
A = (C.c_float * 12)(-1, 0, 0, 128, 0, 0, -1, 128, 0, 1, 0, 128)
g = torch.Generator().manual_seed(0)
tables = [[(torch.randn(n, generator=g) * 0.05).to(dev) for _ in range(3)] for _ in range(2)]
mask = (torch.rand(n, generator=g) > 0.3).float().to(dev)
ax = torch.arange(256, dtype=torch.float32)
i, j, k = torch.meshgrid(ax, ax, ax, indexing="ij")
atlases = {"constant 100": torch.full((256, 256, 256), 100.0, device=dev),
           "smooth": (110. + 60. * torch.sin(i / 17.) * torch.cos(j / 23.) + 40. * torch.sin(k / 13. + 0.5)).to(dev)}
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
side = torch.cuda.Stream()
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)       # 1 GB to stream through the caches


def one(lane, rows, out, atlas):
    # the "tail": the three registration rows are (re)written right before the gather reads them
    for r in range(3):
        rows[r].copy_(tables[lane][r])
    L.check(lib.bfm_deformed_atlas_tile(L.ptr(mask), L.ptr(rows[0]), L.ptr(rows[1]), L.ptr(rows[2]), L.ptr(atlas), 256, 256,
                                        256, A, n, L.ptr(out), L.stream_ptr()), "atlas")


# a co-runner that fills its LDS by LDS-DMA (global_load_lds_dwordx4), the way the conv kernels of the tile flow do
from brainfm_amd import test_utils as TU
from brainfm_amd.engine import _Layer
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
cin = cout = 128
cd = (40, 40, 40)
cA = torch.randn(*cd, cin, device=dev)
cscale, cshift, cbound = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1, torch.full((8,), 6.0, device=dev)
cout_t, cws = torch.empty(*cd, cout, device=dev), torch.empty(1 << 26, dtype=torch.uint8, device=dev)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "corunner", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
ccfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, cd[0], cd[1], cd[2], ccfg), "plan"); ccfg[6] = int(os.environ.get("BFM_DIAG_CORUNNER_VER", "0"))   # 0 conv_mfma (LDS-DMA weights), 3 conv_wino (weights L2 -> VGPR)


def conv_beside(k=6):
    for _ in range(k):
        eng._conv_launch(ly, cA, cin, None, 0, cd, None, cscale, cshift, cbound, 8, ccfg, cout_t, cws)


conv_beside(1)
torch.cuda.synchronize()

for name, atlas in atlases.items():
    # expected outputs: one serial, synchronised run per lane (with the same loads; a constant atlas has its own check)
    want = []
    for lane in range(2):
        rows = [torch.empty(n, device=dev) for _ in range(3)]
        out = torch.empty(n, device=dev)
        one(lane, rows, out, atlas)
        torch.cuda.synchronize()
        want.append(out.clone())
        if name.startswith("constant"):
            inside = out != 0
            assert bool(((out == 100.0) | ~inside).all()), "serial run already wrong"
    for mode in ("graph replay, two lanes", "graph replay, two lanes + 1 GB streaming beside", "eager, two lanes",
                 "graph replay, two lanes + conv_mfma (LDS-DMA) beside", "eager, one lane + conv_mfma (LDS-DMA) beside"):
        bufs, graphs = [], []
        for lane in range(2):
            rows = [torch.empty(n, device=dev) for _ in range(3)]
            out = torch.empty(n, device=dev)
            bufs.append((rows, out))
            if mode.startswith("graph"):
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[lane]):
                    one(lane, rows, out, atlas)                     # warm-up
                    streams[lane].synchronize()
                    with torch.cuda.graph(gr, stream=streams[lane]):
                        one(lane, rows, out, atlas)
                graphs.append(gr)
        bad = 0
        for it in range(150):
            for lane in range(2):
                bufs[lane][1].fill_(-7.0)
            torch.cuda.synchronize()
            if "streaming" in mode:
                with torch.cuda.stream(side):
                    big.mul_(1.0)
            if "conv_mfma" in mode:
                with torch.cuda.stream(side):
                    conv_beside()
            for lane in range(1 if "one lane" in mode else 2):
                with torch.cuda.stream(streams[lane]):
                    if graphs:
                        graphs[lane].replay()
                    else:
                        one(lane, bufs[lane][0], bufs[lane][1], atlas)
            torch.cuda.synchronize()
            for lane in range(1 if "one lane" in mode else 2):
                d = bufs[lane][1] != want[lane]
                b = int(d.sum())
                if b and bad < 5:
                    idx = torch.nonzero(d).reshape(-1)
                    print("   it %d lane %d: %d wrong, first at %s (got %s, want %s)" % (
                        it, lane, b, idx[:4].tolist(), bufs[lane][1][idx[:2]].tolist(), want[lane][idx[:2]].tolist()), flush=True)
                bad += b
        print("%-14s %-50s wrong voxels in 150 x 2 launches: %d" % (name, mode, bad), flush=True)
print("loads:", "plain" if os.environ.get("BFM_ATLAS_PLAIN_LOADS") == "1" else "sc0 sc1")
