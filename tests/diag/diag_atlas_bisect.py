"""Round 5 bisection of the atlas-gather hazard (HISTORY.md section 3.3; -DBFM_DIAG build): which loads of deformed_atlas have to be
ordinary for wrong texels to appear beside conv_mfma / conv_mfma16, and what the wrong values are.
    python tests/diag/diag_atlas_bisect.py [iterations=100]
One atlas stream (eager) + the co-runner on a side stream, constant atlas (any value other than 100 / 0 is a wrong load)
and a smooth one.  Load forms (BFM_ATLAS_PLAIN_LOADS_NOW): 1 ordinary everywhere; 4 ordinary, mask predicate dropped (no
divergent gather); 5 ordinary texels, the four coalesced row loads at agent scope; 7 ordinary, row loads drained before the
first texel load; 3 texels at agent scope (rows ordinary)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brainfm_amd import _lib as L
if os.environ.get("BFM_DIAG_LIB"):                 # a variant build (scripts/build_variant.py)
    L.LIB_PATH = os.path.join(ROOT, os.environ["BFM_DIAG_LIB"])
from brainfm_amd import test_utils as TU
from brainfm_amd.engine import _Layer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
lib = L.load()
n = 160 * 160 * 80
A = (C.c_float * 12)(-1, 0, 0, 128, 0, 0, -1, 128, 0, 1, 0, 128)
g = torch.Generator().manual_seed(0)
table = [(torch.randn(n, generator=g) * 0.05).to(dev) for _ in range(3)]
mask = (torch.rand(n, generator=g) > 0.3).float().to(dev)
ax = torch.arange(256, dtype=torch.float32)
i, j, k = torch.meshgrid(ax, ax, ax, indexing="ij")
atlases = {"constant 100": torch.full((256, 256, 256), 100.0, device=dev),
           "smooth": (110. + 60. * torch.sin(i / 17.) * torch.cos(j / 23.) + 40. * torch.sin(k / 13. + 0.5)).to(dev)}
side, main = torch.cuda.Stream(), torch.cuda.Stream()
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


def cfg_of(ver):
    c = (C.c_int * 8)()
    L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, cd[0], cd[1], cd[2], c), "plan")
    c[6] = ver
    return c


def gather(rows, out, atlas):
    for r in range(3):
        rows[r].copy_(table[r])
    L.check(lib.bfm_deformed_atlas_tile(L.ptr(mask), L.ptr(rows[0]), L.ptr(rows[1]), L.ptr(rows[2]), L.ptr(atlas), 256, 256, 256, A, n,
                                        L.ptr(out), L.stream_ptr()), "atlas")


NAMES = {1: "ordinary loads everywhere", 4: "ordinary, mask predicate dropped", 5: "ordinary texels, rows at agent scope",
         7: "ordinary, rows drained before the texels", 3: "texels at agent scope"}
SPREAD = float(os.environ.get("BFM_BISECT_SPREAD", "1"))          # scales the registration tables: the sampled region of the atlas
table = [t * SPREAD for t in table]
print("registration tables x %g: the gather samples the atlas within +- %.1f voxels of its centre" % (SPREAD, 15 * SPREAD), flush=True)
for ver, vname in ((2, "conv_mfma16"), (0, "conv_mfma"), (4, "conv_wino4d (LDS-DMA raw chunks, round 5)"), (3, "conv_wino (no LDS-DMA)")):
    ccfg = cfg_of(ver)
    eng._conv_launch(ly, cA, cin, None, 0, cd, None, cscale, cshift, cbound, 8, ccfg, cout_t, cws)
    torch.cuda.synchronize()
    for aname, atlas in atlases.items():
        os.environ["BFM_ATLAS_PLAIN_LOADS_NOW"] = "0"
        rows = [torch.empty(n, device=dev) for _ in range(3)]
        want = torch.empty(n, device=dev)
        gather(rows, want, atlas)
        torch.cuda.synchronize()
        for loads in ((1, 4, 5, 7, 3) if SPREAD == 1 else (1,)):
            os.environ["BFM_ATLAS_PLAIN_LOADS_NOW"] = str(loads)
            out = torch.empty(n, device=dev)
            bad, shown = 0, 0
            for it in range(iters):
                out.fill_(-7.0)
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    for _ in range(6):
                        eng._conv_launch(ly, cA, cin, None, 0, cd, None, cscale, cshift, cbound, 8, ccfg, cout_t, cws)
                with torch.cuda.stream(main):
                    gather(rows, out, atlas)
                torch.cuda.synchronize()
                d = out != want
                b = int(d.sum())
                if b and shown < 2:
                    idx = torch.nonzero(d).reshape(-1)
                    runs = int(((idx[1:] - idx[:-1]) != 1).sum()) + 1
                    print("     it %d: %d wrong in %d runs of consecutive voxels, first at %d; got %s want %s" % (
                        it, b, runs, int(idx[0]), [round(v, 4) for v in out[idx[:6]].tolist()], [round(v, 4) for v in want[idx[:6]].tolist()]),
                        flush=True)
                    shown += 1
                bad += b
            print("%-42s %-13s %-42s wrong voxels in %d launches: %d" % (vname, aname, NAMES[loads], iters, bad), flush=True)
