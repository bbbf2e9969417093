"""Where does the stitched deformed_atlas of the graph / lanes path differ from the eager path? (diagnostic)
The per-slot comparison it falls back to on a mismatch reads the dense packed rows and every voxel of a tile's maps: run it
with BFM_COMPACT=0 BFM_MASK_SKIP=0 (the compact rows and the skipped voxels came after the hazard it chased)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, numpy as np
import bench
from brainfm_amd import test_utils as TU
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
vol0, aff0 = bench.make_atlas()
if os.environ.get("DIAG_CONST") == "1":
    vol0 = torch.full_like(vol0, 100.0)
s.set_atlas(vol0, aff0)
print("vol checksum", s.atlas[0].double().sum().item(), hex(s.atlas[0].data_ptr()), flush=True)
full = bench.make_volume(n, dev)
e1, ranges, cnt = TU.tiled_inference(full, s, [80] * 3, [160] * 3, graphs=False)
e1 = {k: e1[k].clone() for k in ("regx", "regy", "regz", "deformed_atlas")}
TU.prepare_tile_graphs(full, s, [80] * 3, [160] * 3)
print("lanes", s.lanes, "gather", TU.GATHER_STITCH, flush=True)
for rep in range(reps):
    if os.environ.get("DIAG_SENTINEL") == "1":
        for key, (gr, static_in, outs) in s._graphs.items():
            outs[0][-1].fill_(-777.0)                      # the atlas row of every captured graph's static map buffer
        if hasattr(s, "_dist_bufs") and "rows" in s._dist_bufs:
            s._dist_bufs["rows"].fill_(-999.0)
        torch.cuda.synchronize()
    g, _, _ = TU.tiled_inference(full, s, [80] * 3, [160] * 3, graphs=True)
    torch.cuda.synchronize()
    bad = {k: int((e1[k] != g[k]).sum()) for k in e1}
    print("rep", rep, bad, "vol checksum", s.atlas[0].double().sum().item(), flush=True)
    if bad["deformed_atlas"]:
        d = e1["deformed_atlas"] != g["deformed_atlas"]
        idx = torch.nonzero(d)
        print("  bbox", idx.min(0)[0].tolist(), idx.max(0)[0].tolist())
        # which tiles' packed rows are wrong?  recompute each tile eagerly and compare with its slot
        buf = s._dist_bufs.get("rows") if hasattr(s, "_dist_bufs") else None
        nk = len(s.stitch_keys())
        off = 0
        for i, r in enumerate(ranges):
            (x0, x1), (y0, y1), (z0, z1) = r
            nv = TU.tile_cost(r)
            rows = buf[off:off + nv * nk].view(nk, nv) if buf is not None else None
            off += nv * nk
            lo = idx.min(0)[0].tolist(); hi = idx.max(0)[0].tolist()
            if not (x0 <= hi[0] and x1 > lo[0] and y0 <= hi[1] and y1 > lo[1] and z0 <= hi[2] and z1 > lo[2]):
                continue
            maps, label, x_cl = TU._run_tile(s, full[:, :, x0:x1, y0:y1, z0:z1])
            ref = maps["deformed_atlas"].reshape(-1) * (x_cl.reshape(-1) != 0)
            torch.cuda.synchronize()
            if rows is not None:
                w = torch.nonzero(rows[nk - 1] != ref).reshape(-1)
                print("  tile", i, r, "slot row mismatches:", int(w.numel()), w[:6].tolist(), w[-3:].tolist() if w.numel() else "",
                      "values", rows[nk - 1][w[:4]].tolist(), "expected", ref[w[:4]].tolist())
                for kk in (12, 13, 14):
                    print("     key", s.stitch_keys()[kk], "slot vs eager mismatches",
                          int((rows[kk] != (maps[s.stitch_keys()[kk]].reshape(-1) * (x_cl.reshape(-1) != 0))).sum()))
    del g
