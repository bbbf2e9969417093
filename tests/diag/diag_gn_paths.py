"""Which GroupNorm of one 256^3 step takes its moments from the producer's rows and which re-reads its input (and why).
   python tests/diag/diag_gn_paths.py [size=256]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brainfm_amd import test_utils as TU, engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
torch.manual_seed(1)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
sess = TU.InferenceSession(ga, ta, dev, passes=3)
sess.use_graphs = False
full = bench.make_volume(n, dev)
log = collections.Counter()
eng = sess.engine
cls = type(eng)
o1, o2 = cls._gn_stats, cls._batch_stats


def short(ly):
    return ly.name.replace("backbone.", "").replace(".basic_module.SingleConv", ".")


def g1(self, ly, A, ca, B, cb, dims, lo_dims, upp, scale, shift, bound, ws_min):
    ra = getattr(A, "_bfm_rows", None)
    rb = getattr(B, "_bfm_rows", None) if B is not None else None
    rows = ra is not None and (B is None or (rb is not None and tuple(dims) == tuple(2 * v for v in lo_dims)))
    log[(short(ly), tuple(dims), "rows" if rows else "re-read (A rows %s, B rows %s)" % (ra is not None, None if B is None else rb is not None))] += 1
    return o1(self, ly, A, ca, B, cb, dims, lo_dims, upp, scale, shift, bound, ws_min)


def g2(self, ly, A, ca, B, cb, S, dims, lo_dims, upp, scale, shift, bound):
    ra = getattr(A, "_bfm_rows", None)
    rb = getattr(B, "_bfm_rows", None) if B is not None else None
    rows = ra is not None and ra[1] <= 128 and (B is None or (rb is not None and rb[1] <= 128))
    log[(short(ly) + " [batch of %d]" % S, tuple(dims), "rows" if rows else "re-read (A rows %s, B rows %s)" % (ra if ra is None else ra[1], None if B is None else (rb if rb is None else rb[1])))] += 1
    return o2(self, ly, A, ca, B, cb, S, dims, lo_dims, upp, scale, shift, bound)


out = TU.tiled_inference(full, sess, graphs=False, batched=True)      # tunes / packs
torch.cuda.synchronize()
cls._gn_stats, cls._batch_stats = g1, g2
out = TU.tiled_inference(full, sess, graphs=False, batched=True)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, dims, how), c in sorted(log.items()):
    print("%-34s %-16s x%-3d %s" % (name, "x".join(map(str, dims)), c, how))
    tot["rows" if how == "rows" else "re-read"] += c
print(dict(tot))
