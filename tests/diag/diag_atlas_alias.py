"""Do the static output buffers of the captured tile graphs overlap each other? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from brainfm_amd import test_utils as TU
dev = torch.device("cuda:0")
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
s.set_atlas(*bench.make_atlas())
full = bench.make_volume(256, dev)
TU.prepare_tile_graphs(full, s, [80] * 3, [160] * 3)
iv = []
for key, (g, static_in, outs) in s._graphs.items():
    maps_buf, names, label, x_cl = outs
    for nm, t in (("static_in", static_in), ("maps_buf", maps_buf), ("label", label), ("x_cl", x_cl)):
        st = t.untyped_storage()
        iv.append((st.data_ptr(), st.data_ptr() + st.nbytes(), key, nm, t.data_ptr()))
    print(key, "maps_buf", hex(maps_buf.data_ptr()), tuple(maps_buf.shape), "storage bytes", maps_buf.untyped_storage().nbytes(),
          "need", maps_buf.numel() * 4, "names", len(names))
iv.sort()
bad = 0
for a, b in zip(iv, iv[1:]):
    if a[1] > b[0] and not (a[3] in ("static_in", "x_cl") and b[3] in ("static_in", "x_cl") and a[2] == b[2]):
        print("OVERLAP", a[2:4], hex(a[0]), hex(a[1]), "with", b[2:4], hex(b[0]), hex(b[1]))
        bad += 1
print("overlaps:", bad)
