"""Device memory of the tile flow after two volumes at 256^3 and 512^3 (HISTORY.md section 2)."""
import os
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from brainfm_amd import test_utils as TU
dev = torch.device("cuda:0")
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
s.set_atlas(*bench.make_atlas())
for n in (256, 512):
    full = bench.make_volume(n, dev)
    TU.prepare_tile_graphs(full, s, [80] * 3, [160] * 3)
    for _ in range(2):
        TU.tiled_inference(full, s, [80] * 3, [160] * 3, batched=True)
    torch.cuda.synchronize()
    print("%d^3: allocated %.1f GB, reserved %.1f GB, peak allocated %.1f GB" % (
        n, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30))
