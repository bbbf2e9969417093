"""HBM-bound kernels of the inference path, one at a time, on the shapes of a 160^3 tile.
Each kernel is launched `reps` times back to back inside one HIP event pair; GB/s = the kernel's compulsory
(algorithmic) bytes / average duration.  usage: python scripts/bench_kernels.py [reps]
Prints one line per kernel; the numbers feed DESIGN.md's per-kernel roofline table."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brainfm_amd import _lib as L
from brainfm_amd import test_utils as TU

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
sess = TU.InferenceSession(ga, ta, dev)
eng = sess.engine
lib = L.load()
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
    print("%-44s %9.1f us  %8.1f MB  %7.0f GB/s  %5.1f %% of 8 TB/s" % (name, ms * 1e3, nbytes / 1e6, gbs, 100 * gbs / PEAK))


D = H = W = 160
nv = D * H * W
x = torch.rand(D, H, W, 1, device=dev)

# --- stem: GN stats over 1 channel + conv 1->32
ly = eng.enc[0][0]
report("gn_stats + conv_stem 1->32 @160^3", timeit(lambda: eng.single_conv(ly, x, (D, H, W))), nv * 4 * 2 + nv * 32 * 4)

# --- gn_stats alone on a 64-channel map (what precedes dec4.2 / the stats a fused epilogue would save)
f64 = torch.randn(D, H, W, 64, device=dev)
ly2 = eng.dec[-1][1]
scale = torch.empty(64, device=dev)
shift = torch.empty(64, device=dev)
bound = torch.empty(ly2.groups, device=dev)
wsb = lib.bfm_gn_stats_workspace(64, 0, D, H, W, None)
ws = torch.empty(wsb + 1024, dtype=torch.uint8, device=dev)
st = L.stream_ptr()
report("gn_stats 64ch @160^3 (partial + finalize)",
       timeit(lambda: L.check(lib.bfm_gn_stats(L.ptr(f64), 64, None, 0, D, H, W, None, L.ptr(ly2.gamma), L.ptr(ly2.beta),
                                               ly2.groups, 1e-5, L.ptr(scale), L.ptr(shift), L.ptr(bound), L.ptr(ws),
                                               ws.numel(), st), "gn")), nv * 64 * 4)

# --- max-pool 64ch 160^3 -> 80^3
report("maxpool2 64ch @160^3", timeit(lambda: eng.maxpool(f64, (D, H, W))), nv * 64 * 4 * (1 + 1 / 8))

# --- fused tail: 64 features -> 15 maps + label
tail = sess.model.head.tail(eng)
xin = x[..., 0].contiguous()
n_maps = len(tail.map_names)
report("tail_heads 64ch -> %d maps + label @160^3" % n_maps,
       timeit(lambda: tail.run(f64, (D, H, W), input_cl=xin, want_feat=False, want_seg=False)),
       nv * (64 * 4 + 4 + n_maps * 4 + 8))
report("tail_heads + feat_norm + softmax out",
       timeit(lambda: tail.run(f64, (D, H, W), input_cl=xin, want_feat=True, want_seg=True)),
       nv * (64 * 4 * 2 + 4 + n_maps * 4 + 8 + 56 * 4))

# --- stitch of one tile into a 256^3 accumulator set
maps, _, _, label = tail.run(f64, (D, H, W), input_cl=xin, want_feat=False, want_seg=False)
names = list(maps.keys())
keys = [k for k in TU.STITCH_KEYS if k in names or k == "label"]
sel = torch.tensor([names.index(k) if k != "label" else -1 for k in keys], dtype=torch.int32, device=dev)
acc = torch.zeros((len(keys), 256, 256, 256), device=dev)
report("stitch_accumulate_multi %d maps @160^3" % len(keys),
       timeit(lambda: L.check(lib.bfm_stitch_accumulate_multi(L.ptr(tail.last_buf), nv, L.ptr(sel), len(keys),
                                                              L.ptr(label), L.ptr(xin), D, H, W, L.ptr(acc), 256, 256,
                                                              256, 40, 40, 40, L.stream_ptr()), "stitch")),
       nv * (4 + (len(keys) - 1) * 4 + 8 + len(keys) * 8))
