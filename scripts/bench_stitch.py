"""Rank 0's stitch of the multi-GPU path on a 256^3 volume (27 tiles, 16 keys): per-tile accumulate + divide against
the one-launch gather (bfm_stitch_gather_multi).  python scripts/bench_stitch.py [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from brainfm_amd import test_utils as TU  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
shape = (n, n, n)
ranges = TU.tiling_ranges(shape, [80] * 3, [160] * 3)
K = 16
srcs = [torch.rand((K, TU.tile_cost(r)), device=dev) for r in ranges]
ops = TU.HipStitchOps(None)
cnt = TU.count_volume(shape, ranges, dev)
alg = sum(s.numel() for s in srcs) * 4 + K * n ** 3 * 4


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def per_tile():
    acc = torch.zeros((K,) + shape, dtype=torch.float32, device=dev)
    for r, rows in zip(ranges, srcs):
        ops.add_all(acc, rows, r, shape)
    ops.finalize_all(acc, cnt)
    return acc


def gather():
    acc = torch.empty((K,) + shape, dtype=torch.float32, device=dev)
    ops.gather_all(acc, srcs, ranges, shape)
    return acc


assert torch.equal(per_tile().view(torch.int32), gather().view(torch.int32))
t0, t1 = timed(per_tile), timed(gather)
print("%d^3, %d tiles, %d keys: per-tile accumulate + divide %.3f ms; one-launch gather %.3f ms = %.2f TB/s of the "
      "%.2f GB it must move" % (n, len(ranges), K, t0, t1, alg / t1 / 1e9, alg / 1e9))

# the compact form on the bench volume (a head-sized ellipsoid): index, rows [K][count], stitch through the index
import bench  # noqa: E402
full = bench.make_volume(n, dev)
idx = ops.index_volume(full, ranges, counts=True)
K2 = 17
dense = [torch.rand((K2, TU.tile_cost(r)), device=dev) for r in ranges]
comp = []
for i, (r, d) in enumerate(zip(ranges, dense)):
    m = full[0, 0, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]].reshape(-1) != 0
    d[:, ~m] = 0
    comp.append(d[:, m].contiguous())


def gather_dense():
    acc = torch.empty((K2,) + shape, dtype=torch.float32, device=dev)
    ops.gather_all(acc, dense, ranges, shape)
    return acc


def gather_compact():
    acc = torch.empty((K2,) + shape, dtype=torch.float32, device=dev)
    ops.gather_all(acc, comp, ranges, shape, index=idx)
    return acc


assert torch.equal(gather_dense().view(torch.int32), gather_compact().view(torch.int32))
t2, t3 = timed(gather_dense), timed(gather_compact)
t4 = timed(lambda: ops.index_volume(full, ranges))
print("bench volume, %d keys: dense gather %.3f ms (%.2f GB of rows); compact gather %.3f ms (%.2f GB of rows); "
      "index of all tiles %.3f ms" % (K2, t2, sum(d.numel() for d in dense) * 4 / 1e9, t3,
                                        sum(c.numel() for c in comp) * 4 / 1e9, t4))
