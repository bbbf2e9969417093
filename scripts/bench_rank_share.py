"""What each rank of an N-rank run of the 256^3 volume computes, timed alone on ONE GPU (no transfers): its batches of
same-shape tiles on its two lanes in the order tiled_inference_distributed runs them, plus -- on rank 0 -- the one-launch
stitch of all 27 tiles.
A model of the N-GPU step without the exchange: max over ranks.   python scripts/bench_rank_share.py [N=8] [size=256]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brainfm_amd import test_utils as TU  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
torch.manual_seed(1)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
sess = TU.InferenceSession(ga, ta, dev, passes=3)
sess.use_graphs = True
full = bench.make_volume(n, dev)
stride, win = [80] * 3, [160] * 3
ranges = TU.tiling_ranges((n, n, n), stride, win)
owner = TU.assign_tiles(ranges, world)
batches_of = [TU.tile_batches(ranges, [i for i in range(len(ranges)) if owner[i] == r_], min_batches=sess.lanes)
              for r_ in range(world)]
ops = TU.HipStitchOps(sess)
nkeys = len(sess.stitch_keys())
offs, total = [], 0
for r in ranges:
    offs.append(total)
    total += TU.tile_cost(r) * nkeys
buf = torch.zeros(total, dtype=torch.float32, device=dev)
srcs = [buf[offs[i]:offs[i] + TU.tile_cost(r) * nkeys].view(nkeys, TU.tile_cost(r)) for i, r in enumerate(ranges)]
acc = torch.empty((nkeys, n, n, n), dtype=torch.float32, device=dev)


def share(rank):
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event()
    start.record(main)
    index = ops.index_volume(full, ranges) if TU.COMPACT else None
    load, last = [0] * sess.lanes, {}
    for batch in batches_of[rank]:
        k = min(range(sess.lanes), key=lambda j: (load[j], j))
        load[k] += sum(TU.tile_time(ranges[i]) for i in batch)
        ims = [full[:, :, ranges[i][0][0]:ranges[i][0][1], ranges[i][1][0]:ranges[i][1][1], ranges[i][2][0]:ranges[i][2][1]]
               for i in batch]
        outs = [buf[offs[i]:offs[i] + TU.tile_cost(ranges[i]) * nkeys] for i in batch]
        _, _, done = ops.run_group(ims, outs, lane=k, after=start, index=index, tiles_idx=batch,
                                   strides=[TU.tile_cost(ranges[i]) for i in batch])
        if done is not None:
            last[k] = done
    for ev in last.values():
        main.wait_event(ev)
    if rank == 0:
        ops.gather_all(acc, srcs, ranges, (n, n, n), index=index)
    return [i for b_ in batches_of[rank] for i in b_]


worst = 0.0
for rank in range(world):
    for _ in range(3):                                      # eager pass, capture, first replay of this rank's batch graphs
        share(rank)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        mine = share(rank)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    worst = max(worst, ms)
    print("rank %d of %d: batches %s (tiles x 80^3 units each)%s: %.2f ms" % (
        rank, world, ["%dx%d" % (len(b_), TU.tile_cost(ranges[b_[0]]) // 512000) for b_ in batches_of[rank]],
        " + stitch of all tiles" if rank == 0 else "", ms))
print("modelled %d-GPU step without the exchange: %.2f ms = %.0f Mvoxel/s" % (world, worst, n ** 3 / worst / 1e3))
