"""Benchmark of the north-star path: whole-volume multi-task tiled inference on a 256^3 volume.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--passes 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic 256^3 volume (27 overlapping tiles, win 160 /
stride 80, all 9 task heads, fused tail, on-device stitching), input resident in HBM.  N>1 shards the
tiles of the SAME volume over ranks (strong scaling) and gathers the masked tile outputs to rank 0
over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def make_volume(n, device):
    """rand inside a centred ellipsoid (semi-axes 100,110,90 scaled), exact 0 outside."""
    g = torch.Generator().manual_seed(0)
    s = n / 256.0
    ax = torch.arange(n, dtype=torch.float32) - (n - 1) / 2.0
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    ell = ((zz / (100 * s)) ** 2 + (yy / (110 * s)) ** 2 + (xx / (90 * s)) ** 2) <= 1
    v = torch.rand((n, n, n), generator=g) * ell
    return v[None, None].to(device)


def conv_flops_tile(dims, fm=(64, 128, 256, 512, 1024, 2048)):
    """Algorithmic FLOPs (2*MACs) of all 3x3x3 convs + heads for one tile (SURVEY 8d formula)."""
    d = list(dims)
    total, sizes = 0.0, []
    for i, co in enumerate(fm):
        if i > 0:
            d = [v // 2 for v in d]
        ci = 1 if i == 0 else fm[i - 1]
        c1 = max(co // 2, ci)
        nv = d[0] * d[1] * d[2]
        total += 2.0 * 27 * nv * (ci * c1 + c1 * co)
        sizes.append(list(d))
    rev = list(reversed(fm))
    rs = list(reversed(sizes))
    for i in range(len(rev) - 1):
        nv = rs[i + 1][0] * rs[i + 1][1] * rs[i + 1][2]
        total += 2.0 * 27 * nv * ((rev[i] + rev[i + 1]) * rev[i + 1] + rev[i + 1] * rev[i + 1])
    nv = dims[0] * dims[1] * dims[2]
    total += 2.0 * nv * 64 * 69
    return total


def cpu_baseline(state_dict, full, n):
    """The CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host, rank 0 only,
    on a bounded sample: one central 128^3 tile of the volume (the size of BASELINE.json's configs[0]), all heads:
    ~10-15 s of CPU work on the GPU box's host."""
    from oracle import unet_ref as O
    cores = torch.get_num_threads()
    s = 128 if n >= 160 else max(16, n // 2)
    tile = full[:, :, n // 2 - s // 2:n // 2 + s // 2, n // 2 - s // 2:n // 2 + s // 2,
                n // 2 - s // 2:n // 2 + s // 2].cpu().contiguous()
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    t0 = time.time()
    with torch.no_grad():
        O.forward_all(tile, sd, f_maps=64, num_levels=6)
    dt = time.time() - t0
    return {"value": tile.numel() / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": "one central %d^3 tile of the volume, all 9 heads, oracle/unet_ref.py (torch-CPU fp32), "
                      "1 run, %.1f s" % (s, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--passes", type=int, default=3, help="3 = fp32-grade split-f16 MFMA (parity mode), 1 = fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layer-table", default=None, help="write the per-layer conv timing table of the instrumented pass here")
    ap.add_argument("--dist-path", action="store_true",
                    help="run the multi-GPU code path (pack, RCCL gather, root accumulation) even with one rank")
    ap.add_argument("--no-graphs", action="store_true", help="submit every kernel from python instead of hipGraph replay")
    ap.add_argument("--roofline-reps", type=int, default=3,
                    help="back-to-back launches per HIP-event bracket in the instrumented conv pass")
    args = ap.parse_args()

    import torch.distributed as dist
    from brainfm_amd import test_utils as TU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # dry run of the N > 1 path on a one-GPU box: BFM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # BFM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks per device); never set for a measurement
    if os.environ.get("BFM_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("BFM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.dist_path
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n = args.size
    torch.manual_seed(1)                                   # default nn init under seed 1
    ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
    sess = TU.InferenceSession(ga, ta, dev, passes=args.passes)
    full = make_volume(n, dev)
    stride, win = [80] * 3, [160] * 3
    ranges = TU.tiling_ranges((n, n, n), stride, win)
    eng = sess.engine
    sess.use_graphs = not args.no_graphs

    def step():
        if use_dist:
            return TU.tiled_inference_distributed(full, sess, stride, win)
        return TU.tiled_inference(full, sess, stride, win)

    # setup (untimed, once per session): tune the conv variants and capture one hipGraph per tile shape
    if sess.use_graphs:
        TU.prepare_tile_graphs(full, sess, stride, win, world=world, rank=rank)   # setup, like weight packing
    for _ in range(args.warmup):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    debug = os.environ.get("BFM_BENCH_DEBUG") == "1"       # per-step times on stderr (adds a device sync per step)
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        if debug:
            torch.cuda.synchronize()
            print("rank %d step %.1f ms" % (rank, 1e3 * (time.perf_counter() - ts)), file=sys.stderr, flush=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # dominant kernel: conv_mfma.  The timed region replays hipGraphs, which HIP events cannot bracket per kernel, so
    # the same step is run once more eagerly right after it with every conv launch issued `reps` times back to back
    # inside one HIP event pair on the launch stream (idempotent; back-to-back so the bracket holds kernel time rather
    # than python submission gaps).  Launch duration = bracket / reps.
    prof = []
    if args.roofline_reps > 0:                      # 0: skip (used for rocprofv3 runs that should hold the timed steps only)
        eng.prof = []
        eng.prof_reps = args.roofline_reps
        g = sess.use_graphs
        sess.use_graphs = False
        step()
        torch.cuda.synchronize()
        sess.use_graphs = g
        prof = eng.prof
        eng.prof = None
    k_ms = sum(p[0].elapsed_time(p[1]) / p[4] for p in prof)
    k_fl = sum(p[2] for p in prof)
    k_by = sum(p[3] for p in prof)
    if args.layer_table and rank == 0:
        tab = {}
        for p in prof:
            e = tab.setdefault(p[5], [0, 0.0, 0.0])
            e[0] += 1
            e[1] += p[0].elapsed_time(p[1]) / p[4]
            e[2] += p[2]
        with open(args.layer_table, "w") as f:
            f.write("# conv launches of one step grouped by (layer, Cin, Cout, dims, plan[WM,WN,TD,TH,TW,splitk,ver,0])\n")
            f.write("%-12s %5s %5s %-16s %-28s %5s %10s %8s %7s\n" % ("layer", "cin", "cout", "dims", "plan", "n", "ms_total",
                                                                  "us_avg", "TF/s"))
            for key, (cnt, ms, fl) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                f.write("%-12s %5d %5d %-16s %-28s %5d %10.3f %8.1f %7.1f\n" % (
                    key[0], key[1], key[2], "x".join(map(str, key[3])), ",".join(map(str, key[4])), cnt, ms,
                    ms * 1e3 / cnt, fl / (ms * 1e-3) / 1e12))
    agg = torch.tensor([k_ms, k_fl, k_by, float(len(prof))], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(agg)
    k_ms, k_fl, k_by, k_n = [float(v) for v in agg.tolist()]
    achieved = k_fl / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    # HBM traffic of the conv kernels: rocprofv3 PMC passes cannot run inside this process; the committed summary of the
    # same workload (scripts/pmc_traffic.py, FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md) is reported per launch
    traffic, traffic_note = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_conv_hbm_traffic.json")
    if os.path.exists(tpath) and args.size == 256:
        try:
            tj = json.load(open(tpath))
            traffic = tj["kernels"]["conv"]["hbm_bytes_per_launch"]
            traffic_note = "profiles/r01_conv_hbm_traffic.json: " + tj["source"]
        except Exception:
            traffic = None
    peak = 2500.0                                            # dense f16 MFMA, MI355X_MICROARCH.md
    if rank == 0:
        tile_vox = sum(TU.tile_cost(r) for r in ranges)
        flops_step = sum(conv_flops_tile([r[a][1] - r[a][0] for a in range(3)]) for r in ranges)
        line = {
            "metric": "voxels/sec whole-volume multi-task inference, 256^3 tiled",
            "value": n ** 3 * args.steps / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f16x3-split (fp32-grade)" if args.passes == 3 else "f16", "data": "synthetic",
            "config": {"workload": "%d^3 volume, reference tiling win160/stride80 -> %d tiles, UNet3D f64 x6 levels, "
                                   "9 heads (69 ch), fused tail + on-device stitch" % (n, len(ranges)),
                       "tile_voxels_per_step": tile_vox, "tile_voxels_per_s": tile_vox * args.steps / dt,
                       "algorithmic_tflop_per_step": flops_step / 1e12,
                       "end_to_end_tflops": flops_step * args.steps / dt / 1e12, "mfma_passes": args.passes,
                       "submission": "hipGraph replay per tile shape" if sess.use_graphs else "eager",
                       "parallelism": "tiles sharded over %d rank(s), gather to rank 0" % world,
                       "tiles_in_flight_per_gpu": sess.lanes if sess.use_graphs else 1},
            "roofline": {"bound": "mfma", "kernel": "conv_mfma* (the %d conv launches of one step, all variants)" % int(k_n),
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "kernel_ms_per_step": k_ms / max(world, 1), "avg_launch_us": k_ms * 1e3 / max(k_n, 1),
                         "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": k_by / max(k_n, 1), "algorithmic_bytes_per_step": k_by,
                         "note": "achieved = algorithmic conv FLOPs of one step / summed launch durations; durations "
                                 "from HIP events around %d back-to-back launches of each conv in an instrumented "
                                 "eager replay of the step right after the timed region; the kernel issues %dx the "
                                 "algorithmic FLOPs in f16 MFMA" % (args.roofline_reps, args.passes)},
        }
        if not args.no_cpu_baseline and world == 1:
            sd = {k: v for k, v in sess.model.state_dict().items()}
            line["cpu_baseline"] = cpu_baseline(sd, full, n)
        else:
            line["cpu_baseline"] = None
        try:                                                  # librccl's version banner sits in the C stdio buffer and
            import ctypes                                     # would otherwise land after the JSON line at exit
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
