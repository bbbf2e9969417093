"""Benchmark of the north-star path: whole-volume multi-task tiled inference on a 256^3 volume.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--passes 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic 256^3 volume (27 overlapping tiles, win 160 /
stride 80, all 9 task heads, fused tail, the deformed atlas per tile, on-device stitching of the 17 keys
scripts/demo_test.py:107-119 stitches), input resident in HBM.  N>1 shards the tiles of the SAME volume
over ranks (strong scaling) and gathers the masked tile outputs to rank 0 over RCCL.  Prints ONE JSON line
on rank 0.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process
per device, before this process has made any GPU call) and relays rank 0's line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CONV_KERNELS = ["conv_wino", "conv_wino4", "conv_wino_masked", "conv_wino_uniform", "conv_upfold", "conv_mfma", "conv_mfma_ws", "conv_mfma16"]
_VER_NAME = {0: "conv_mfma", 1: "conv_mfma_ws", 2: "conv_mfma16", 3: "conv_wino", 4: "conv_wino4"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_conv_hbm_traffic.json")
# One body, several launch forms: the Winograd kernel in full, over the boxes the tile mask keeps, and as the rest / uniform
# pair.  The roofline's top level reports the GROUP with the largest share of the conv time (round 2 picked by kernel
# name, which put conv_upfold on top while half the time ran in the four names of the Winograd body).
KERNEL_GROUPS = {"conv_wino (all forms)": ["conv_wino", "conv_wino4", "conv_wino_masked", "conv_wino_uniform"],
                 "conv_upfold": ["conv_upfold"],
                 "conv_mfma family (direct)": ["conv_mfma", "conv_mfma_ws", "conv_mfma16"]}
CPU_BASELINE_CACHE = os.path.join(ROOT, "gpurun_out", "cpu_baseline_last.json")


def make_volume(n, device):
    """rand inside a centred ellipsoid (semi-axes 100,110,90 scaled), exact 0 outside."""
    g = torch.Generator().manual_seed(0)
    s = n / 256.0
    ax = torch.arange(n, dtype=torch.float32) - (n - 1) / 2.0
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    ell = ((zz / (100 * s)) ** 2 + (yy / (110 * s)) ** 2 + (xx / (90 * s)) ** 2) <= 1
    v = torch.rand((n, n, n), generator=g) * ell
    return v[None, None].to(device)


def make_atlas():
    """Stand-in for files/gca.mgz (utils/test_utils.py:38-43): a smooth 256^3 float volume with the conformed-space
    vox2ras matrix that file carries ([[-1,0,0,128],[0,0,1,-128],[0,-1,0,128]])."""
    import numpy as np
    ax = torch.arange(256, dtype=torch.float32)
    i, j, k = torch.meshgrid(ax, ax, ax, indexing="ij")
    vol = 110. + 60. * torch.sin(i / 17.) * torch.cos(j / 23.) + 40. * torch.sin(k / 13. + 0.5)
    aff = np.array([[-1., 0., 0., 128.], [0., 0., 1., -128.], [0., -1., 0., 128.], [0., 0., 0., 1.]])
    return vol, aff


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


def host_cpu_info():
    """CPU model, sockets, physical cores per socket, hardware threads (lscpu / /proc/cpuinfo)."""
    info = {"model": None, "sockets": None, "cores_per_socket": None, "threads": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for ln in out.splitlines():
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                info["model"] = v
            elif k == "Socket(s)":
                info["sockets"] = int(v)
            elif k == "Core(s) per socket":
                info["cores_per_socket"] = int(v)
    except Exception:
        pass
    return info


def cpu_baseline(state_dict, full, n, ranges, full_protocol=False, sess=None, fast_sess=None):
    """The CPU oracle (oracle/unet_ref.py: a torch-CPU fp32 port of the reference's path) timed on this host, rank 0
    only, the way BASELINE.md section 4 / SURVEY 8(d) set it: threads = the physical cores of one socket, per tile
    shape of the reference tiling 1 warm-up + 3 timed runs (median), and the whole volume's time extrapolated from
    count(shape) x median(shape) and labelled as such (~2 min of CPU work on 64 cores).  full_protocol=False
    (--cpu-baseline-quick): the two large shapes get one timed run each instead."""
    import numpy as np
    from oracle import unet_ref as O
    info = host_cpu_info()
    threads = info["cores_per_socket"] or torch.get_num_threads()
    prev = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    shapes = {}                                            # one class per multiset of extents: (160,80,80) stands for its
    for r in ranges:                                       # three orientations (same arithmetic, same cost)
        s = tuple(sorted((b - a for a, b in r), reverse=True))
        if s not in shapes:
            shapes[s] = [0, r]
        shapes[s][0] += 1
    order = sorted(shapes, key=lambda s: s[0] * s[1] * s[2])
    per_shape, total_s, spent = {}, 0.0, 0.0
    parity = None
    try:
        for idx, s in enumerate(order):
            cnt, r = shapes[s]
            (x0, x1), (y0, y1), (z0, z1) = r
            tile = full[:, :, x0:x1, y0:y1, z0:z1].cpu().contiguous()
            small = idx < 2 or full_protocol or len(order) <= 2
            runs = []
            with torch.no_grad():
                if small:
                    O.forward_all(tile, sd, f_maps=64, num_levels=6)                    # warm-up
                for _ in range(3 if small else 1):
                    t0 = time.perf_counter()
                    ref = O.forward_all(tile, sd, f_maps=64, num_levels=6)
                    runs.append(time.perf_counter() - t0)
                if idx == 0 and sess is not None:
                    parity = label_parity(sess, tile, ref, sd)
                    if fast_sess is not None:                       # the same tile through the passes=1 session
                        parity["fast_mode"] = label_parity(fast_sess, tile, ref, sd)
            med = float(np.median(runs))
            spent += sum(runs) * (4.0 / 3.0 if small else 1.0)
            per_shape["x".join(map(str, s))] = {"tiles": cnt, "median_s": med, "runs": len(runs),
                                                 "warmup": 1 if small else 0}
            total_s += cnt * med
    finally:
        torch.set_num_threads(prev)
    return {"value": n ** 3 / total_s, "unit": "voxels/s", "cores": int(threads), "kind": "port", "label_parity": parity,
            "cpu_model": info["model"], "sockets": info["sockets"], "cores_per_socket": info["cores_per_socket"],
            "hardware_threads": info["threads"], "per_tile_shape": per_shape,
            "extrapolated_volume_s": total_s,
            "sample": "oracle/unet_ref.py (torch-CPU fp32, all 9 heads) on one tile of each shape class of the "
                      "reference tiling (%s; orientations of a shape share its time), %s; "
                      "value = %d^3 voxels / sum(count x median) = EXTRAPOLATED whole-volume time %.1f s, not a timed "
                      "27-tile run; %.0f s of CPU work on %d threads (physical cores of one socket)"
                      % (", ".join(per_shape),
                         "1 warm-up + 3 timed runs (median) on every shape (BASELINE.md section 4)" if full_protocol else
                         "1 warm-up + 3 timed runs (median) on the two small shapes, 1 timed run on the large ones "
                         "(--cpu-baseline-quick)", n, total_s, spent, int(threads))}


ELEMENTWISE_RTOL, ELEMENTWISE_ATOL_REL = 1e-3, 1e-5


def elementwise_failures(a, b):
    """Fraction of elements with |a - b| > 1e-3 |b| + 1e-5 max|b| (the element-wise form beside the
    max-norm one, which never looks at a low-valued voxel's own relative error)."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    tol = ELEMENTWISE_RTOL * b.abs() + ELEMENTWISE_ATOL_REL * float(b.abs().max())
    return float(((a - b).abs() > tol).double().mean())


def label_parity(sess, tile, ref, sd):
    """The HIP path's labels on the tile the CPU baseline just evaluated, against that fp32 oracle result: number of
    differing voxels, and -- when there are any -- the relative gap of the two best class probabilities at those voxels
    in a float64 evaluation of the same network (a difference there is a tie fp32 cannot resolve, HISTORY.md section 1)."""
    from oracle import unet_ref as O
    dev = sess.device
    out, _ = sess.forward_fused(tile.to(dev), want_feat=False, want_seg=False)
    lab = out["label"].cpu()
    differ = lab != ref["label"]
    nd = int(differ.sum())
    res = {"tile": list(tile.shape[2:]), "n_voxels": int(lab.numel()), "label_flips_vs_fp32_oracle": nd,
           "max_fp64_gap": None, "oracle_flips_vs_fp64": None, "hip_flips_vs_fp64": None}
    worst, ew = 0.0, {}
    for k, v in ref.items():
        if k in out and k not in ("feat", "label", "segmentation"):
            a, b = out[k].cpu().double(), v.double()
            worst = max(worst, float((a - b).abs().max() / max(1e-6, float(b.abs().max()))))
            ew[k] = elementwise_failures(a, b)
    res["max_rel_err_float_maps"] = worst
    wk = max(ew, key=ew.get) if ew else None
    res["elementwise"] = {"rule": "|a - b| <= %g |b| + %g max|b| per element" % (ELEMENTWISE_RTOL, ELEMENTWISE_ATOL_REL),
                          "failing_fraction_worst_map": ew.get(wk), "worst_map": wk,
                          "failing_fraction_mean_over_maps": (sum(ew.values()) / len(ew)) if ew else None}
    if nd:
        with torch.no_grad():
            r64 = O.forward_all(tile.double(), {k: v.double() for k, v in sd.items()}, f_maps=64, num_levels=6)
        top2 = torch.topk(r64["segmentation"], 2, dim=1).values
        gap = ((top2[:, 0] - top2[:, 1]) / top2[:, 0])[:, None]
        res["max_fp64_gap"] = float(gap[differ].max())
        res["oracle_flips_vs_fp64"] = int((ref["label"] != r64["label"]).sum())
        res["hip_flips_vs_fp64"] = int((lab != r64["label"]).sum())
    return res


def fast_mode_block(sess, full, stride, win, reps=5):
    """BASELINE config 2's class on the line: the opt-in passes=1 mode (one plain f16 product per
    term instead of the three split-f16 passes; NOT the parity class) -- one 160^3 volume through forward_fused and the
    256^3 volume through the tile loop, timed, with its measured distance from the parity-mode result of this run on the
    160^3 volume (the distance from the fp32 oracle on the CPU baseline's 80^3 tile is added under label_parity.fast_mode)."""
    from brainfm_amd import test_utils as TU
    dev = sess.device
    ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
    fast = TU.InferenceSession(ga, ta, dev, state_dict={k: v for k, v in sess.model.state_dict().items()}, passes=1)
    fast.atlas = sess.atlas                               # the same resident atlas volume and inverse affine
    fast.use_graphs = sess.use_graphs
    n = full.shape[-1]
    c0 = max(0, (n - 160) // 2)
    tile = full[:, :, c0:c0 + 160, c0:c0 + 160, c0:c0 + 160].contiguous()

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ms160 = timed(lambda: fast.forward_fused(tile, want_feat=False, want_seg=False))
    ms160_parity = timed(lambda: sess.forward_fused(tile, want_feat=False, want_seg=False))
    a, _ = fast.forward_fused(tile, want_feat=False, want_seg=False)
    b, _ = sess.forward_fused(tile, want_feat=False, want_seg=False)
    worst, ew = 0.0, 0.0
    for k, v in b.items():
        if k in a and k not in ("feat", "label", "segmentation"):
            worst = max(worst, float((a[k].double() - v.double()).abs().max() / max(1e-6, float(v.double().abs().max()))))
            ew = max(ew, elementwise_failures(a[k], v))
    flips = float((a["label"] != b["label"]).double().mean())
    if fast.use_graphs:
        TU.prepare_tile_graphs(full, fast, stride, win)
    ms_vol = timed(lambda: TU.tiled_inference(full, fast, stride, win, batched=True))
    return fast, {"what": "passes=1: plain f16 products (BASELINE config 2's 'bf16' class); opt-in, not the parity class",
                  "volume_160_ms": ms160, "volume_160_parity_mode_ms": ms160_parity,
                  "volume_160_mvoxel_per_s": 160 ** 3 / ms160 / 1e3,
                  "tiled_%d_ms_per_volume" % n: ms_vol, "tiled_%d_mvoxel_per_s" % n: n ** 3 / ms_vol / 1e3,
                  "vs_parity_mode_160": {"max_rel_err_float_maps": worst, "label_flip_fraction": flips,
                                         "elementwise_failing_fraction_worst_map": ew},
                  "stated_tolerance": "1e-1 max-norm on float maps (tests/test_gpu_infer.py::test_config2_single_160_volume_all_heads)"}


def synthesis_block(dev, items=8):
    """Hot path B (BASELINE.json north_star: the data-synthesis kernels; SURVEY config 5's generator half) on the bench
    line: one BrainIDGen.__getitem__ counterpart -- 192^3 Voronoi label case, 4 augmented 160^3 samples, pathology on --
    timed per item with a device synchronisation after each (median over `items`); the same item with a resident
    pathology_prob volume and random_shape_prob 0, which runs warp -> Perlin velocity -> dopri5 -> binarize as ONE chain
    (item_with_pde_ms); and the streaming kernels of the item timed alone at 160^3 inside replayed hipGraphs, TWICE
    (VERDICT r4 #2): COLD -- every launch of a replay works on its own buffers, >= 1.2 GB per replay, far more than the
    256 MB Infinity Cache holds, so the bytes come from HBM: frac = algorithmic bytes / time / 8 TB/s -- and HOT -- the same
    buffers every launch (round 4's numbers), which live in the Infinity Cache: reported as a rate, with its fraction of
    the measured on-die rate of a streaming copy of the same size (hot_copy_TB_per_s), never of the HBM peak."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import config5_lib as C5
    from brainfm_amd import _lib as L
    from brainfm_amd import generator as G
    from brainfm_amd import generator_utils as GU
    from brainfm_amd import shapeid as SH
    N = 160
    nv = N ** 3
    st_np, st_t = np.random.get_state(), torch.random.get_rng_state()

    def time_items(ds, n_items):
        for _ in range(2):
            ds[0]
        torch.cuda.synchronize()
        ts = []
        for _ in range(n_items):
            t0 = time.perf_counter()
            ds[0]
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return ts

    np.random.seed(100)
    torch.manual_seed(100)
    ga = C5.gen_args(N)
    ds = G.build_datasets(ga, str(dev), cases=[C5.voronoi_case(7)])["all"]
    ts = time_items(ds, items)
    item = float(np.median(ts))
    ns = ga.generator.all_samples
    # the chain config 5 names: file-based lesion probability -> warp -> Perlin velocity + dopri5 -> binarize, inside the item
    np.random.seed(100)
    torch.manual_seed(100)
    ga_p = C5.gen_args(N, random_shape_prob=0.0)
    ds_p = G.build_datasets(ga_p, str(dev), cases=[C5.voronoi_case(7, pathology_prob=True)])["all"]
    ts_p = time_items(ds_p, max(4, items // 2))
    item_pde = float(np.median(ts_p))
    del ds_p

    def timed(fn, sets, replays=5):
        """us per launch of fn(j), j = 0..sets-1 once per replay, outputs kept alive inside the capture (distinct memory)."""
        keep = [fn(j) for j in range(sets)]
        del keep
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for j in range(sets):
                keep.append(fn(j))
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / (sets * replays) * 1e3
        del keep, g
        return us

    COLD_BYTES = 1.2e9

    def both(make, call, touched):
        """(hot us, cold us): `make()` -> one set of input buffers, call(set) -> output(s); touched = bytes one call moves."""
        k = int(min(96, max(8, -(-COLD_BYTES // touched))))
        pool = [make() for _ in range(k)]
        hot = timed(lambda j: call(pool[0]), 20)
        cold = timed(lambda j: call(pool[j]), k)
        del pool
        torch.cuda.empty_cache()
        return hot, cold, k

    ax = torch.arange(N, device=dev, dtype=torch.float32)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    c_i, c_j, c_k = (zz * 1.1 + 5 + 0.3 * torch.sin(yy / 9)).contiguous(), (yy * 1.1 + 6).contiguous(), (xx * 1.1 + 4).contiguous()
    del zz, yy, xx
    rnd = lambda *shape: torch.rand(*shape, device=dev)
    np.random.seed(0)
    grads = SH.perlin_gradients((2, 2, 2), (True, False, False))
    gdev = torch.from_numpy(np.ascontiguousarray(grads, dtype=np.float64)).to(dev)
    lib = L.load()

    def perlin(buf):
        L.check(lib.bfm_perlin3d(L.ptr(gdev), N, N, N, 2, 2, 2, L.ptr(buf), L.stream_ptr()), "perlin3d")
        return buf

    kern = {}
    # name -> (hot us, cold us, sets, algorithmic bytes).  The gather's coordinates are cloned per set: they are inputs too
    kern["interp_linear (fast_3D_interp_torch, 192^3 -> 160^3)"] = both(
        lambda: (rnd(192, 192, 192), c_i.clone(), c_j.clone(), c_k.clone()),
        lambda b: GU.fast_3D_interp_torch(b[0], b[1], b[2], b[3]), 4 * (192 ** 3 + 4 * nv)) + (nv * 20,)
    kern["zoom_linear (myzoom_torch 6^3x3 -> 160^3x3)"] = both(lambda: rnd(6, 6, 6, 3), lambda b: GU.myzoom_torch(b, N / 6.0),
                                                              12 * nv) + (nv * 12,)
    kern["zoom_linear (bias field 5^3 -> 160^3)"] = both(lambda: rnd(5, 5, 5), lambda b: GU.myzoom_torch(b, N / 5.0), 4 * nv) + (nv * 4,)
    kern["conv1d_axis x3 (gaussian_blur_3d sigma 1.5)"] = both(lambda: rnd(N, N, N), lambda b: GU.gaussian_blur_3d(b, [1.5, 1.5, 1.5], dev),
                                                              8 * nv) + (nv * 24,)
    kern["ew_unary gamma"] = both(lambda: rnd(N, N, N), lambda b: GU.ew_unary(L.EW_GAMMA, b, 300.0, 1.1), 8 * nv) + (nv * 8,)
    kern["ew_binary mul_exp (bias field)"] = both(lambda: (rnd(N, N, N), rnd(N, N, N)), lambda b: GU.ew_binary(L.EW_MUL_EXP, b[0], b[1]),
                                                  12 * nv) + (nv * 12,)
    kern["reduce max (partial + fold)"] = both(lambda: rnd(N, N, N), lambda b: GU.reduce_dev(1, b), 4 * nv) + (nv * 4,)
    kern["randn_philox"] = both(lambda: None, lambda b: GU.draws.randn((N, N, N), dev), 4 * nv) + (nv * 4,)
    kern["perlin3d (fp64 out)"] = both(lambda: torch.empty((N, N, N), dtype=torch.float64, device=dev), perlin, 8 * nv) + (nv * 8,)
    kern["percentile_f64 (radix select, 14 launches)"] = both(
        lambda: perlin(torch.empty((N, N, N), dtype=torch.float64, device=dev)), lambda b: SH.percentile_dev(b, 91.0), 8 * nv) + (nv * 8 * 7,)
    # the on-die ceiling the hot numbers are read against: a plain device-to-device copy of one 160^3 fp32 volume, hot
    cp_src, cp_dst = rnd(N, N, N), torch.empty(N, N, N, device=dev)
    hot_copy_us = timed(lambda j: cp_dst.copy_(cp_src), 20)
    hot_copy = 8.0 * nv / hot_copy_us / 1e6                                   # TB/s, bytes read + written
    table = {}
    for k, (hot, cold, sets, b) in kern.items():
        table[k] = {"us_cold": round(cold, 2), "us_hot": round(hot, 2), "algorithmic_MB": b / 1e6, "cold_sets_per_replay": sets,
                    "TB_per_s_cold": round(b / cold / 1e6, 3), "frac_of_8TBs_cold": round(b / cold / 1e6 / 8.0, 4),
                    "TB_per_s_hot": round(b / hot / 1e6, 3),
                    "frac_of_hot_copy_rate": round(min(1.0, b / hot / 1e6 / hot_copy), 4)}
    # the pathology shape augmentation alone (Generator/utils.py:542-560), longest case nt = max_nt
    t = torch.from_numpy(np.arange(10) * 0.1)
    pde = SH.AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict={}, BC="neumann",
                        dt=0.1, device=dev)
    _, P0 = SH.generate_shape_3d((N, N, N), [2, 2, 2], 90.0, dev)
    shp_args = ga.pathology_shape_generator
    orig_randint = np.random.randint
    np.random.randint = lambda a, b=None: shp_args.max_nt
    try:
        ode = []
        for r in range(4):
            np.random.seed(10 + r)
            pde.nfe = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            GU.augment_pathology(P0, pde, t, shp_args, dev)
            torch.cuda.synchronize()
            if r:
                ode.append((time.perf_counter() - t0, (pde.nfe - 2) // 6))
    finally:
        np.random.randint = orig_randint
    ode_ms = float(np.median([o[0] for o in ode])) * 1e3
    ode_steps = int(np.median([o[1] for o in ode]))
    ode_bytes = ode_steps * (6 * (8 + 12 + 4) + 4 * 21 + 8) * nv          # per step: y (fp64) + V + k out per stage, the k_j
    np.random.set_state(st_np)
    torch.random.set_rng_state(st_t)
    worst = min(table, key=lambda k: table[k]["frac_of_8TBs_cold"])
    return {"workload": "BrainIDGen.__getitem__ counterpart: 192^3 Voronoi label case resident in HBM -> deformation, "
                        "8 float targets + one-hot segmentation + Perlin pathology, %d augmented %d^3 samples" % (ns, N),
            "item_ms_median": item * 1e3, "item_ms_min": min(ts) * 1e3, "items_per_s": 1.0 / item,
            "generated_voxels_per_s": ns * nv / item, "samples_per_item": ns,
            "host_syncs_per_item": 2 + ns, "round3_item_ms": 168.0,
            "item_with_pde_ms": item_pde * 1e3,
            "item_with_pde": "the same item with a resident pathology_prob volume and random_shape_prob 0: the lesion map is "
                             "warped (trilinear gather), advected by a Perlin curl velocity field through dopri5 "
                             "(augment_pathology, nt drawn in [2, 10]) and binarised inside the item -- the chain BASELINE "
                             "config 5 names; median of %d items" % len(ts_p),
            "kernels_160": table, "lowest_frac_kernel": worst,
            "hot_copy_TB_per_s": round(hot_copy, 3),
            "augment_pathology_160": {"ms_median": ode_ms, "steps": ode_steps, "nt": int(shp_args.max_nt),
                                      "algorithmic_GB": ode_bytes / 1e9,
                                      "TB_per_s": ode_bytes / ode_ms / 1e9, "frac_of_8TBs": ode_bytes / ode_ms / 1e9 / 8.0,
                                      "working_set_MB": 220,
                                      "note": "dopri5 over the upwind advection PDE, step controller on the device; per "
                                              "step and voxel: 6 stages x (fp64 state 8 B + 3 velocities 12 B + k out 4 B) "
                                              "+ the 21 reads of earlier k_j (4 B) + y1 out 8 B; its ~220 MB working set "
                                              "sits at the edge of the 256 MB Infinity Cache, so the fraction is an upper "
                                              "bound on an HBM fraction (measured HBM bytes: profiles/r05_synth_hbm_traffic.json)"},
            "note": "kernel times: hipGraph replays between two HIP events on the launch stream.  cold = each launch of a replay "
                    "on its own input and output buffers (cold_sets_per_replay sets, >= 1.2 GB touched per replay: HBM); hot = 20 "
                    "launches on one set (Infinity-Cache resident; hot_copy_TB_per_s = a device-to-device copy measured the same "
                    "way, the on-die rate the hot numbers are read against); algorithmic bytes per SURVEY 8(d) (coordinates + "
                    "one touch of the source + output; fp64 where the reference computes in fp64); measured HBM bytes per "
                    "kernel: profiles/r05_synth_hbm_traffic.json"}


def config4_block(sess, dev, rank, world, use_dist, stride, win, n=512, steps=2):
    """BASELINE config 4: a 512^3 volume (the bench volume's construction, scaled), the reference tiling's 216 tiles sharded
    over the ranks, compact rows gathered to rank 0 and stitched there; timed like `value` (K volumes back to back between
    barrier + synchronize brackets, max over ranks).  Collective: every rank calls it."""
    import torch.distributed as dist
    from brainfm_amd import test_utils as TU
    full = make_volume(n, dev) if rank == 0 else None
    if world > 1:
        full = TU.broadcast_volume(full, dev, shape=(n, n, n))
    xs = {}

    def step():
        if use_dist:
            xs.clear()
            return TU.tiled_inference_distributed(full if rank == 0 else None, sess, stride, win, shape=(n, n, n), stats=xs,
                                                  broadcast=True)
        return TU.tiled_inference(full, sess, stride, win, batched=True)

    if sess.use_graphs:
        TU.prepare_tile_graphs(full, sess, stride, win, world=world, rank=rank)
    step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        acc = step()[0]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item()) / steps * 1e3
    exposed = None
    if use_dist and rank == 0 and "ev_own_done" in xs:
        exposed = max(0.0, xs["ev_own_done"].elapsed_time(xs["ev_gathers_done"]))
    nt = len(TU.tiling_ranges((n, n, n), stride, win))
    del acc, full
    torch.cuda.empty_cache()
    return {"workload": "%d^3 volume, reference tiling -> %d tiles over %d rank(s), 17 stitched keys" % (n, nt, world),
            "ms_per_volume": ms, "value": n ** 3 / ms * 1e3, "unit": "voxels/s", "steps": steps, "scaling": "strong",
            "exchange": None if not use_dist else {"bytes_sent_per_peer": xs.get("bytes_sent_per_peer"), "rounds": xs.get("rounds"),
                                                  "broadcast_bytes": xs.get("broadcast_bytes"), "exchange_exposed_ms": exposed}}


def config5_block(dev, rank, world, use_dist, size=160, items=3):
    """BASELINE config 5: per rank one generator item (192^3 Voronoi label case -> 4 augmented 160^3 samples, pathology on)
    feeding ONE training iteration of the full-width net, gradients averaged by one flat all-reduce (DDP, weak scaling,
    batch = N items).  Collective: every rank calls it (scripts/bench_config5.py is the stand-alone form)."""
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import config5_lib as C5
    st_np, st_t = np.random.get_state(), torch.random.get_rng_state()
    ds, step, ga = C5.build(dev, size, rank)

    def one():
        t0 = time.perf_counter()
        _, _, _, target, samples = ds[0]
        tg, sm = C5.collate(target, samples)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, total, ok = step.step([s["input"] for s in sm], tg, sm)
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1, total, ok

    one()                                                       # warm-up: conv variants, optimiser state, packed weights
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    tg = tt = ar = 0.0
    per_item = []
    for _ in range(items):
        if use_dist:
            dist.barrier()                                      # an item = one DDP step: the ranks start it together
        t_it = time.perf_counter()
        a, b, total, ok = one()
        per_item.append(time.perf_counter() - t_it)
        tg += a
        tt += b
        ev = step.__dict__.get("allreduce_events") or []
        if len(ev) == 3:
            ar += max(0.0, ev[0].elapsed_time(ev[1]))           # end of the backward pass -> end of the last bucket's all-reduce
    # median item (the step is submitted from Python: a host hiccup in one item must not halve the reported rate), the
    # slowest rank's
    t = torch.tensor([float(np.median(per_item))], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_med = float(t.item())
    t_all = t_med * items
    ns = ga.generator.all_samples
    ev = step.__dict__.get("allreduce_events") or []
    out = {"workload": "per rank: 192^3 label case -> generator (pathology on) -> %d augmented %d^3 samples -> one training "
                       "iteration of the 64 x 6 net (10 losses), bucketed gradient all-reduce over %d rank(s)" % (ns, size, world),
           "items_per_s_per_gpu": items / t_all, "items_per_s": world * items / t_all,
           "generated_and_trained_mvoxel_per_s": world * items * ns * size ** 3 / t_all / 1e6,
           "generator_ms_per_item": tg / items * 1e3, "iteration_ms_per_item": tt / items * 1e3,
           "allreduce_ms_per_iteration": (ar / items) if world > 1 and len(ev) == 3 else None,
           "allreduce_bytes": ev[2] if len(ev) == 3 else None,
           "allreduce_exposed_ms": (ar / items) if world > 1 and len(ev) == 3 else None,
           "allreduce_note": "gradients live in one persistent flat buffer in backward order (train.GradStore), cut into "
                             "BFM_GRAD_BUCKETS (6) buckets; each bucket is all-reduced on a communication stream as soon as the "
                             "LAST sample's backward pass completes it (DDP with accumulation: no_sync for the earlier samples); "
                             "allreduce_ms_per_iteration = allreduce_exposed_ms = time from the end of the backward pass to the "
                             "end of the last bucket's collective, i.e. what is NOT hidden",
           "item_s_each_rank0": [round(v, 4) for v in per_item], "rate_from": "median item time, max over ranks",
           "last_loss": float(total), "stepped": bool(ok), "scaling": "weak", "items_timed": items}
    del ds, step
    np.random.set_state(st_np)
    torch.random.set_rng_state(st_t)
    torch.cuda.empty_cache()
    return out


def training_block(dev, size=128, reps=2):
    """SURVEY N2 on the bench line: one training iteration of the full-width net (f_maps 64, 6 levels, the demo head set,
    16 losses) on one size^3 sample of synthetic data -- forward, losses, backward, per-parameter clip, AdamW, re-packed
    weights -- timed with HIP events on torch's stream (scripts/bench_train.py is the stand-alone form, DDP included)."""
    from brainfm_amd import backward as BW
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
    torch.manual_seed(1)
    s = TU.InferenceSession(ga, ta, dev, passes=3)
    tail = s.model.head.tail(s.engine)
    names = ["T1", "T1_grad", "T2", "T2_grad", "FLAIR", "FLAIR_grad", "CT", "CT_grad", "seg_ce", "seg_dice", "distance",
             "bias_field_log", "registration", "registration_grad", "SR", "SR_grad"]
    ns = tail.desc.n_seg
    step = TR.TrainStep(s.engine, tail, names, {"loss_" + n: 1.0 for n in names}, torch.full((ns,), 1.0 / ns), 4, lr=1e-4)
    g = torch.Generator().manual_seed(0)
    dims = (size,) * 3
    xs = [torch.rand((1, 1) + dims, generator=g).to(dev)]
    lab = torch.randint(0, ns, (1,) + dims, generator=g)
    target = {"segmentation": torch.nn.functional.one_hot(lab, ns).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)}
    for k in ("T1", "T2", "FLAIR", "CT"):
        target[k] = torch.rand((1, 1) + dims, generator=g).to(dev)
    target["distance"] = torch.randn((1, 4) + dims, generator=g).to(dev)
    target["registration"] = torch.randn((1, 3) + dims, generator=g).to(dev)
    samples = [{"bias_field_log": torch.randn((1, 1) + dims, generator=g).to(dev) * 0.3,
                "high_res_residual": torch.randn((1, 1) + dims, generator=g).to(dev) * 0.2}]

    def forward_only():
        feats, _ = BW.backbone_forward_train(s.engine, s.engine.to_cl(xs[0]), dims)
        tail.run_raw(feats[-1][0], dims)

    forward_only()                                   # tunes the conv variants, packs the weights
    torch.cuda.synchronize()
    tf = tb = to = 0.0
    total, ok = float("nan"), False
    for r in range(reps + 1):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        forward_only()
        e[1].record()
        _, total, grads = step.loss_and_grads(xs, target, samples)
        e[2].record()
        ok, _ = step.apply(grads)
        e[3].record()
        torch.cuda.synchronize()
        if r == 0:
            continue                                  # first pass allocates the optimiser state and the dgrad packs
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
        to += e[2].elapsed_time(e[3])
    tf, tb, to = tf / reps, tb / reps, to / reps
    return {"what": "one training iteration, full-width net, one %d^3 sample, 16 losses, synthetic data" % size,
            "iteration_ms": round(tb + to, 2), "forward_alone_ms": round(tf, 2), "forward_losses_backward_ms": round(tb, 2),
            "clip_adamw_repack_ms": round(to, 2), "mvoxel_per_s": round(size ** 3 / (tb + to) / 1e3, 2),
            "loss": round(float(total), 4), "stepped": bool(ok), "reps": reps, "dtype": "f32 (split-f16 x3 matrix-core products)"}


def launch_ranks(n, argv):
    """--gpus N without a launcher: one child per device, started before this process makes any GPU call (a process
    that has initialised the GPU must not be replaced or forked into ranks).  Relays rank 0's JSON line."""
    share = os.environ.get("BFM_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()                      # counts devices without initialising the GPU
    if not share and have < n:
        print("bench.py: --gpus %d but only %d device(s) visible (BFM_BENCH_SHARE_GPU=1 BFM_BENCH_BACKEND=gloo runs "
              "the N-rank path on one device as a dry run)" % (n, have), file=sys.stderr)
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            env.setdefault("BFM_BENCH_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(rcs) or line is None:
        print("bench.py: rank exit codes %s" % rcs, file=sys.stderr)
        sys.exit(1)
    print(line, flush=True)


def kernel_of(p):
    tag, cfg = p[5][0], p[5][4]
    if tag.endswith("[masked]"):
        return "conv_wino_masked"
    if tag.endswith("[uniform]"):
        return "conv_wino_uniform"
    return "conv_upfold" if tag.endswith("up") else _VER_NAME.get(cfg[6], "conv_mfma")


def issue_factor(p, passes):
    """fp16 MFMA FLOPs a conv launch ISSUES per algorithmic FLOP (2*27*Cin*Cout per voxel): the split passes times what the
    kernel's algebra saves -- Winograd F(2,3) along x 18/27 tap-rows, F(4,3) 13.5/27, the up-folded form 8 taps of 27
    (profiles/r06_conv_wino4d_pmc.txt, r06_conv_upfold_pmc.txt: SQ_INSTS_MFMA x 32 768 agrees)."""
    tag, cfg = p[5][0], p[5][4]
    if tag.endswith("up"):
        return passes * 8.0 / 27.0
    return passes * {3: 18.0 / 27.0, 4: 13.5 / 27.0}.get(cfg[6], 1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--passes", type=int, default=3, help="3 = fp32-grade split-f16 MFMA (parity mode), 1 = fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-quick", action="store_true",
                    help="1 timed run instead of warm-up + 3 on the two large tile shapes (the default is the full protocol of "
                         "BASELINE.md section 4 on all four shapes, ~2 min on 64 cores)")
    ap.add_argument("--no-atlas", action="store_true", help="16 stitched keys (without the deformed atlas)")
    ap.add_argument("--layer-table", default=None, help="write the per-layer conv timing table of the instrumented pass here")
    ap.add_argument("--dist-path", action="store_true",
                    help="run the multi-GPU code path (pack, RCCL gather, root accumulation) even with one rank")
    ap.add_argument("--no-dense-check", action="store_true", help="skip the extra steps on a volume without zeros")
    ap.add_argument("--no-graphs", action="store_true", help="submit every kernel from python instead of hipGraph replay")
    ap.add_argument("--no-synthesis", action="store_true", help="skip the synthesis block (hot path B) of the line")
    ap.add_argument("--no-training", action="store_true", help="skip the training block (SURVEY N2) of the line")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the passes=1 block (BASELINE config 2's class) of the line")
    ap.add_argument("--no-config4", action="store_true", help="skip the 512^3 block (BASELINE config 4) of the line")
    ap.add_argument("--no-config5", action="store_true", help="skip the generator + training-iteration block (BASELINE config 5)")
    ap.add_argument("--roofline-reps", type=int, default=3,
                    help="back-to-back launches per HIP-event bracket in the instrumented conv pass")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:])
        return

    import torch.distributed as dist
    from brainfm_amd import test_utils as TU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is what runs" % (args.gpus, world),
              file=sys.stderr)
    # dry run of the N > 1 path on a one-GPU box: BFM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # BFM_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks per device); never set for a measurement
    share = os.environ.get("BFM_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    elif torch.cuda.device_count() <= local:
        print("bench.py: rank %d needs cuda:%d, %d device(s) visible" % (rank, local, torch.cuda.device_count()),
              file=sys.stderr)
        sys.exit(2)
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
    if not args.no_atlas:
        sess.set_atlas(*make_atlas())
    # Only rank 0 holds the volume (the reference reads one file in one process, scripts/demo_test.py:71): the peers get it
    # by broadcast -- here once for the untimed setup, and inside every timed step (tiled_inference_distributed).
    full = make_volume(n, dev) if rank == 0 else None
    if world > 1:
        if backend == "nccl":
            assert dist.get_world_size() == world, "RCCL world size %d != WORLD_SIZE %d" % (dist.get_world_size(), world)
        full = TU.broadcast_volume(full, dev, shape=(n, n, n))
    stride, win = [80] * 3, [160] * 3
    ranges = TU.tiling_ranges((n, n, n), stride, win)
    eng = sess.engine
    sess.use_graphs = not args.no_graphs

    xstats = {}                                            # what the last distributed step's exchange moved

    def step(vol=None):
        vol = full if vol is None else vol
        if use_dist:
            xstats.clear()
            return TU.tiled_inference_distributed(vol if rank == 0 else None, sess, stride, win, shape=(n, n, n),
                                                  stats=xstats, broadcast=True)
        return TU.tiled_inference(vol, sess, stride, win, batched=True)      # eager (--no-graphs) runs the same batches

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
        acc = step()[0]
        if debug:
            torch.cuda.synchronize()
            print("rank %d step %.1f ms" % (rank, 1e3 * (time.perf_counter() - ts)), file=sys.stderr, flush=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exposed_ms = None
    if use_dist and rank == 0 and "ev_own_done" in xstats:
        exposed_ms = max(0.0, xstats["ev_own_done"].elapsed_time(xstats["ev_gathers_done"]))
    exchange = None
    if use_dist:
        exchange = {k: xstats.get(k) for k in ("world", "rounds", "round_bytes_per_peer", "bytes_sent_per_peer",
                                               "compact_rows", "broadcast_bytes")}
        exchange["exchange_exposed_ms"] = exposed_ms
        exchange["note"] = ("last timed step; broadcast_bytes = the volume, rank 0 to every peer inside the step; "
                            "bytes_sent_per_peer = the padded rows every peer ships to rank 0 over its rounds; "
                            "exchange_exposed_ms = on rank 0's stream, from its own tiles being done to the last gather "
                            "having arrived (what the round-wise gathers did not hide)")
    n_keys = len(acc) if acc is not None else None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # per-volume latency (volume resident -> all stitched outputs resident on rank 0): the same step with a
    # device sync (and a barrier) after every volume; median.  `value` above is the pipelined rate of K volumes back to back.
    lat = []
    for _ in range(min(args.steps, 10)):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        lat.append(time.perf_counter() - ts)
    lat.sort()
    lat_med = torch.tensor([lat[len(lat) // 2] if lat else 0.0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(lat_med, op=dist.ReduceOp.MAX)
    lat_med = float(lat_med.item())

    # The same build, graphs and step on a volume WITHOUT an exact-zero background (uniform noise everywhere): no box, run or
    # voxel is left out, every shortcut that depends on the data (engine.mask_skip, uniform_skip, the compact rows) finds
    # nothing to skip.  Reported next to `value` so that the data-independent rate is on the line too.
    dense_ms = None
    if not args.no_dense_check:
        dense = None
        if rank == 0:
            g = torch.Generator(device="cpu").manual_seed(5)
            dense = (torch.rand((1, 1, n, n, n), generator=g) + 0.05).to(dev)
        for _ in range(2):
            step(dense)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        td = time.perf_counter()
        nd = max(2, min(args.steps, 5))
        for _ in range(nd):
            step(dense)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        tdm = torch.tensor([time.perf_counter() - td], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tdm, op=dist.ReduceOp.MAX)
        dense_ms = float(tdm.item()) / nd * 1e3
        del dense

    # dominant kernels: the conv family.  The timed region replays hipGraphs, which HIP events cannot bracket per kernel, so
    # the same step is run once more eagerly right after it with every conv launch issued `reps` times back to back
    # inside one HIP event pair on the launch stream (idempotent; back-to-back so the bracket holds kernel time rather
    # than python submission gaps).  Launch duration = bracket / reps.
    peak = 2500.0                                            # dense f16 MFMA, MI355X_MICROARCH.md

    def conv_profile(vol, table=None):
        """Instrumented eager replay of one step on `vol` -> (per-kernel dict, family dict, launches per kernel of this rank)."""
        eng.prof = []
        eng.prof_reps = args.roofline_reps
        g = sess.use_graphs
        sess.use_graphs = False
        step(vol)
        torch.cuda.synchronize()
        sess.use_graphs = g
        prof = eng.prof
        eng.prof = None
        per = {k: [0.0, 0.0, 0.0, 0.0, 0.0] for k in CONV_KERNELS}  # ms, flops, bytes, launches, issued fp16 MFMA flops
        for p in prof:
            e = per[kernel_of(p)]
            e[0] += p[0].elapsed_time(p[1]) / p[4]
            e[1] += p[2]
            e[2] += p[3]
            e[3] += 1
            e[4] += p[2] * issue_factor(p, args.passes)
        if table and rank == 0:
            tab = {}
            for p in prof:
                e = tab.setdefault((kernel_of(p),) + p[5], [0, 0.0, 0.0])
                e[0] += 1
                e[1] += p[0].elapsed_time(p[1]) / p[4]
                e[2] += p[2]
            with open(table, "w") as f:
                f.write("# conv launches of one step grouped by (kernel, layer, Cin, Cout, dims, plan[WM,WN,TD,TH,TW,splitk,ver,0])\n")
                f.write("%-13s %-12s %5s %5s %-16s %-28s %5s %10s %8s %7s\n" % ("kernel", "layer", "cin", "cout", "dims", "plan",
                                                                            "n", "ms_total", "us_avg", "TF/s"))
                for key, (cnt, ms, fl) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                    f.write("%-13s %-12s %5d %5d %-16s %-28s %5d %10.3f %8.1f %7.1f\n" % (
                        key[0], key[1], key[2], key[3], "x".join(map(str, key[4])), ",".join(map(str, key[5])), cnt, ms,
                        ms * 1e3 / cnt, fl / (ms * 1e-3) / 1e12))
        mine = {k: int(e[3]) for k, e in per.items() if e[3] > 0}
        mat = torch.tensor([per[k] for k in CONV_KERNELS], device=dev, dtype=torch.float64)
        if world > 1:
            # per kernel: FLOPs, bytes and launches summed over the ranks, time = the SLOWEST rank's (what the step waits for)
            allm = [torch.zeros_like(mat) for _ in range(world)]
            dist.all_gather(allm, mat)
            st = torch.stack(allm)
            mat = st.sum(0)
            mat[:, 0] = st[:, :, 0].max(0).values
            rank_ms = [float(v) for v in st[:, :, 0].sum(1).tolist()]
        else:
            rank_ms = [float(mat[:, 0].sum())]
        per = {k: [float(v) for v in mat[i].tolist()] for i, k in enumerate(CONV_KERNELS)}
        k_ms = max(rank_ms)                                      # conv kernel time of the slowest rank
        k_fl = sum(e[1] for e in per.values())
        kernels = {}
        for k, (ms, fl, by, cnt, iss) in per.items():
            if cnt <= 0:
                continue
            ach = fl / (ms * 1e-3) / 1e12 / max(world, 1)       # per GPU: the ranks' FLOPs over the slowest rank's time
            kernels[k] = {"launches_per_step": int(cnt), "ms_per_step": ms, "avg_launch_us": ms * 1e3 * max(world, 1) / cnt,
                          "achieved": ach, "frac": ach / peak, "algorithmic_bytes_per_launch": by / cnt,
                          "mfma_issue_frac": iss / (ms * 1e-3) / 1e12 / max(world, 1) / peak}
        groups = {}
        for gname, members in KERNEL_GROUPS.items():
            ms = sum(per[m][0] for m in members)
            fl = sum(per[m][1] for m in members)
            cnt = sum(per[m][3] for m in members)
            if cnt <= 0:
                continue
            ach = fl / (ms * 1e-3) / 1e12 / max(world, 1)
            iss = sum(per[m][4] for m in members)
            groups[gname] = {"kernels": [m for m in members if per[m][3] > 0], "launches_per_step": int(cnt),
                             "ms_per_step": ms, "achieved": ach, "frac": ach / peak,
                             "mfma_issue_frac": iss / (ms * 1e-3) / 1e12 / max(world, 1) / peak,
                             "share_of_conv_time": ms / sum(per[m][0] for m in CONV_KERNELS)}
        fam_ach = k_fl / (k_ms * 1e-3) / 1e12 / max(world, 1) if k_ms > 0 else 0.0
        k_iss = sum(e[4] for e in per.values())
        family = {"achieved": fam_ach, "frac": fam_ach / peak, "kernel_ms_per_step": k_ms,
                  "mfma_issue_frac": (k_iss / (k_ms * 1e-3) / 1e12 / max(world, 1) / peak) if k_ms > 0 else 0.0,
                  "kernel_ms_per_rank": rank_ms if world > 1 else None,
                  "launches_per_step": int(sum(e[3] for e in per.values())),
                  "algorithmic_bytes_per_step": sum(e[2] for e in per.values())}
        return kernels, groups, family, mine

    # BASELINE configs 4 and 5 on the same line, every rank taking part: a 512^3 volume through the same
    # tile flow (216 tiles sharded over the ranks, compact rows gathered to rank 0), and the generator feeding one training
    # iteration per rank with the flat gradient all-reduce.  Both are deterministic in what they launch and collective in
    # the same places on every rank; neither is `value`.
    config4 = config5 = None
    if not args.no_config4 and n == 256:
        config4 = config4_block(sess, dev, rank, world, use_dist, stride, win)
    if not args.no_config5:
        config5 = config5_block(dev, rank, world, use_dist)

    kernels, groups, family, mine = {}, {}, {}, {}
    dense_family = None
    if args.roofline_reps > 0:                      # 0: skip (used for rocprofv3 runs that should hold the timed steps only)
        kernels, groups, family, mine = conv_profile(None, args.layer_table)
        if not args.no_dense_check:
            g = torch.Generator(device="cpu").manual_seed(5)
            dvol = (torch.rand((1, 1, n, n, n), generator=g) + 0.05).to(dev) if rank == 0 else None
            _, dgroups, dense_family, _ = conv_profile(dvol)
            dense_family["per_group"] = {k: {"ms_per_step": v["ms_per_step"], "achieved": v["achieved"], "frac": v["frac"]}
                                         for k, v in dgroups.items()}
            del dvol
    # HBM traffic per kernel: rocprofv3 PMC passes cannot run inside this process; the committed summary of the same
    # workload (scripts/pmc_traffic.py: FETCH_SIZE x2 + WRITE_SIZE, separate passes, per MI355X_MICROARCH.md) is used --
    # and refused when the launch counts it recorded per kernel differ from this run's (a stale file)
    traffic_tab, traffic_note = {}, None
    if os.path.exists(TRAFFIC_FILE) and world == 1:
        try:
            tj = json.load(open(TRAFFIC_FILE))
            theirs = {k: int(v["launches"]) for k, v in tj["kernels"].items() if k in CONV_KERNELS}
            if mine == theirs:
                traffic_tab = {k: v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}
                traffic_note = os.path.relpath(TRAFFIC_FILE, ROOT) + ": " + tj["source"]
            else:
                traffic_note = ("%s refused: it recorded launches %s, this run has %s -- regenerate it "
                                "(scripts/pmc_traffic.py)" % (os.path.relpath(TRAFFIC_FILE, ROOT), theirs, mine))
        except Exception as e:                                # noqa: BLE001
            traffic_note = "unreadable traffic file: %r" % (e,)
    elif world > 1:
        traffic_note = "not reported for N > 1: the PMC summary is a one-rank run and the launches per rank differ"
    for k in kernels:
        kernels[k]["traffic"] = traffic_tab.get(k)
    for gname, gv in groups.items():                           # a group's traffic: launch-weighted over its kernels
        tb = [(kernels[m]["traffic"], kernels[m]["launches_per_step"]) for m in gv["kernels"]]
        gv["traffic_per_launch"] = (sum(t * c for t, c in tb) / sum(c for _, c in tb)) if tb and all(t is not None for t, _ in tb) else None
        gv["algorithmic_bytes_per_launch"] = sum(kernels[m]["algorithmic_bytes_per_launch"] * kernels[m]["launches_per_step"]
                                                 for m in gv["kernels"]) / gv["launches_per_step"]
        gv["avg_launch_us"] = gv["ms_per_step"] * 1e3 * max(world, 1) / gv["launches_per_step"]
    dominant = max(groups, key=lambda k: groups[k]["ms_per_step"]) if groups else None
    if rank == 0:
        tile_vox = sum(TU.tile_cost(r) for r in ranges)
        flops_step = sum(conv_flops_tile([r[a][1] - r[a][0] for a in range(3)]) for r in ranges)
        # work the tile loop's mask makes unnecessary (engine.mask_skip): boxes of the last convolution and runs of 64
        # head voxels whose tile input is all zero.  Counted here on the host from the same tiles.
        fm0, n_head = eng.fm[0], 69                           # conv_flops_tile counts the same 69 head channels
        inside = conv_vox = head_vox = 0
        for r in ranges:
            t = full[0, 0, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]].contiguous()
            inside += int((t != 0).sum().item())
            if eng.mask_skip:
                conv_vox += eng.masked_voxels(t, tuple(t.shape))
                f = t.reshape(-1)
                f = torch.nn.functional.pad(f, (0, -f.numel() % 64)).reshape(-1, 64)
                head_vox += min(int((f != 0).any(dim=1).sum().item()) * 64, t.numel())
            else:
                conv_vox += t.numel()
                head_vox += t.numel()
        computed_step = flops_step - (tile_vox - conv_vox) * 2.0 * 27 * fm0 * fm0 - (tile_vox - head_vox) * 2.0 * fm0 * n_head
        # boxes of the two layers that read the first activations where the input is constant around them run a quarter of
        # their products (engine.uniform_skip)
        uni = {}
        if eng.uniform_skip:
            chans = {("enc", 0, 1): (fm0 // 2, fm0), ("dec", 0): (fm0, fm0), ("enc", 1, 0): (fm0, fm0),
                     ("enc", 1, 1): (fm0, 2 * fm0), ("dec", 1): (2 * fm0, 2 * fm0)}
            for r in ranges:
                t = full[0, 0, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]].contiguous()
                for key, rad in eng.UNIFORM_RADIUS.items():
                    lvl = key[1]
                    fl = eng.uniform_flags(t.unsqueeze(-1), tuple(t.shape), rad, lvl)
                    if fl is None:
                        continue
                    ld = [v >> lvl for v in t.shape]
                    nb = eng.lib.bfm_conv3x3x3_wino_rows(ld[0], ld[1], ld[2], eng.passes)
                    f = fl[:nb]
                    reused = int((f != 0).sum().item()) - int(torch.unique(f[f != 0]).numel())
                    v = max(reused, 0) / float(nb) * ld[0] * ld[1] * ld[2]    # all flagged boxes but one per class
                    uni[key] = uni.get(key, 0.0) + v
                    computed_step -= 2.0 * 27 * chans[key][0] * chans[key][1] * v
        uni2, uni3 = uni.get(("enc", 0, 1), 0.0), uni.get(("dec", 0), 0.0)
        uni_l1 = sum(v for k, v in uni.items() if k[1] == 1) / max(sum(1 for k in uni if k[1] == 1), 1)
        dk = groups.get(dominant, {})
        line = {
            "metric": "voxels/sec whole-volume multi-task inference, 256^3 tiled",
            "value": n ** 3 * args.steps / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f16x3-split (fp32-grade)" if args.passes == 3 else "f16", "data": "synthetic",
            "rccl_ranks": world if (use_dist and backend == "nccl") else 0,
            "backend": backend if use_dist else None,
            "devices": ["cuda:%d" % (0 if share else r) for r in range(world)],
            "timing": "value = K volumes back to back between two barrier+synchronize brackets (no sync between volumes, "
                      "max over ranks); latency_ms_median = one volume at a time, synchronised after each",
            "latency_ms_median": lat_med * 1e3,
            "dense_volume": None if dense_ms is None else {
                "ms_per_step": dense_ms, "value": n ** 3 / dense_ms * 1e3, "unit": "voxels/s",
                "conv_family": dense_family,
                "note": "the same build and graphs on a volume of uniform noise without a zero background: nothing is "
                        "skipped (config.tile_mask describes what the headline volume, SURVEY config 3's ellipsoid with "
                        "exact zeros outside, lets the exact shortcuts leave out); conv_family = the instrumented conv "
                        "pass on this volume: the data-independent fraction of the MFMA peak"},
            "exchange": exchange,
            "config": {"workload": "%d^3 volume, reference tiling win160/stride80 -> %d tiles, UNet3D f64 x6 levels, "
                                   "9 heads (69 ch), fused tail + deformed atlas + on-device stitch of %s keys"
                                   % (n, len(ranges), n_keys),
                       "stitched_keys": n_keys,
                       "tile_voxels_per_step": tile_vox, "tile_voxels_per_s": tile_vox * args.steps / dt,
                       "algorithmic_tflop_per_step": flops_step / 1e12,
                       "computed_tflop_per_step": computed_step / 1e12,
                       "end_to_end_tflops": computed_step * args.steps / dt / 1e12,
                       "tile_mask": {"skip": bool(eng.mask_skip), "tile_voxels_inside_mask_frac": inside / tile_vox,
                                     "last_conv_voxels_computed_frac": conv_vox / tile_vox,
                                     "head_voxels_computed_frac": head_vox / tile_vox,
                                     "uniform_box_frac_enc0_conv2": uni2 / tile_vox,
                                     "uniform_box_frac_dec4_conv1_skip": uni3 / tile_vox,
                                     "uniform_box_frac_level1_layers": uni_l1 / (tile_vox / 8.0),
                                     "note": "the tile loop keeps out * (tile input != 0) (scripts/demo_test.py:88-100); "
                                             "the last convolution (4x4x16 boxes) and the heads (runs of 64 voxels) leave "
                                             "out what holds no non-zero input; stitched results are bit-identical "
                                             "(BFM_MASK_SKIP=0 computes everything); algorithmic_tflop_per_step is the "
                                             "reference's dense work, computed_tflop_per_step and end_to_end_tflops what "
                                             "was evaluated"},
                       "mfma_passes": args.passes,
                       "submission": "hipGraph replay per tile shape" if sess.use_graphs else "eager",
                       "parallelism": "tiles sharded over %d rank(s), gather to rank 0" % world,
                       "tiles_in_flight_per_gpu": sess.lanes if sess.use_graphs else 1},
            "roofline": {"bound": "mfma", "kernel": dominant, "kernel_members": dk.get("kernels"),
                         "achieved": dk.get("achieved"), "peak": peak, "unit": "TFLOP/s", "frac": dk.get("frac"),
                         "mfma_issue_frac": dk.get("mfma_issue_frac"),
                         "mfma_issue_note": "issued fp16 MFMA FLOPs / peak: the algorithmic FLOPs times the split passes times the "
                                            "kernel's reduction (F(2,3) 18/27, F(4,3) 13.5/27, up-fold 8/27, direct 1); what the "
                                            "matrix pipe sustains on random operands fed from LDS + L2 is 0.56 of the nameplate "
                                            "(1 410 TFLOP/s, profiles/r02_mfma_sustained_micro.txt)",
                         "traffic": dk.get("traffic_per_launch"), "traffic_source": traffic_note,
                         "avg_launch_us": dk.get("avg_launch_us"),
                         "algorithmic_bytes_per_launch": dk.get("algorithmic_bytes_per_launch"),
                         "share_of_conv_time": dk.get("share_of_conv_time"),
                         "per_group": groups, "per_kernel": kernels, "conv_family": family,
                         "note": "top level = the GROUP of conv kernels with the largest share of the step's conv time (the "
                                 "Winograd body runs under three launch names), every group under per_group, every kernel "
                                 "name under per_kernel, the whole family under conv_family; achieved = algorithmic conv "
                                 "FLOPs of the boxes a launch evaluates (2*27*Cin*Cout*voxels, also for Winograd / up-folded "
                                 "layers; config.tile_mask says what the masked / uniform forms leave out) / summed launch "
                                 "durations; durations from HIP events around %d back-to-back launches of each conv in an "
                                 "instrumented eager replay of the step right after the timed region; the kernels issue "
                                 "up to %dx the algorithmic FLOPs in f16 MFMA; conv_wino_uniform is a pair of kernels per "
                                 "launch: conv_wino_rest (matrix-bound, the unflagged boxes) + a kernel that streams one "
                                 "box's result to its class mates (HBM-bound); with N > 1 ranks a kernel's time is the "
                                 "slowest rank's, FLOPs and launches are summed over the ranks" % (args.roofline_reps, args.passes)},
        }
        line["synthesis"] = None
        if not args.no_synthesis and world == 1:
            try:
                line["synthesis"] = synthesis_block(dev)
            except Exception as e:                            # noqa: BLE001 -- never lose the headline line to the extra block
                line["synthesis"] = {"error": repr(e)}
        line["training"] = None
        if not args.no_training and world == 1:
            try:
                line["training"] = training_block(dev)
            except Exception as e:                            # noqa: BLE001
                line["training"] = {"error": repr(e)}
        line["config4"] = config4
        line["config5"] = config5
        line["fast_mode"] = None
        fast_sess = None
        if not args.no_fast_mode and world == 1 and args.passes == 3:
            try:
                fast_sess, line["fast_mode"] = fast_mode_block(sess, full, stride, win)
            except Exception as e:                            # noqa: BLE001
                line["fast_mode"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            sd = {k: v for k, v in sess.model.state_dict().items()}
            line["cpu_baseline"] = cpu_baseline(sd, full, n, ranges, not args.cpu_baseline_quick, sess, fast_sess)
            line["label_parity"] = line["cpu_baseline"].pop("label_parity")
            try:                                              # for the N > 1 lines of the same box (rank 0 at N = 1 only)
                os.makedirs(os.path.dirname(CPU_BASELINE_CACHE), exist_ok=True)
                json.dump({"size": n, "cpu_baseline": line["cpu_baseline"], "label_parity": line["label_parity"]},
                          open(CPU_BASELINE_CACHE, "w"))
            except OSError:
                pass
        else:
            line["cpu_baseline"] = None
            if world > 1 and os.path.exists(CPU_BASELINE_CACHE):
                try:
                    cj = json.load(open(CPU_BASELINE_CACHE))
                    if cj.get("size") == n and cj["cpu_baseline"].get("cpu_model") == host_cpu_info()["model"]:
                        line["cpu_baseline"] = dict(cj["cpu_baseline"], copied_from="the N = 1 run of this box (%s)"
                                                    % os.path.relpath(CPU_BASELINE_CACHE, ROOT))
                        line["label_parity"] = cj.get("label_parity")
                except Exception:                             # noqa: BLE001
                    pass
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
