"""Helper of tests/test_gpu_infer.py::test_distributed_path_two_ranks_share_one_gpu: two gloo ranks that share cuda:0 run
the real multi-GPU code path (agreement on the conv variants, tile sharding, lanes, graph replay, pack kernels, round-wise
asynchronous gathers, ordered accumulation on rank 0) and rank 0 compares with the single-process result bit for bit.  Prints OK / FAIL."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port, q):
    from conftest import load_npz, sd_from_npz
    from brainfm_amd import test_utils as TU
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    d = load_npz("infer_tiled.npz")
    f_maps, levels, groups, stride, win = [int(v) for v in d["cfg"]]
    ga, ta = TU.default_inference_args(f_maps=f_maps, num_levels=levels, num_groups=groups)
    s = TU.InferenceSession(ga, ta, dev, state_dict=sd_from_npz(d), passes=3)
    s.set_atlas(d["atlas"], d["atlas_aff"])                      # 17 stitched keys, as scripts/demo_test.py:102-119
    full = torch.from_numpy(d["full"]).to(dev)
    ref = None
    if rank == 0:
        ref, _, _ = TU.tiled_inference(full, s, [stride] * 3, [win] * 3, graphs=False)
        ref = {k: v.clone() for k, v in ref.items()}
        assert len(ref) == 17 and "deformed_atlas" in ref
    s.use_graphs = True
    ok = True
    for rep in range(4):
        acc, _, _ = TU.tiled_inference_distributed(full, s, [stride] * 3, [win] * 3)
        if rank == 0:
            bad = [k for k in ref if not torch.equal(acc[k], ref[k])]
            print("rep", rep, "bad", bad, flush=True)
            ok = ok and not bad
    # a 64-wide net (MFMA convs, timed variants): rank 1 starts from deliberately different variant choices; the
    # agreement step (test_utils.agree_on_conv_variants) must bring it onto rank 0's table, and the volume must again
    # equal the single-process one bit for bit
    from oracle import unet_ref as O
    from brainfm_amd import engine as E
    sd = O.random_state_dict(1, 64, 3, seed=5)
    ga, ta = TU.default_inference_args(f_maps=64, num_levels=3)
    s2 = TU.InferenceSession(ga, ta, dev, state_dict=sd, passes=3)
    g = torch.Generator().manual_seed(21)
    full2 = torch.rand(1, 1, 48, 40, 56, generator=g).to(dev)
    ref2 = None
    if rank == 0:
        ref2, _, _ = TU.tiled_inference(full2, s2, [16] * 3, [32] * 3, graphs=False)
        ref2 = {k: v.clone() for k, v in ref2.items()}
    elif rank == 1:
        TU._run_tile(s2, full2[:, :, :32, :32, :32], raw=True)            # times the variants here ...
        mine = s2.engine.conv_choices()
        s2.engine.adopt_conv_choices({k: (1 if v == 0 else 0) for k, v in mine.items()})   # ... and takes other ones
    s2.use_graphs = True
    for rep in range(3):
        acc, _, _ = TU.tiled_inference_distributed(full2, s2, [16] * 3, [32] * 3)
        if rank == 0:
            bad = [k for k in ref2 if not torch.equal(acc[k], ref2[k])]
            print("wide net rep", rep, "bad", bad, flush=True)
            ok = ok and not bad
    # the same net on a volume whose upper part is exactly zero: tiles that keep nothing at all (empty rows in the compact
    # exchange), tiles that keep a few columns, boxes that see only the constant background (the uniform shortcut)
    full3 = full2.clone()
    full3[:, :, :, :, 24:] = 0
    full3[:, :, 30:, :, :] = 0
    ref3 = None
    if rank == 0:
        s2.use_graphs = False
        ref3, rng3, _ = TU.tiled_inference(full3, s2, [16] * 3, [32] * 3, graphs=False)
        ref3 = {k: v.clone() for k, v in ref3.items()}
        empty = [r for r in rng3 if not bool((full3[:, :, r[0][0]:r[0][1], r[1][0]:r[1][1], r[2][0]:r[2][1]] != 0).any())]
        assert empty, "the volume should have a tile without any input"
        s2.use_graphs = True
    for rep in range(2):
        acc, _, _ = TU.tiled_inference_distributed(full3, s2, [16] * 3, [32] * 3)
        if rank == 0:
            bad = [k for k in ref3 if not torch.equal(acc[k], ref3[k])]
            print("half-empty volume rep", rep, "bad", bad, flush=True)
            ok = ok and not bad
    tables = [None] * world
    dist.all_gather_object(tables, s2.engine.conv_choices())
    if rank == 0:
        print("conv variants per rank", [{v: list(t.values()).count(v) for v in set(t.values())} for t in tables], flush=True)
        ok = ok and len(tables[0]) > 0 and all(all(t.get(k) == v for k, v in tables[0].items()) for t in tables[1:])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(ok)

if __name__ == "__main__":
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    ok = (not q.empty()) and q.get(timeout=5)
    print("exit codes", [p.exitcode for p in ps])
    print("OK" if ok and all(p.exitcode == 0 for p in ps) else "FAIL")
