"""Helper of tests/test_gpu_infer.py::test_distributed_path_two_ranks_share_one_gpu: two gloo ranks that share cuda:0 run
the real multi-GPU code path (tile sharding, lanes, graph replay, pack kernels, round-wise asynchronous gathers, ordered
accumulation on rank 0) and rank 0 compares with the single-process result bit for bit.  Prints OK / FAIL."""
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
    full = torch.from_numpy(d["full"]).to(dev)
    ref = None
    if rank == 0:
        ref, _, _ = TU.tiled_inference(full, s, [stride] * 3, [win] * 3, graphs=False)
        ref = {k: v.clone() for k, v in ref.items()}
    s.use_graphs = True
    ok = True
    for rep in range(4):
        acc, _, _ = TU.tiled_inference_distributed(full, s, [stride] * 3, [win] * 3)
        if rank == 0:
            bad = [k for k in ref if not torch.equal(acc[k], ref[k])]
            print("rep", rep, "bad", bad, flush=True)
            ok = ok and not bad
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(ok)

if __name__ == "__main__":
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    for p in ps: p.join(300)
    ok = (not q.empty()) and q.get(timeout=5)
    print("exit codes", [p.exitcode for p in ps])
    print("OK" if ok and all(p.exitcode == 0 for p in ps) else "FAIL")
