"""Helper of tests/test_gpu_train.py::test_two_ranks_training_step_equals_averaged_gradients: two gloo ranks that share
cuda:0 run TrainStep.step on DIFFERENT samples (DDP semantics: gradients averaged over ranks by one flat all-reduce, then
identical AdamW updates); rank 0 compares the parameters after the step with a single-process step on the hand-averaged
gradients.  Prints OK / FAIL."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def build(c, dev):
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    from conftest import sd_from_npz
    d = c["d"]
    ga, ta = TU.default_inference_args(f_maps=c["f_maps"], num_levels=c["levels"], left_hemis_only=True, num_groups=c["groups"])
    s = TU.InferenceSession(ga, ta, dev, state_dict=sd_from_npz(d), passes=3)
    step = TR.TrainStep(s.engine, s.model.head.tail(s.engine), c["loss_names"], c["loss_weights"], d["weights_ce"],
                        c["all_samples"], max_surf_distance=c["max_dist"], bias_field_log_type="l2" if c["bias_l2"] else "l1",
                        lr=c["lr"], weight_decay=c["wd"], betas=(c["b1"], c["b2"]), eps=c["eps"], clip_max_norm=0.0)
    target = {k[7:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("target/")}
    xs = [torch.from_numpy(d["x%d" % i]) for i in range(c["n_samples"])]
    samples = [{"bias_field_log": torch.from_numpy(d["bias_field_log%d" % i]),
                "high_res_residual": torch.from_numpy(d["high_res_residual%d" % i])} for i in range(c["n_samples"])]
    return step, xs, target, samples


def worker(rank, world, port, q):
    from test_oracle_train import load_case
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    c = load_case()
    step, xs, target, samples = build(c, dev)
    # rank r trains on sample r only
    ld, tot, ok = step.step([xs[rank]], target, [samples[rank]])
    after = {k: v.clone() for k, v in step.parameters().items()}
    good = bool(ok)
    if rank == 0:
        ref, xs2, target2, samples2 = build(c, dev)
        g = []
        for r in range(world):
            _, _, gr = ref.loss_and_grads([xs2[r]], target2, [samples2[r]])
            g.append(gr)
        mean = {k: sum(gg[k] for gg in g) / world for k in g[0]}
        ref.apply(mean)
        worst = 0.0
        for k, v in ref.parameters().items():
            worst = max(worst, float((v - after[k]).abs().max()))
        print("max |param difference| after one step: %.3e (lr %.1e)" % (worst, c["lr"]), flush=True)
        good = good and worst <= 2e-2 * c["lr"]
    # a non-finite loss on ONE rank (Trainer/engine.py:124-145 decides on the all-reduced loss): both ranks must skip the
    # iteration together -- the rank with the finite loss must not wait in the gradient all-reduce for the other one --
    # and leave the parameters untouched
    bad_x = xs[rank].clone()
    if rank == 1:
        bad_x[0, 0, 0, 0, 0] = float("nan")
    ld2, tot2, ok2 = step.step([bad_x], target, [samples[rank]])
    untouched = all(torch.equal(v, after[k]) for k, v in step.parameters().items())
    print("rank %d: non-finite iteration -> total %r stepped %r parameters untouched %r" % (rank, tot2, ok2, untouched), flush=True)
    good = good and (not ok2) and untouched and (tot2 != tot2 or abs(tot2) == float("inf"))
    # ... and the next finite iteration runs again on both
    ld3, tot3, ok3 = step.step([xs[rank]], target, [samples[rank]])
    good = good and bool(ok3)
    step_state_before4 = {k: (m.clone(), v.clone()) for k, (m, v) in step.state.items()}
    steps_before4, t_before4 = dict(step.steps), step.t
    # gradient accumulation over two samples per rank (each rank sees both, in opposite order): the first sample goes the
    # ordinary way, the LAST one writes into the GradStore, adds the first one's gradients slot by slot and starts the
    # buckets' all-reduces under its own backward pass; mean over the ranks of (g_a + g_b) == g_0 + g_1, i.e. what ONE process
    # computes for both samples
    before4 = {k: v.clone() for k, v in step.parameters().items()}
    order = [rank, 1 - rank]
    ld4, tot4, ok4 = step.step([xs[i] for i in order], target, [samples[i] for i in order])
    good = good and bool(ok4)
    if rank == 0:
        ref2, xs3, target3, samples3 = build(c, dev)
        for k, v in ref2.parameters().items():
            v.copy_(before4[k])
        ref2._weights_changed()
        ref2.state = {k: (m.clone(), v.clone()) for k, (m, v) in step_state_before4.items()} if step_state_before4 else {}
        ref2.steps, ref2.t = dict(steps_before4), t_before4
        _, _, gr = ref2.loss_and_grads([xs3[0], xs3[1]], target3, [samples3[0], samples3[1]])
        ref2.apply(gr)
        worst = max(float((v - step.parameters()[k]).abs().max()) for k, v in ref2.parameters().items())
        print("two samples per rank: max |param difference| %.3e (lr %.1e)" % (worst, c["lr"]), flush=True)
        good = good and worst <= 2e-2 * c["lr"]
    after = {k: v.clone() for k, v in step.parameters().items()}
    flags = [None] * world
    dist.all_gather_object(flags, bool(good))
    good = all(flags)
    # every rank must hold the same parameters
    flat = torch.cat([v.reshape(-1) for v in after.values()])
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(other[0], o) for o in other)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(good and same)


if __name__ == "__main__":
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(300)
    ok = (not q.empty()) and q.get(timeout=5)
    print("exit codes", [p.exitcode for p in ps])
    print("OK" if ok and all(p.exitcode == 0 for p in ps) else "FAIL")
