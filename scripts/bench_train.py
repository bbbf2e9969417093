"""One training iteration of the full-width net (f_maps 64, 6 levels, 69 head channels) on a 128^3 crop --
the reference's training crop (cfgs/generator/default.yaml:63) -- timed per phase with HIP events on torch's stream.
usage: python scripts/bench_train.py [size=128] [samples=1] [reps=3]
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_train.py ...
Under torch.distributed every rank trains on its own samples (weak scaling, BASELINE config 4: batch = N) and the
gradients are averaged with one flat RCCL all-reduce before the optimiser (DDP semantics); rank 0 reports.
Run under rocprofv3 --kernel-trace --stats for the per-kernel table (profiles/r01_train_kernel_trace.txt)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brainfm_amd import backward as BW
from brainfm_amd import test_utils as TU
from brainfm_amd import train as TR

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device("cuda:%d" % local)
torch.cuda.set_device(dev)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)                       # default nn init under seed 1, as bench.py
s = TU.InferenceSession(ga, ta, dev, passes=3)
tail = s.model.head.tail(s.engine)
names = ["T1", "T1_grad", "T2", "T2_grad", "FLAIR", "FLAIR_grad", "CT", "CT_grad", "seg_ce", "seg_dice", "distance",
         "bias_field_log", "registration", "registration_grad", "SR", "SR_grad"]
ns = tail.desc.n_seg
step = TR.TrainStep(s.engine, tail, names, {"loss_" + n: 1.0 for n in names}, torch.full((ns,), 1.0 / ns), 4, lr=1e-4)
g = torch.Generator().manual_seed(rank)
dims = (N, N, N)
xs = [torch.rand((1, 1) + dims, generator=g).to(dev) for _ in range(n_samples)]
lab = torch.randint(0, ns, (1,) + dims, generator=g)
target = {"segmentation": torch.nn.functional.one_hot(lab, ns).permute(0, 4, 1, 2, 3).float().contiguous().to(dev)}
for k in ("T1", "T2", "FLAIR", "CT"):
    target[k] = torch.rand((1, 1) + dims, generator=g).to(dev)
target["distance"] = torch.randn((1, 4) + dims, generator=g).to(dev)
target["registration"] = torch.randn((1, 3) + dims, generator=g).to(dev)
samples = [{"bias_field_log": torch.randn((1, 1) + dims, generator=g).to(dev) * 0.3,
            "high_res_residual": torch.randn((1, 1) + dims, generator=g).to(dev) * 0.2} for _ in range(n_samples)]


def ev():
    return torch.cuda.Event(enable_timing=True)


def forward_only():
    for x in xs:
        feats, tape = BW.backbone_forward_train(s.engine, s.engine.to_cl(x), dims)
        tail.run_raw(feats[-1][0], dims)


forward_only()                                   # tunes the conv variants, packs the weights
torch.cuda.synchronize()
tf = tb = to = 0.0
for r in range(reps + 1):
    e = [ev() for _ in range(4)]
    e[0].record()
    forward_only()
    e[1].record()
    loss_dict, total, grads = step.loss_and_grads(xs, target, samples)
    e[2].record()
    TR.allreduce_mean_(grads)                 # no-op without a process group
    ok, _ = step.apply(grads)
    e[3].record()
    torch.cuda.synchronize()
    if r == 0:
        continue                                  # first pass allocates the optimiser state and the dgrad packs
    tf += e[0].elapsed_time(e[1])
    tb += e[1].elapsed_time(e[2])
    to += e[2].elapsed_time(e[3])
tf, tb, to = tf / reps, tb / reps, to / reps
if world > 1:
    t = torch.tensor([tf, tb, to], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tf, tb, to = t.tolist()
nv = N ** 3 * n_samples * world
if rank == 0:
    print("ranks %d; crop %d^3 x %d samples per rank: forward %.1f ms | forward+losses+backward %.1f ms (backward alone "
          "~%.1f) | all-reduce + clip + AdamW %.1f ms" % (world, N, n_samples, tf, tb, tb - tf, to))
    print("iteration %.1f ms = %.2f Mvox/s; loss %.4f stepped=%s; peak memory %.1f GB"
          % (tb + to, nv / (tb + to) / 1e3, total, ok, torch.cuda.max_memory_allocated() / 2 ** 30))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
