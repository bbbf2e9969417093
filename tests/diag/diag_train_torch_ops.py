"""Which torch operators (at::native kernels) and host synchronisations a steady-state training iteration still makes, and
from where (the first frame inside brainfm_amd).    python tests/diag/diag_train_torch_ops.py [size=128]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["128"]) + ["1", "1"]
import runpy
import torch
from torch.utils._python_dispatch import TorchDispatchMode

ns = runpy.run_path(os.path.join(ROOT, "scripts", "bench_train.py"))      # warm: tuned, packed, optimiser state allocated
step, xs, target, samples, TR = ns["step"], ns["xs"], ns["target"], ns["samples"], ns["TR"]
forward_only = ns["forward_only"]
log = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "brainfm_amd" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not name.startswith(("empty", "view", "reshape", "as_strided", "detach", "alias", "slice", "select", "permute", "unsqueeze",
                                "squeeze", "expand", "t.", "transpose", "_unsafe_view", "unbind", "split", "narrow", "lift_fresh",
                                "is_", "size", "stride", "numel", "sym_", "_local_scalar_dense")):
            log[(name, where())] += 1
        elif name.startswith("_local_scalar_dense"):
            log[("HOST SYNC (.item() / float())", where())] += 1
        return func(*args, **(kwargs or {}))


torch.cuda.synchronize()
with Spy():
    forward_only()
    loss_dict, total, grads = step.loss_and_grads(xs, target, samples)
    TR.allreduce_mean_(grads)
    ok, _ = step.apply(grads)
torch.cuda.synchronize()
tot = collections.Counter()
for (name, w), c in sorted(log.items(), key=lambda kv: -kv[1]):
    print("%4d  %-34s %s" % (c, name, w))
    tot[name] += c
print(dict(tot))
