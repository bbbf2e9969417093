"""augment_pathology (Generator/utils.py:542-560) at 160^3: three Perlin potentials -> curl velocity -> dopri5 integration
of the advection PDE over nt output times, controller on the device (BFM_ODE_DEVICE=0: the host-controlled loop of
rounds 1-3).  usage: python scripts/bench_ode.py [size=160] [reps=5]"""
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from brainfm_amd import generator_utils as GU
from brainfm_amd import shapeid as SH

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
args = Namespace(perlin_res=[2, 2, 2], integ_method="dopri5", bc="neumann", V_multiplier=500, dt=0.1, max_nt=10,
                 pathol_thres=0.2, pathol_tol=1e-5, mask_percentile_min=85., mask_percentile_max=99.)
t = torch.from_numpy(np.arange(args.max_nt) * args.dt)
pde = SH.AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict={}, BC="neumann",
                    dt=args.dt, device=dev)
np.random.seed(0)
_, P0 = SH.generate_shape_3d((N, N, N), args.perlin_res, 90.0, dev)


class FixedNt:
    """augment_pathology draws nt = randint(1, max_nt + 1); time the longest case (nt = max_nt)."""
    def __enter__(self):
        self.orig = np.random.randint
        np.random.randint = lambda a, b=None: args.max_nt
    def __exit__(self, *a):
        np.random.randint = self.orig


for mode in ("1", "0"):
    os.environ["BFM_ODE_DEVICE"] = mode
    ts, steps = [], []
    for r in range(reps + 1):
        np.random.seed(10 + r)
        pde.nfe = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with FixedNt():
            out = GU.augment_pathology(P0, pde, t, args, dev)
        torch.cuda.synchronize()
        if r:
            ts.append(time.perf_counter() - t0)
            steps.append((pde.nfe - 2) // 6)
    print("augment_pathology %d^3, nt = %d, controller on the %s: median %.2f ms (min %.2f), %s steps"
          % (N, args.max_nt, "device" if mode == "1" else "host", 1e3 * float(np.median(ts)), 1e3 * min(ts), steps))
