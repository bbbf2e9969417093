"""Golden vectors for interpol.grid_push / grid_count / grid_grad (order 1, 3-D, all 7 boundary conditions) and for
the first-order backward of grid_pull / grid_push (utils/interpol/autograd.py, pushpull.py), from the reference's
vendored torch-interpol on the CPU.  Run: python tests/golden/make_golden_pushgrad.py -> interpol_pushgrad.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

R = ref_import.setup()
import torch  # noqa: E402


def main():
    from utils import interpol
    g = torch.Generator().manual_seed(33)
    out = {}
    ins, outs = (7, 6, 8), (5, 9, 6)
    vol = torch.randn((2, 3) + ins, generator=g)
    # coordinates spilling over every face so that all boundary rules are exercised
    grid = torch.rand((2,) + outs + (3,), generator=g) * (torch.tensor(ins, dtype=torch.float32) + 5.0) - 2.5
    src = torch.randn((2, 3) + outs, generator=g)
    out["vol"], out["grid"], out["src"] = vol.numpy(), grid.numpy(), src.numpy()
    for bound in range(7):
        for ext in (0, 1):
            k = "b%d_e%d/" % (bound, ext)
            out[k + "push"] = interpol.grid_push(src, grid, list(ins), 1, bound, bool(ext)).numpy()
            out[k + "grad"] = interpol.grid_grad(vol, grid, 1, bound, bool(ext)).numpy()
            if bound in (0, 3):
                out[k + "count"] = interpol.grid_count(grid, list(ins), 1, bound, bool(ext)).numpy()
                v = vol.clone().requires_grad_(True)
                gr = grid.clone().requires_grad_(True)
                y = interpol.grid_pull(v, gr, 1, bound, bool(ext))
                w = torch.sin(torch.arange(y.numel(), dtype=torch.float32)).reshape(y.shape)
                (y * w).sum().backward()
                out[k + "pull_dinput"], out[k + "pull_dgrid"] = v.grad.numpy(), gr.grad.numpy()
                s_ = src.clone().requires_grad_(True)
                gr = grid.clone().requires_grad_(True)
                y = interpol.grid_push(s_, gr, list(ins), 1, bound, bool(ext))
                w2 = torch.cos(torch.arange(y.numel(), dtype=torch.float32)).reshape(y.shape)
                (y * w2).sum().backward()
                out[k + "push_dinput"], out[k + "push_dgrid"] = s_.grad.numpy(), gr.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "interpol_pushgrad.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
