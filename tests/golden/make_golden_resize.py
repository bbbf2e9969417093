"""Golden vectors for interpol.resize(interpolation=3, bound='dct2', prefilter=True) -- the `bspline_zooming` call of
Generator/datasets.py:337-338 -- and for the prefilter alone, from the reference's vendored torch-interpol (CPU).
Run:  python tests/golden/make_golden_resize.py   -> tests/golden/interpol_resize.npz"""
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
    from utils.interpol.coeff import spline_coeff_nd
    g = torch.Generator().manual_seed(21)
    out = {}
    cases = {"up": ((20, 24, 18), (32, 32, 32), "edge"), "down": ((30, 26, 34), (16, 20, 12), "edge"),
             "centers": ((12, 10, 14), (25, 17, 14), "c")}
    for name, (ins, outs, anchor) in cases.items():
        x = torch.rand(ins, generator=g) * 5 - 1
        x[:3] = 0
        out[name + "/x"] = x.numpy()
        out[name + "/coeff"] = spline_coeff_nd(x.clone(), [3, 3, 3], [3, 3, 3], 3).numpy()
        y = interpol.resize(x, shape=list(outs), anchor=anchor, interpolation=3, bound="dct2", prefilter=True)
        out[name + "/y"] = y.numpy()
        out[name + "/shape"] = np.array(outs)
        print(name, ins, "->", tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(os.path.join(HERE, "interpol_resize.npz"), **out)


if __name__ == "__main__":
    main()
