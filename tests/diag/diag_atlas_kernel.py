"""Is the stand-alone deformed-atlas gather flaky on its own / next to other work? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from brainfm_amd import _lib as L
dev = torch.device("cuda:0")
lib = L.load()
n = 160 * 160 * 160
g = torch.Generator().manual_seed(0)
rx, ry, rz = [(torch.randn(n, generator=g) * 0.05).to(dev) for _ in range(3)]
mask = torch.ones(n, device=dev)
vol = torch.full((256, 256, 256), 100.0, device=dev)
A = (C.c_float * 12)(-1, 0, 0, 128, 0, 0, -1, 128, 0, 1, 0, 128)
out = torch.empty(n, device=dev)
def run(stream=None):
    L.check(lib.bfm_deformed_atlas_tile(L.ptr(mask), L.ptr(rx), L.ptr(ry), L.ptr(rz), L.ptr(vol), 256, 256, 256, A, n,
                                        L.ptr(out), L.stream_ptr()), "atlas")
bad = 0
for it in range(200):
    out.fill_(-1.0)
    run()
    torch.cuda.synchronize()
    b = int((out != 100.0).sum())
    bad += b
    if b:
        idx = torch.nonzero(out != 100.0).reshape(-1)
        print("alone it", it, "bad", b, idx[:8].tolist(), out[idx[:4]].tolist(), flush=True)
print("alone: total bad", bad, flush=True)
# next to a busy second stream (large matmuls) and next to the conv kernels of this library
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
bad = 0
for it in range(100):
    out.fill_(-1.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            c = a @ a
    run()
    torch.cuda.synchronize()
    b = int((out != 100.0).sum())
    bad += b
    if b and it < 10:
        idx = torch.nonzero(out != 100.0).reshape(-1)
        print("with matmul it", it, "bad", b, idx[:8].tolist(), out[idx[:4]].tolist(), flush=True)
print("next to matmuls: total bad", bad, flush=True)
from brainfm_amd import test_utils as TU
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
torch.manual_seed(1)
s = TU.InferenceSession(ga, ta, dev, passes=3)
x = torch.rand(1, 1, 80, 80, 80, device=dev)
TU._run_tile(s, x, raw=True)
torch.cuda.synchronize()
bad = 0
for it in range(60):
    out.fill_(-1.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        TU._run_tile(s, x, raw=True)
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    b = int((out != 100.0).sum())
    bad += b
    if b and it < 10:
        idx = torch.nonzero(out != 100.0).reshape(-1)
        print("with tile it", it, "bad", b, idx[:8].tolist(), out[idx[:4]].tolist(), flush=True)
print("next to a tile's kernels: total bad", bad, flush=True)
