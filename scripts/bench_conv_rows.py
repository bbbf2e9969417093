"""Cost of emitting output-moment rows from the conv epilogue: one 64->64 layer at 160^3, each variant, with/without."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from brainfm_amd import _lib as L, test_utils as TU

dev = torch.device("cuda:0")
torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
sess = TU.InferenceSession(ga, ta, dev)
eng = sess.engine
lib = L.load()
D = H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 160
ly = eng.dec[-1][1]
A = torch.randn(D, H, W, 64, device=dev)
scale = torch.rand(64, device=dev) + 0.5
shift = torch.randn(64, device=dev) * 0.1
bound = torch.full((ly.groups,), 6.0, device=dev)
out = torch.empty(D, H, W, 64, device=dev)
st = L.stream_ptr()
for ver in (0, 2):
    cfg = (C.c_int * 8)()
    L.check(lib.bfm_conv3x3x3_mfma_plan(64, 64, D, H, W, cfg), "plan")
    cfg[6] = ver
    eng._pack(ly, True, ver)
    n = lib.bfm_conv3x3x3_mfma_rows(64, 64, D, H, W, cfg)
    rows = torch.empty(lib.bfm_moment_rows_bytes(n, 64), dtype=torch.uint8, device=dev)
    ws = torch.empty(1024, dtype=torch.uint8, device=dev)
    for use in (False, True, False, True):
        def run():
            L.check(lib.bfm_conv3x3x3_mfma_ex(L.ptr(A), 64, None, 0, D, H, W, None, L.ptr(scale), L.ptr(shift),
                                              L.ptr(bound), ly.groups, L.ptr(ly.wpacked), ly.wexp, 64, 0.01, 3, cfg,
                                              L.ptr(out), L.ptr(ws), ws.numel(), L.ptr(rows) if use else None, st), "conv")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); e1.synchronize()
        print("ver %d rows=%-5s %8.1f us" % (ver, use, e0.elapsed_time(e1) * 100))
