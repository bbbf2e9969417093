"""How fast the uniform-box streaming kernels write (conv_wino4d_uniform / conv_wino_uniform[_pool]): a tensor whose image is
constant almost everywhere, so that the pair's time is the streaming kernel's.    python tests/diag/diag_uniform_stream.py [size=160]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
cin = cout = 64
A = torch.randn(S, S, S, cin, device=dev)
w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
scale = torch.rand(cin, device=dev) + 0.5
shift = torch.randn(cin, device=dev) * 0.1
bound = torch.full((8,), 6.0, device=dev)
img = torch.zeros(S, S, S, device=dev)
img[:8, :8, :8] = 1.0
st = L.stream_ptr()


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = torch.empty(S, S, S, cout, device=dev)
nb = out.numel() * 4
for name, kind in (("F(4,3) pair", 4), ("F(2,3) pair", 3), ("F(2,3) pair + pooling", 5)):
    f4 = kind == 4
    wp = torch.empty((lib.bfm_pack_conv_weights_wino4_bytes if f4 else lib.bfm_pack_conv_weights_wino_bytes)(cin, cout, 3), dtype=torch.uint8, device=dev)
    wexp = C.c_int(0)
    L.check((lib.bfm_pack_conv_weights_wino4 if f4 else lib.bfm_pack_conv_weights_wino)(L.ptr(w), cin, cout, float(w.abs().max()), 3, L.ptr(wp), C.byref(wexp), st), "pack")
    flags = torch.empty(lib.bfm_uniform_boxes_bytes(S, S, S, 3), dtype=torch.uint8, device=dev)
    L.check(lib.bfm_uniform_boxes_level(L.ptr(img), S, S, S, 0, 2, 3, L.ptr(flags), st), "flags")
    nrows = lib.bfm_conv3x3x3_wino_rows(S, S, S, 3)
    frac = float((flags[:nrows] != 0).float().mean())
    scratch = torch.empty((lib.bfm_conv3x3x3_wino4_uniform_scratch if f4 else lib.bfm_conv3x3x3_wino_uniform_scratch)(cout), dtype=torch.uint8, device=dev)
    rows = torch.empty(lib.bfm_moment_rows_bytes(nrows, cout), dtype=torch.uint8, device=dev)
    pooled = torch.empty(S // 2, S // 2, S // 2, cout, device=dev)
    prow = torch.empty(lib.bfm_moment_rows_bytes(nrows, cout), dtype=torch.uint8, device=dev)
    for accum in (0, 1):
        for with_rows in (False, True):
            common = (L.ptr(A), cin, S, S, S, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8, L.ptr(wp), wexp.value, cout, 0.01, 3, accum,
                      L.ptr(out), L.ptr(rows) if with_rows else None, L.ptr(flags), L.ptr(scratch))
            if kind == 4:
                fn = lambda: L.check(lib.bfm_conv3x3x3_wino4_uniform(*common, st), "u4")
            elif kind == 3:
                fn = lambda: L.check(lib.bfm_conv3x3x3_wino_uniform(*common, st), "u3")
            else:
                fn = lambda: L.check(lib.bfm_conv3x3x3_wino_uniform_pool(*common, L.ptr(pooled), L.ptr(prow) if with_rows else None, st), "u5")
            ms = timed(fn)
            moved = nb * frac * (2 if accum else 1)
            print("%-22s %d^3 x %d, %.1f %% of the boxes flagged, accumulate %d, moment rows %s: %.3f ms = %.2f TB/s of output%s"
                  % (name, S, cout, 100 * frac, accum, with_rows, ms, moved / ms * 1e-9, " read + written" if accum else " written"), flush=True)
cp = torch.empty_like(out)
print("device-to-device copy of the same tensor: %.3f ms = %.2f TB/s read + written" % (timed(lambda: cp.copy_(out)), 2 * nb / timed(lambda: cp.copy_(out)) * 1e-9))
print("fill of the same tensor: %.3f ms" % timed(lambda: cp.fill_(1.0)))
