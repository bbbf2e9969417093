"""Where a deep-level conv_mfma workgroup spends its time (diagnostics build: BFM_HIPCC_EXTRA=-DBFM_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from brainfm_amd import _lib as L, test_utils as TU
from brainfm_amd.engine import _Layer
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ver = int(sys.argv[4]) if len(sys.argv) > 4 else 0
splitk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
D = H = W = 5
dev = torch.device("cuda:0"); torch.manual_seed(0)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, dev).engine
lib = L.load()
A = torch.randn(S, D, H, W, cin, device=dev); scale = torch.rand(S, cin, device=dev) + 0.5
shift = torch.randn(S, cin, device=dev) * 0.1; bound = torch.full((S, 8), 6.0, device=dev)
out = torch.empty(S, D, H, W, cout, device=dev)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "bench", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
cfg = (C.c_int * 8)(); L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan"); cfg[6] = ver
if splitk: cfg[5] = splitk
print("plan", list(cfg))
eng._pack(ly, True, cfg[6])
st = L.stream_ptr()
need = lib.bfm_conv3x3x3_mfma_batch_workspace(cin, cout, S, D, H, W, cfg[5])
ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
def run():
    L.check(lib.bfm_conv3x3x3_mfma_batch(L.ptr(A), cin, None, 0, S, D, H, W, None, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8,
                                         L.ptr(ly.wpacked), ly.wexp, cout, 0.01, 3, cfg, L.ptr(out), L.ptr(ws), ws.numel(),
                                         None, 0, st), "conv")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
wbytes = 27.0 * cin * cout * 4
print("S=%d %d->%d ver %d: %.1f us  %.1f TFLOP/s alg  weights %.2f TB/s" % (S, cin, cout, ver, ms * 1e3, 2.0 * 27 * cin * cout * 125 * S / ms / 1e9, wbytes / ms / 1e9))
if hasattr(lib, "bfm_debug_stamps"):
    buf = (C.c_longlong * 4096)()
    lib.bfm_debug_stamps.argtypes = [C.c_void_p]; lib.bfm_debug_stamps(buf)
    for base in (0, 1024):
        n = buf[base + 1003]
        print("block", base, "chunks", n, "t_wait", buf[base + 1000], "t_bar", buf[base + 1001])
        t0 = buf[base]
        for k in range(min(n, 6)):
            s = [buf[base + k * 8 + i] - t0 for i in range(4)]
            nxt = (buf[base + (k + 1) * 8] if k + 1 < n else buf[base + 1002]) - t0
            print("  chunk %d: top %d  bar1 +%d  stage +%d  bar2 +%d  rows +%d" % (k, s[0], s[1] - s[0], s[2] - s[1], s[3] - s[2], nxt - s[3]))
