"""What a conv_mfma launch of the deep (split-K, batched) layers spends outside its K loop (-DBFM_MFMA_ABLATE build:
BFM_MFMA_ABL 32 = no K loop, 16 = no epilogue; ablated launches compute wrong results).
    python tests/diag/diag_mfma_overheads.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
CASES = [(3072, 1024, (5, 5, 5), 8), (1024, 1024, (5, 5, 5), 8), (1024, 512, (10, 10, 10), 8), (512, 512, (10, 10, 10), 8),
         (256, 256, (20, 20, 20), 8), (256, 256, (40, 40, 40), 1), (2048, 2048, (2, 2, 2), 8)]
for cin, cout, dims, S in CASES:
    D, H, W = dims
    A = torch.randn(S, D, H, W, cin, device=dev)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02).contiguous()
    scale = (torch.rand(S, cin, device=dev) + 0.5)
    shift = torch.randn(S, cin, device=dev) * 0.1
    bound = torch.full((S, 8), 6.0, device=dev)
    cfg = (C.c_int * 8)()
    L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan")
    for ver in (0, 2):
        cfg[6] = ver
        fn_b = lib.bfm_pack_conv_weights_mfma16_bytes if ver == 2 else lib.bfm_pack_conv_weights_mfma_bytes
        fn_p = lib.bfm_pack_conv_weights_mfma16 if ver == 2 else lib.bfm_pack_conv_weights_mfma
        wp = torch.empty(fn_b(cin, cout), dtype=torch.uint8, device=dev)
        wexp = C.c_int(0)
        L.check(fn_p(L.ptr(w), cin, cout, float(w.abs().max()), L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
        out = torch.empty(S, D, H, W, cout, device=dev)
        nws = lib.bfm_conv3x3x3_mfma_batch_workspace(cin, cout, S, D, H, W, cfg[5])
        ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=dev)

        def go():
            rc = lib.bfm_conv3x3x3_mfma_batch(L.ptr(A), cin, None, 0, S, D, H, W, None, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8,
                                              L.ptr(wp), wexp.value, cout, 0.01, 3, cfg, L.ptr(out), L.ptr(ws), ws.numel(), None, 0,
                                              L.stream_ptr())
            return rc

        if go() != 0:
            print("%d -> %d %s x%d ver %d: plan not launchable" % (cin, cout, dims, S, ver))
            continue

        def timed(reps=10):
            go(); go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                go()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        res = {}
        for abl in (0, 16, 32, 48):
            os.environ["BFM_MFMA_ABL"] = str(abl)
            res[abl] = timed()
        os.environ.pop("BFM_MFMA_ABL", None)
        print("%4d -> %4d %-12s x%d ver %d plan %s: full %7.1f us | no epilogue %7.1f | no K loop %7.1f | neither %7.1f   (incl. the split-K reduce)"
              % (cin, cout, "x".join(map(str, dims)), S, ver, list(cfg)[:6], res[0], res[16], res[32], res[48]), flush=True)
