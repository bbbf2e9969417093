"""Split-K factor sweep on the deep batched layers (bfm_conv3x3x3_mfma_batch, plan from bfm_conv3x3x3_mfma_plan with cfg[5]
overridden): time per launch incl. the slab reduction.    python tests/diag/diag_splitk_sweep.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from brainfm_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
CASES = [(3072, 1024, (5, 5, 5)), (1024, 1024, (5, 5, 5)), (2048, 2048, (2, 2, 2)), (1024, 2048, (2, 2, 2)), (1536, 512, (10, 10, 10)),
         (1024, 512, (10, 10, 10)), (512, 512, (10, 10, 10)), (512, 1024, (5, 5, 5)), (256, 512, (10, 10, 10)), (768, 256, (20, 20, 20)),
         (256, 256, (20, 20, 20))]
for S in (8, 2):
    for cin, cout, dims in CASES:
        D, H, W = dims
        A = torch.randn(S, D, H, W, cin, device=dev)
        w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02).contiguous()
        scale = (torch.rand(S, cin, device=dev) + 0.5)
        shift = torch.randn(S, cin, device=dev) * 0.1
        bound = torch.full((S, 8), 6.0, device=dev)
        cfg = (C.c_int * 8)()
        L.check(lib.bfm_conv3x3x3_mfma_plan(cin, cout, D, H, W, cfg), "plan")
        cfg[6] = 0
        wp = torch.empty(lib.bfm_pack_conv_weights_mfma_bytes(cin, cout), dtype=torch.uint8, device=dev)
        wexp = C.c_int(0)
        L.check(lib.bfm_pack_conv_weights_mfma(L.ptr(w), cin, cout, float(w.abs().max()), L.ptr(wp), C.byref(wexp), L.stream_ptr()), "pack")
        out = torch.empty(S, D, H, W, cout, device=dev)
        planned = cfg[5]
        res = []
        for sk in sorted({planned, max(1, planned // 2), max(1, planned // 4), max(1, planned * 3 // 4), min(cin // 16, planned * 2)}):
            cfg[5] = sk
            nws = lib.bfm_conv3x3x3_mfma_batch_workspace(cin, cout, S, D, H, W, sk)
            ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=dev)
            nr = lib.bfm_conv3x3x3_mfma_rows(cin, cout, D, H, W, cfg)
            rows = torch.empty(max(lib.bfm_moment_rows_bytes(S * max(nr, 1), cout), 256), dtype=torch.uint8, device=dev)

            def go():
                return lib.bfm_conv3x3x3_mfma_batch(L.ptr(A), cin, None, 0, S, D, H, W, None, L.ptr(scale), L.ptr(shift), L.ptr(bound), 8,
                                                    L.ptr(wp), wexp.value, cout, 0.01, 3, cfg, L.ptr(out), L.ptr(ws), ws.numel(),
                                                    L.ptr(rows) if nr > 0 else None, 0, L.stream_ptr())
            if go() != 0:
                continue
            go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.append((sk, e0.elapsed_time(e1) * 100))
        cfg[5] = planned
        best = min(res, key=lambda t: t[1])
        print("x%d %4d -> %4d %-9s planned split %2d: %s   best %d (%.0f us, %+.1f %% vs planned)" % (
            S, cin, cout, "x".join(map(str, dims)), planned, "  ".join("%d: %.0f us" % t for t in res), best[0], best[1],
            100 * (best[1] / dict(res)[planned] - 1)), flush=True)
