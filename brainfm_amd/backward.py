"""Backward pass of the U-Net backbone on the HIP kernels (first correct version).

The reference trains through torch autograd over ``Trainer/models/unet3d/buildingblocks.py`` (SingleConv 'gcl' :31-60,
MaxPool3d :185-186, nearest upsample + concat :265-276,361-363) and ``unet3d/model.py:195-209`` (get_feature with the
L2-normalised last map).  Here the forward pass of ``UNetEngine`` is re-run in *training mode* -- same kernels, but every
SingleConv keeps a tape (its inputs, the GroupNorm affine, mean / rstd, its output) -- and ``backbone_backward`` walks
the tapes in reverse:

    dP  = dY * LeakyReLU'(Y)                              bfm_lrelu_bwd
    dW  = correlate(GN(x), dP)                            bfm_conv3x3x3_wgrad   (exact-fp32 matrix cores)
    dXn = conv3x3x3(dP, W^T, taps mirrored)               the forward conv kernels on transposed weights
    dx, dgamma, dbeta = GroupNorm backward                bfm_gn_bwd  (the low-res half sums over its replica boxes)
    MaxPool3d backward                                    bfm_maxpool2_bwd

What this does not cover yet: the task heads and losses (criterion.py), AMP loss scaling, the optimiser, DDP.
Gradients are returned under the reference's parameter names.
"""
import ctypes as C
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from .engine import _Layer


TRAIN_FAST = os.environ.get("BFM_TRAIN_FAST", "1") != "0"
WGRAD_PASSES = int(os.environ.get("BFM_WGRAD_PASSES", "3"))     # 3: split-fp16 matrix core; 0: exact fp32 matrix core


class ConvTape:
    __slots__ = ("ly", "A", "B", "dims", "lo_dims", "scale", "shift", "mean", "rstd", "out", "bound")


def _start_tables(eng, lo, hi):
    """Per low-res index: first full-res index mapped to it (exclusive prefix sum of the replication counts)."""
    key = ("start", tuple(lo), tuple(hi))
    if key not in eng._up_cache:
        from .engine import nearest_index_map
        tabs = []
        for a in range(3):
            m = nearest_index_map(lo[a], hi[a])
            rep = np.bincount(m, minlength=lo[a])
            start = np.concatenate([[0], np.cumsum(rep)[:-1]]).astype(np.int32)
            tabs.append(torch.from_numpy(start).to(eng.device))
        eng._up_cache[key] = tabs
    return eng._up_cache[key]


def train_single_conv(eng, ly, A, dims, B=None, lo_dims=None):
    """Forward of one SingleConv in training mode: same kernels as inference (generic two-source implicit GEMM),
    GroupNorm statistics from the tensors, everything the backward needs kept on the tape."""
    D, H, W = dims
    ca = A.shape[-1]
    cb = 0 if B is None else B.shape[-1]
    st = L.stream_ptr()
    up = eng._upsample_desc(lo_dims, dims) if B is not None else None
    upp = C.byref(up) if up is not None else None
    dev = eng.device
    scale = torch.empty(ly.cin, dtype=torch.float32, device=dev)
    shift = torch.empty(ly.cin, dtype=torch.float32, device=dev)
    bound = torch.empty(ly.groups, dtype=torch.float32, device=dev)
    mean = torch.empty(ly.groups, dtype=torch.float32, device=dev)
    rstd = torch.empty(ly.groups, dtype=torch.float32, device=dev)
    mfma = eng._mfma_ok(ly, ca, cb)
    cfg, wsc = None, 0
    if mfma:
        cfg = eng._plan(ly.cin, ly.cout, dims, B is not None)
        wsc = eng.lib.bfm_conv3x3x3_mfma_workspace(ly.cin, ly.cout, D, H, W, cfg[5])
    ws = eng._workspace(max(eng.lib.bfm_gn_stats_workspace(ca, cb, D, H, W, upp), wsc))
    L.check(eng.lib.bfm_gn_stats_train(L.ptr(A), ca, L.ptr(B), cb, D, H, W, upp, L.ptr(ly.gamma), L.ptr(ly.beta),
                                       ly.groups, eng.eps, L.ptr(scale), L.ptr(shift), L.ptr(bound), L.ptr(mean),
                                       L.ptr(rstd), L.ptr(ws), ws.numel(), st), "gn_stats_train " + ly.name)
    out = torch.empty((D, H, W, ly.cout), dtype=torch.float32, device=dev)
    if mfma:
        eng._conv_launch(ly, A, ca, B, cb, dims, upp, scale, shift, bound, ly.groups, cfg, out, ws)
    else:
        eng._pack(ly, False)
        L.check(eng.lib.bfm_conv3x3x3_direct(L.ptr(A), ca, L.ptr(B), cb, D, H, W, upp, L.ptr(scale), L.ptr(shift),
                                             L.ptr(ly.wpacked), ly.cout, eng.slope, L.ptr(out), st),
                "conv_direct " + ly.name)
    t = ConvTape()
    t.ly, t.A, t.B, t.dims, t.lo_dims = ly, A, B, tuple(dims), (tuple(lo_dims) if lo_dims is not None else None)
    t.scale, t.shift, t.mean, t.rstd, t.out, t.bound = scale, shift, mean, rstd, out, bound
    return out, t


def _dgrad_cout(ly):
    """Output channels of the data-gradient conv: Cin of the layer, padded to 64 where that opens the matrix-core path
    (level 0: 64 -> 32 at 128^3 took 170 ms on the direct kernel, 2 ms padded to 64 -> 64; the one-channel stem 17 -> 1 ms)."""
    if ly.cin % 64 and ly.cout % 16 == 0:
        return (ly.cin + 63) // 64 * 64
    return ly.cin


def refresh_dgrad(ly, dg):
    """(Re)derive the transposed, tap-mirrored weights of `ly` into its data-gradient layer `dg` (one kernel pass)."""
    L.check(L.load().bfm_transpose_mirror_weights(L.ptr(ly.w_raw), ly.cout, ly.cin, dg.cout, L.ptr(dg.w_raw),
                                                  L.stream_ptr()), "transpose_mirror_weights " + ly.name)


def _dgrad_layer(eng, ly):
    """The transposed, tap-mirrored weights of `ly` as a layer of their own: conv(dP, W') = d(loss)/d(conv input)."""
    dg = _Layer()
    dg.gamma, dg.beta = None, None
    cout = _dgrad_cout(ly)
    dg.name, dg.cin, dg.cout, dg.groups = ly.name + "[dgrad]", ly.cout, cout, 1
    dg.w_raw = torch.empty((cout, ly.cout, 3, 3, 3), dtype=torch.float32, device=eng.device)
    dg.kind, dg.wpacked, dg.wexp, dg.packs, dg.skip = None, None, 0, {}, None
    refresh_dgrad(ly, dg)
    eng.pack_count += 1
    return dg


def _identity_affine(eng, c):
    """(ones, zeros) of c floats on the engine's device: the identity GroupNorm affine of the data-gradient convolutions."""
    cache = eng.__dict__.setdefault("_identity_affine", {})
    if c not in cache:
        cache[c] = (torch.ones(c, dtype=torch.float32, device=eng.device), torch.zeros(c, dtype=torch.float32, device=eng.device))
    return cache[c]


def _add(a, b):
    """a + b of two same-shape fp32 gradients through bfm_ew_binary (no torch operator in the step)."""
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape:
        return a + b
    from .generator_utils import ew_binary
    return ew_binary(L.EW_ADD, a, b)


def backward_single_conv(eng, t, dY, need_input_grad=True):
    """Returns (dA, dB, grads) for one taped SingleConv; dB is the gradient of the LOW-RES tensor (or None)."""
    ly = t.ly
    D, H, W = t.dims
    nv = D * H * W
    ca = t.A.shape[-1]
    cb = 0 if t.B is None else t.B.shape[-1]
    dev = eng.device
    st = L.stream_ptr()
    lib = eng.lib
    dY = dY.contiguous()
    dP = torch.empty_like(dY)
    bnd = torch.empty(1, dtype=torch.float32, device=dev)         # max |dP|, written by the same kernel
    L.check(lib.bfm_lrelu_bwd_ex(L.ptr(dY), L.ptr(t.out), dY.numel(), eng.slope, L.ptr(dP), L.ptr(bnd), st), "lrelu_bwd")
    up = eng._upsample_desc(t.lo_dims, t.dims) if t.B is not None else None
    upp = C.byref(up) if up is not None else None
    # ---- weight gradient (split-fp16 matrix-core kernel on the wide layers, exact fp32 one elsewhere / on request)
    wsb = lib.bfm_conv3x3x3_wgrad_workspace(ly.cin, ly.cout, D, H, W)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    sink = getattr(eng, "grad_sink", None)                       # train.GradStore: gradients land in their flat slot
    dW = (sink.out(ly.name + ".conv.weight", (ly.cout, ly.cin, 3, 3, 3)) if sink is not None else
          torch.empty((ly.cout, ly.cin, 3, 3, 3), dtype=torch.float32, device=dev))
    L.check(lib.bfm_conv3x3x3_wgrad_ex(L.ptr(dP), ly.cout, L.ptr(t.A), ca, L.ptr(t.B), cb, D, H, W, upp, L.ptr(t.scale),
                                       L.ptr(t.shift), L.ptr(bnd), L.ptr(t.bound), ly.groups, WGRAD_PASSES, L.ptr(dW),
                                       L.ptr(ws), ws.numel(), st), "conv_wgrad " + ly.name)
    if sink is not None:
        sink.done(ly.name + ".conv.weight")
    grads = OrderedDict()
    grads[ly.name + ".conv.weight"] = dW
    if not need_input_grad and ly.cin < 8:
        # the stem: dgamma / dbeta of its one-channel GroupNorm still need dXn; it is cheap (Cout' = 1)
        pass
    # ---- data gradient w.r.t. the normalised input: forward conv kernels on transposed weights, identity affine
    dg = ly.packs.get("dgrad_layer")                             # transposed weights live with their layer
    if dg is None:
        dg = ly.packs["dgrad_layer"] = _dgrad_layer(eng, ly)
    ly.touch("dgrad_layer")
    ones, zeros = _identity_affine(eng, ly.cout)                 # cached per width: no fill launches per layer and iteration
    dXn = torch.empty((D, H, W, dg.cout), dtype=torch.float32, device=dev)
    if ly.cout % 16 == 0 and dg.cout % 64 == 0:
        # the data gradient is a plain single-source conv of dP with the transposed weights: it goes through the
        # engine's own launcher, so it is autotuned over the same variants as a forward layer (Winograd included)
        key = (ly.cout, dg.cout, tuple(t.dims), False, False)
        cfg = eng._plan(ly.cout, dg.cout, t.dims, False)
        wsc = lib.bfm_conv3x3x3_mfma_workspace(ly.cout, dg.cout, D, H, W, cfg[5])
        ws2 = torch.empty(max(wsc, 256), dtype=torch.uint8, device=dev)

        def _launch(c):
            eng._conv_launch(dg, dP, ly.cout, None, 0, t.dims, None, ones, zeros, bnd, 1, c, dXn, ws2, None, slope=1.0)
        # F(4,3) where the table says so (round 5; BFM_TRAIN_F43=0: _autotune hands back a narrowed COPY, variant 3)
        from brainfm_amd.engine import TRAIN_F43
        cfg = eng._autotune(dg, key, _launch, vers=(0, 1, 2) if key[3] else ((0, 1, 2, 3, 4) if TRAIN_F43 else (0, 1, 2, 3)))
        _launch(cfg)
        if dg.cout != ly.cin:
            dXn = dXn[..., :ly.cin].contiguous()
    else:
        eng._pack(dg, False)
        L.check(lib.bfm_conv3x3x3_direct(L.ptr(dP), ly.cout, None, 0, D, H, W, None, L.ptr(ones), L.ptr(zeros),
                                         L.ptr(dg.wpacked), ly.cin, 1.0, L.ptr(dXn), st), "conv dgrad(direct) " + ly.name)
    # ---- GroupNorm backward
    dA = torch.empty((D, H, W, ca), dtype=torch.float32, device=dev)
    dB = torch.empty(tuple(t.lo_dims) + (cb,), dtype=torch.float32, device=dev) if cb else None
    dgamma = sink.out(ly.name + ".groupnorm.weight", (ly.cin,)) if sink is not None else torch.empty(ly.cin, dtype=torch.float32, device=dev)
    dbeta = sink.out(ly.name + ".groupnorm.bias", (ly.cin,)) if sink is not None else torch.empty(ly.cin, dtype=torch.float32, device=dev)
    starts = _start_tables(eng, t.lo_dims, t.dims) if cb else [None, None, None]
    wsg = torch.empty(lib.bfm_gn_bwd_workspace(ly.cin, D, H, W), dtype=torch.uint8, device=dev)
    L.check(lib.bfm_gn_bwd(L.ptr(dXn), L.ptr(t.A), ca, L.ptr(t.B), cb, D, H, W, upp, L.ptr(starts[0]), L.ptr(starts[1]),
                           L.ptr(starts[2]), L.ptr(t.mean), L.ptr(t.rstd), L.ptr(ly.gamma), ly.groups, L.ptr(dA),
                           L.ptr(dB), L.ptr(dgamma), L.ptr(dbeta), L.ptr(wsg), wsg.numel(), st), "gn_bwd " + ly.name)
    if sink is not None:
        sink.done(ly.name + ".groupnorm.weight")
        sink.done(ly.name + ".groupnorm.bias")
    grads[ly.name + ".groupnorm.weight"] = dgamma
    grads[ly.name + ".groupnorm.bias"] = dbeta
    return dA, dB, grads


def backbone_forward_train(eng, x_cl, dims, fast=None):
    """UNetEngine.backbone_cl in training mode.  Returns (feats, tape): feats as backbone_cl (deepest first, the last
    one not normalised), tape = what backbone_backward needs.
    fast: run the inference layer path itself (autotuned variants, Winograd, up-folded decoder convs, GroupNorm moments
    from producer rows) with the engine's tape hook collecting inputs / scale / shift / mean / rstd / outputs; otherwise
    the generic two-source kernels of train_single_conv (BFM_TRAIN_FAST=0 makes that the default)."""
    if fast is None:
        fast = TRAIN_FAST
    if fast:
        eng.tape = []
        try:
            feats = eng.backbone_cl(x_cl, dims)
            rec = eng.tape
        finally:
            eng.tape = None
        tape = {"enc": [], "dec": [], "pool": []}
        convs = []
        for r in rec:
            if "pool_in" in r:
                tape["pool"].append((r["pool_in"], r["pool_dims"]))
                continue
            t = ConvTape()
            t.ly, t.A, t.B, t.dims, t.lo_dims = r["ly"], r["A"], r["B"], r["dims"], r["lo_dims"]
            t.scale, t.shift, t.mean, t.rstd, t.out, t.bound = r["scale"], r["shift"], r["mean"], r["rstd"], r["out"], r["bound"]
            convs.append(t)
        ne = len(eng.enc)
        if len(convs) != 2 * (ne + len(eng.dec)):
            raise L.BfmError("training tape has %d conv records, expected %d" % (len(convs), 2 * (ne + len(eng.dec))))
        for i in range(ne):
            tape["enc"].append((convs[2 * i], convs[2 * i + 1]))
        for j in range(len(eng.dec)):
            tape["dec"].append((convs[2 * ne + 2 * j], convs[2 * ne + 2 * j + 1]))
        return feats, tape
    tape = {"enc": [], "dec": [], "pool": []}
    skips = []
    x, d = x_cl, tuple(dims)
    for i, (l1, l2) in enumerate(eng.enc):
        if i > 0:
            xin, din = x, d
            x, d = eng.maxpool(x, d)
            tape["pool"].append((xin, din))
        x, t1 = train_single_conv(eng, l1, x, d)
        x, t2 = train_single_conv(eng, l2, x, d)
        tape["enc"].append((t1, t2))
        skips.insert(0, (x, d))
    skips = skips[1:]
    feats = [(x, d)]
    for (l1, l2), (skip, sd_) in zip(eng.dec, skips):
        y, t1 = train_single_conv(eng, l1, skip, sd_, B=x, lo_dims=d)
        x, t2 = train_single_conv(eng, l2, y, sd_)
        d = sd_
        tape["dec"].append((t1, t2))
        feats.append((x, d))
    return feats, tape


def backbone_backward(eng, tape, dfeats):
    """dfeats: gradients w.r.t. the decoder feature maps returned by backbone_forward_train (same order, channels-last;
    None = zero).  Returns {parameter name: gradient}."""
    grads = OrderedDict()
    nlev = len(tape["enc"])
    dskip = [None] * nlev                       # gradient flowing into encoder level i's output from its decoder use
    # decoders, last to first: dec[j] consumes skip level (nlev-2-j) and the previous x
    g = dfeats[-1]
    for j in range(len(tape["dec"]) - 1, -1, -1):
        t1, t2 = tape["dec"][j]
        if g is None:
            g = torch.zeros_like(t2.out)
        dy, _, gr = backward_single_conv(eng, t2, g)
        grads.update(gr)
        dsk, dlow, gr = backward_single_conv(eng, t1, dy)
        grads.update(gr)
        dskip[nlev - 2 - j] = dsk
        g = dlow
        if dfeats[j] is not None:
            g = _add(g, dfeats[j])
    # g is now the gradient w.r.t. the deepest encoder output
    for i in range(nlev - 1, -1, -1):
        t1, t2 = tape["enc"][i]
        if dskip[i] is not None:
            g = _add(g, dskip[i]) if g is not None else dskip[i]
        dy, _, gr = backward_single_conv(eng, t2, g)
        grads.update(gr)
        dx, _, gr = backward_single_conv(eng, t1, dy, need_input_grad=i > 0)
        grads.update(gr)
        if i > 0:
            xin, din = tape["pool"][i - 1]
            dIn = torch.empty_like(xin)
            L.check(eng.lib.bfm_maxpool2_bwd(L.ptr(xin), L.ptr(dx.contiguous()), xin.shape[-1], din[0], din[1], din[2],
                                             L.ptr(dIn), L.stream_ptr()), "maxpool2_bwd")
            g = dIn
    return grads
