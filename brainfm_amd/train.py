"""One training iteration of the multi-task U-Net on the HIP kernels.

Mirrors the reference's iteration -- Trainer/engine.py:96-147:

    outputs, _ = model(samples)                       -> backward.backbone_forward_train + Tail.run_raw   (per sample)
    outputs = processor(outputs, ...)                 -> softmax / clamp folded into the loss kernels
    loss_dict = criterion(outputs, target, samples)   -> bfm_loss_l1 / bfm_loss_grad_l1 / bfm_loss_seg
    losses = sum(loss_dict[k] * weight_dict[k])
    scaler.scale(losses).backward()                   -> bfm_head_bwd, bfm_normalize_bwd, backward.backbone_backward
    scaler.unscale_(optimizer); clip_gradients(...)   -> bfm_grad_sumsq (+ non-finite flag), per-parameter coefficient
    scaler.step(optimizer); scaler.update()           -> bfm_adamw_step, LossScaler.update

with the criterion of Trainer/models/criterion.py (SetMultiCriterion: sum over the samples / all_samples) restricted to
the supervised dense heads: T1 T2 FLAIR CT (+_grad, optional <key>_DM weights), SR(+_grad), distance, registration
(+_grad), bias_field_log (l1 | l2, soft mask 1 - seg[:, 0]), seg_ce, seg_dice.  Any other loss name raises.

Everything numeric runs in the HIP library; torch holds buffers, adds the per-sample gradients and runs the RCCL
all-reduce.  There is no CPU path: without the extension or a HIP device construction fails.
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import backward as BW
from . import models as M

IMAGE_KEYS = ("T1", "T2", "FLAIR", "CT")
SUPPORTED = set(IMAGE_KEYS) | {k + "_grad" for k in IMAGE_KEYS} | {
    "SR", "SR_grad", "distance", "surface", "registration", "registration_grad", "bias_field_log", "seg_ce", "seg_dice",
    "pathol_ce", "pathol_dice"}


class LossScaler:
    """torch.cuda.amp.GradScaler's state machine (scripts/train.py:164, Trainer/engine.py:139-147): the loss is
    multiplied by `scale`, gradients are divided by it before clipping, a step with a non-finite gradient is skipped and
    halves the scale, `growth_interval` clean steps in a row double it."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval = int(growth_interval)
        self.enabled = bool(enabled)
        self._good = 0

    def update(self, found_inf):
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor
            self._good = 0
        else:
            self._good += 1
            if self._good == self.growth_interval:
                self.scale *= self.growth_factor
                self._good = 0


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0.0):
    """utils/misc.py:1265-1276: linear warm-up then half a cosine, one value per iteration."""
    import numpy as np
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 else np.array([])
    it = np.arange(epochs * niter_per_ep - warm_iters)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
    return np.concatenate((warm, sched))


def multistep_scheduler(base_value, lr_drops, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0.0, gamma=0.1):
    """utils/misc.py:1251-1262 (the default lr schedule of cfgs/trainer/default_train.yaml).  Kept quirk: the milestones
    index the array WITHOUT its warm-up part, so every drop lands warmup_epochs later than lr_drops says."""
    import numpy as np
    warm_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warm_iters) if warmup_epochs > 0 else np.array([])
    sched = np.ones(epochs * niter_per_ep - warm_iters) * base_value
    for m in lr_drops:
        sched[m * niter_per_ep:] *= gamma
    return np.concatenate((warm, sched))


def allreduce_mean_(grads, group=None, events=None):
    """DDP's gradient averaging (scripts/train.py:153-158) as ONE flat all-reduce over all parameters: xGMI rings are
    per-link bound, so a single ~100 MB bucket beats many small ones.  In place; no-op without a process group.
    events: a list that receives (start, end) device events around the collective (it is not overlapped with anything: what
    it takes is what it exposes; bench.py's config 5 block reports it)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    keys = list(grads.keys())
    flat = torch.cat([grads[k].reshape(-1) for k in keys])
    if events is not None and flat.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if events is not None and flat.is_cuda:
        e1.record()
        del events[:]
        events.extend([e0, e1, flat.numel() * flat.element_size()])
    flat /= dist.get_world_size(group)
    off = 0
    for k in keys:
        n = grads[k].numel()
        grads[k].copy_(flat[off:off + n].view_as(grads[k]))
        off += n
    return grads


class _Sink:
    """What the backward pass sees of a GradStore (engine.grad_sink): `out` hands a kernel its slot, `done` adds the
    earlier samples' sum for that slot and counts its bucket down."""

    def __init__(self, store, partial, row_of):
        self.store, self.partial, self.row_of = store, partial, row_of

    def out(self, name, shape):
        return self.store.out(name, shape)

    def done(self, name):
        if self.partial is not None:
            v = self.store.views[name]
            if name in ("head.weight_all", "head.bias_all"):
                kind = "weight" if name == "head.weight_all" else "bias"
                for task, (r0, n) in self.row_of.items():
                    v[r0:r0 + n].add_(self.partial["head.final_conv_%s.%s" % (task, kind)])
            else:
                v.add_(self.partial[name])
        self.store.done(name)


class GradStore:
    """All parameter gradients of one iteration in ONE persistent flat fp32 buffer, laid out in the order the backward pass
    completes them (heads, decoders last to first, encoders last to first), cut into a few buckets of about equal bytes.
    The weight-gradient kernels write straight into their slot (`out`), `done` counts a bucket down, and a bucket whose last
    tensor is complete is all-reduced on a communication stream while the backward pass goes on -- DDP's bucketed overlap
    (scripts/train.py:153-158) without DDP's copies: no torch.cat of 252 tensors in front of the collective and no 252
    copy_ launches behind it.  The optimiser reads the slots in place; the mean's 1 / world is folded into
    its gradient scale.  xGMI rings are per-link bound, so buckets are large (default 6 of ~180 MB for the 264 M-parameter
    net; BFM_GRAD_BUCKETS).  Every rank issues the same collectives in the same order whatever its loss turns out to be; an
    iteration that is skipped afterwards only wastes them."""

    def __init__(self, named_shapes, device, n_buckets=None):
        self.device = device
        nb = int(os.environ.get("BFM_GRAD_BUCKETS", "6")) if n_buckets is None else int(n_buckets)
        self.names = [n for n, _ in named_shapes]
        sizes = [int(np.prod(s)) for _, s in named_shapes]
        offs = np.concatenate([[0], np.cumsum([(n + 3) // 4 * 4 for n in sizes])])      # 16-byte aligned slots
        self.total = int(offs[-1])
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.views = OrderedDict()
        for (name, shape), o, n in zip(named_shapes, offs[:-1], sizes):
            self.views[name] = self.flat[int(o):int(o) + n].view(tuple(shape))
        # bucket b = tensors [first[b], first[b + 1]): cut where the running byte count passes b / nb of the total
        nb = max(1, min(nb, len(self.names)))
        self.first = [0]
        for i in range(1, len(self.names)):
            if len(self.first) < nb and offs[i] >= self.total * len(self.first) / nb:
                self.first.append(i)
        self.first.append(len(self.names))
        self.range = [(int(offs[self.first[b]]), int(offs[self.first[b + 1]])) for b in range(len(self.first) - 1)]
        self.bucket_of = {}
        for b in range(len(self.first) - 1):
            for i in range(self.first[b], self.first[b + 1]):
                self.bucket_of[self.names[i]] = b
        self.comm = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.group = None
        self.works = []
        self.pending = []
        self.launched = []
        self.events = None
        self.active = False

    def begin(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.pending = [self.first[b + 1] - self.first[b] for b in range(len(self.range))]
        self.launched = [False] * len(self.range)
        self.works = []
        self.events = None
        self.active = True

    def out(self, name, shape):
        v = self.views[name]
        if tuple(v.shape) != tuple(shape):
            raise L.BfmError("gradient slot %s has shape %s, the kernel writes %s" % (name, tuple(v.shape), tuple(shape)))
        return v

    def done(self, name):
        """The kernel that completes `name` has been enqueued on the current stream."""
        b = self.bucket_of[name]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.world <= 1:
            return
        import torch.distributed as dist
        lo, hi = self.range[b]
        seg = self.flat[lo:hi]
        if self.comm is None:
            self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record()                                                   # the bucket's last producer, on the compute stream
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ev)
            self.works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Backward is over: launch what is left (a slot nobody wrote is all-reduced as the zeros it holds), then make the
        compute stream wait for the collectives.  events = (backward end, collectives end, bytes): the time between the two
        is what the all-reduce EXPOSES (bench.py: config5.allreduce_exposed_ms)."""
        for b in range(len(self.range)):
            if not self.launched[b]:
                self._launch(b)
        self.active = False
        if self.world <= 1:
            return
        if self.comm is None:
            for w in self.works:
                w.wait()
            return
        main = torch.cuda.current_stream(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        with torch.cuda.stream(self.comm):
            for w in self.works:
                w.wait()
            e1.record(self.comm)
        main.wait_event(e1)
        self.events = (e0, e1, self.total * 4)
        self.works = []


# bfm_adam_tensor_t (include/brainfm_hip.h): p, g, m, v, n, grad_scale, bias1, bias2_sqrt, first_chunk, reserved
_ADAM_DESC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("grad_scale", "<f4"),
                       ("bias1", "<f4"), ("bias2_sqrt", "<f4"), ("first_chunk", "<i4"), ("reserved", "<i8")])
assert _ADAM_DESC.itemsize == 64


class TrainStep:
    """Parameters live in the engine (layer .w_raw / .gamma / .beta) and the tail (head_w / head_b); `step` updates
    them in place and drops the packed-weight caches so that the next forward repacks."""

    def __init__(self, engine, tail, loss_names, loss_weights, weights_ce, all_samples, max_surf_distance=3.0,
                 bias_field_log_type="l2", lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, clip_max_norm=0.0,
                 scaler=None):
        bad = [n for n in loss_names if n not in SUPPORTED]
        if bad:
            raise L.BfmError("losses outside the HIP training path: %s" % bad)
        self.eng, self.tail = engine, tail
        self.lib = engine.lib
        self.dev = engine.device
        self.loss_names = list(loss_names)
        self.loss_weights = dict(loss_weights)
        self.all_samples = float(all_samples)
        self.max_dist = float(max_surf_distance)
        self.bias_l2 = 1 if bias_field_log_type == "l2" else 0
        self.wce = torch.as_tensor(weights_ce).to(device=self.dev, dtype=torch.float32).contiguous()
        self.lr, self.wd, self.betas, self.eps, self.clip = float(lr), float(weight_decay), betas, float(eps), float(clip_max_norm)
        self.scaler = scaler if scaler is not None else LossScaler(enabled=False)
        self.t = 0                                          # iterations stepped so far
        self.steps = {}                                     # per-parameter AdamW step counts (a head without an active
                                                            # loss is not stepped: its .grad is None in the reference)
        self._touched_heads = set()
        self._active_rows = set()
        self.state = {}                                   # name -> (m, v)
        nseg = tail.desc.n_seg
        self.sample_lanes = max(1, int(os.environ.get("BFM_TRAIN_LANES", "2")))
        self._lane_streams = []
        self._lane = 0
        self._ws_lane = {}
        self._ws_bytes = (self.lib.bfm_loss_workspace(max(nseg, 1)), self.lib.bfm_loss_l1_multi_workspace())

    @property
    def _ws(self):
        return self._lane_ws()[0]

    @property
    def _ws_multi(self):
        return self._lane_ws()[1]

    def _lane_ws(self):
        """Reduction scratch of the loss kernels, one set per sample lane."""
        w = self._ws_lane.get(self._lane)
        if w is None:
            w = self._ws_lane[self._lane] = tuple(torch.empty(b, dtype=torch.uint8, device=self.dev) for b in self._ws_bytes)
        return w

    # ------------------------------------------------------------------ parameters
    def parameters(self):
        """{reference parameter name: device tensor (views for the heads)}."""
        p = OrderedDict()
        for pair in self.eng.enc + self.eng.dec:
            for ly in pair:
                p[ly.name + ".groupnorm.weight"] = ly.gamma
                p[ly.name + ".groupnorm.bias"] = ly.beta
                p[ly.name + ".conv.weight"] = ly.w_raw
        for task, (r0, n) in self.tail.row_of.items():
            p["head.final_conv_%s.weight" % task] = self.tail.head_w[r0:r0 + n]
            p["head.final_conv_%s.bias" % task] = self.tail.head_b[r0:r0 + n]
        return p

    def grad_store(self):
        """The persistent gradient buffer (GradStore), slots in the order the backward pass completes them."""
        st = self.__dict__.get("_grad_store")
        if st is None:
            named = [("head.weight_all", (self.tail.n_out, self.tail.c_feat)), ("head.bias_all", (self.tail.n_out,))]
            for pair in list(reversed(self.eng.dec)) + list(reversed(self.eng.enc)):
                for ly in reversed(pair):
                    named.append((ly.name + ".conv.weight", tuple(ly.w_raw.shape)))
                    named.append((ly.name + ".groupnorm.weight", tuple(ly.gamma.shape)))
                    named.append((ly.name + ".groupnorm.bias", tuple(ly.beta.shape)))
            st = self._grad_store = GradStore(named, self.dev)
        return st

    def _weights_changed(self):
        """Right after the optimiser: every packed form a layer holds (forward variants, skip half, up-folded, transposed
        data-gradient layer) is rebuilt in place, so the next pass packs nothing lazily."""
        self.eng.repack_all(refresh=BW.refresh_dgrad)
        hw = torch.zeros(1, dtype=torch.float32, device=self.dev)
        L.check(self.lib.bfm_absmax_f32(L.ptr(self.tail.head_w), 1, self.tail.head_w.numel(), self.tail.head_w.numel(),
                                        L.ptr(hw), L.stream_ptr()), "absmax head_w")
        self.tail.desc.head_wmax = float(hw.item())

    # ------------------------------------------------------------------ checkpoints (scripts/train.py:205-214)
    def _ref_shape(self, name, t):
        if name.startswith("head.") and name.endswith(".weight"):
            return tuple(t.shape) + (1, 1, 1)                      # nn.Conv3d(C, n, 1) weight
        return tuple(t.shape)

    def state_dict(self):
        """The model's state_dict under the reference's names and shapes (CPU copies)."""
        return OrderedDict((k, v.detach().reshape(self._ref_shape(k, v)).cpu().clone()) for k, v in self.parameters().items())

    def optimizer_state_dict(self, lr=None, weight_decay=None):
        """torch.optim.AdamW.state_dict() layout for the single parameter group of scripts/train.py:46-53 (parameters in
        model.named_parameters() order == self.parameters() order)."""
        names = list(self.parameters().keys())
        state = {}
        for i, k in enumerate(names):
            if k in self.state:
                m, v = self.state[k]
                shp = self._ref_shape(k, self.parameters()[k])
                state[i] = {"step": torch.tensor(float(self.steps.get(k, self.t))), "exp_avg": m.reshape(shp).cpu().clone(),
                            "exp_avg_sq": v.reshape(shp).cpu().clone()}
        group = {"lr": self.lr if lr is None else lr, "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.wd if weight_decay is None else weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def save_checkpoint(self, path, epoch=0, **extra):
        """torch.save of {'model', 'optimizer', 'epoch', ...}: what utils.save_on_master writes in scripts/train.py, and
        what brainfm_amd.models.load_checkpoint / the reference's load_checkpoint read back."""
        ckp = {"model": self.state_dict(), "optimizer": self.optimizer_state_dict(), "epoch": int(epoch),
               "scaler_scale": self.scaler.scale}
        ckp.update(extra)
        torch.save(ckp, path)

    def load_checkpoint(self, path):
        """Parameters (matched by name suffix, as utils/checkpoint.py:558-571) and AdamW moments back into the engine."""
        ckp = M.read_checkpoint_file(path)
        key = next((k for k in ckp if "model" in k), None)
        sd = ckp[key] if key is not None else ckp
        params = self.parameters()
        for k, p in params.items():
            cands = [lk for lk in sd if lk == k or lk.endswith("." + k) or k.endswith("." + lk)]
            if not cands:
                raise KeyError("checkpoint has no tensor matching '%s'" % k)
            src = sd[max(cands, key=len)]
            if src.numel() != p.numel():
                raise ValueError("shape mismatch for %s: %s vs %s" % (k, tuple(src.shape), tuple(p.shape)))
            p.copy_(src.reshape(p.shape).to(device=self.dev, dtype=torch.float32))
        self._weights_changed()
        opt = ckp.get("optimizer")
        if opt and opt.get("state"):
            names = list(params.keys())
            self.state = {}
            for i, st in opt["state"].items():
                k = names[int(i)]
                self.state[k] = (st["exp_avg"].reshape(-1).to(device=self.dev, dtype=torch.float32).contiguous(),
                                 st["exp_avg_sq"].reshape(-1).to(device=self.dev, dtype=torch.float32).contiguous())
                self.steps[k] = int(float(st["step"]))
            self.t = max(self.steps.values()) if self.steps else 0
        if "scaler_scale" in ckp and self.scaler.enabled:
            self.scaler.scale = float(ckp["scaler_scale"])
        return ckp

    # ------------------------------------------------------------------ losses
    def _col(self, task, j=0):
        r0, n = self.tail.row_of[task]
        if j >= n:
            raise L.BfmError("head '%s' has %d channels" % (task, n))
        return r0 + j

    def _t(self, x, shape_tail):
        x = torch.as_tensor(x).to(device=self.dev, dtype=torch.float32).contiguous()
        if tuple(x.shape[-3:]) != tuple(shape_tail):
            raise L.BfmError("target of shape %s does not match the volume %s" % (tuple(x.shape), tuple(shape_tail)))
        return x

    def _sample_losses(self, raw, dims, target, sample, dRaw, vals, scale, rows=False):
        """Launch every loss of one sample; values land in `vals` (device fp64, one slot list per loss name).
        rows: raw / dRaw are (n_out, nvox) rows instead of channels-last (nvox, n_out)."""
        lib, st = self.lib, L.stream_ptr()
        D, H, W = dims
        nvox = D * H * W
        n_out = self.tail.n_out
        ws, wsn = L.ptr(self._ws), self._ws.numel()
        slots = {}
        k = 0
        dense = []                                        # (slot index, col, target, weight, mask, clamp, l2, coef): one launch
        grads_l1 = []                                     # (slot index, col, target, weight, coef): the gradient-L1 entries, one launch
        keep = []                                         # tensors the deferred launch reads
        active = set()                                    # head rows some loss of this sample differentiates
        pathol_done = False

        def slot(name, n=1):
            nonlocal k
            slots.setdefault(name, []).append((k, n))
            k += n
            return C.c_void_p(vals.data_ptr() + 8 * (k - n))

        def seg_loss(r0, ns, tgt, c_ce, c_dice):
            P = torch.empty(nvox * ns, dtype=torch.float32, device=self.dev)     # [nvox][ns], rows: [ns][nvox]
            if rows:
                L.check(lib.bfm_loss_seg_rows(L.ptr(raw), nvox, n_out, r0, ns, L.ptr(tgt), L.ptr(self.wce), L.ptr(self.wce),
                                              nvox, c_ce, c_dice, L.ptr(P), L.ptr(dRaw), slot("seg", 1 + 2 * ns), ws, wsn, st),
                        "loss_seg_rows")
            else:
                L.check(lib.bfm_loss_seg(L.ptr(raw), n_out, r0, ns, L.ptr(tgt), L.ptr(self.wce), L.ptr(self.wce), nvox,
                                         c_ce, c_dice, L.ptr(P), L.ptr(dRaw), slot("seg", 1 + 2 * ns), ws, wsn, st), "loss_seg")

        for name in self.loss_names:
            coef = scale * self.loss_weights.get("loss_" + name, 0.0) / self.all_samples
            if name in IMAGE_KEYS or (name.endswith("_grad") and name[:-5] in IMAGE_KEYS) or name in ("SR", "SR_grad"):
                is_grad = name.endswith("_grad")
                key = name[:-5] if is_grad else name
                if key == "SR":
                    head, tgt, wt = "high_res_residual", sample.get("high_res_residual"), None
                else:
                    head, tgt = key, target.get(key)
                    wt = (1.0 - self._t(target[key + "_DM"], dims)) if (key + "_DM") in target else None
                if tgt is None or head not in self.tail.row_of:
                    continue                                          # criterion.py:274-281: shape mismatch -> 0
                tgt = self._t(tgt, dims)
                co = self._col(head)
                active.add(co)
                if is_grad:
                    slot(name)
                    grads_l1.append((k - 1, co, tgt, wt, coef))
                else:
                    slot(name)
                    dense.append((k - 1, co, tgt, wt, None, 0.0, 0, coef))
            elif name in ("distance", "surface", "registration", "registration_grad"):
                # criterion.py:175-185: loss_image / loss_image_grad of the whole multi-channel map (surface: 8 channels,
                # Trainer/models/__init__.py:103-106; its head has no processor, so no clamp)
                head = "registration" if name.startswith("registration") else name
                if head not in target or head not in self.tail.row_of:
                    continue
                nch = self.tail.row_of[head][1]
                tgt = self._t(target[head], dims).reshape(-1, D, H, W)
                if tgt.shape[0] != nch:
                    continue
                for j in range(nch):
                    co = self._col(head, j)
                    active.add(co)
                    if name == "registration_grad":
                        slot(name)
                        grads_l1.append((k - 1, co, tgt[j], None, coef / nch))
                    else:
                        clampv = self.max_dist if head == "distance" else 0.0
                        slot(name)
                        dense.append((k - 1, co, tgt[j], None, None, clampv, 0, coef / nch))
            elif name in ("pathol_ce", "pathol_dice"):
                # criterion.py:193-212 on sigmoid(raw) (PatholProcessor): both names of a sample in ONE bfm_loss_pathol call,
                # issued when the first of them comes up
                if pathol_done:
                    continue
                pathol_done = True
                if "pathology" not in target or "pathology" not in self.tail.row_of:
                    continue                                          # shape mismatch / no head -> 0, as the reference
                tgt = self._t(target["pathology"], dims).reshape(-1)
                if tgt.numel() != nvox:
                    continue
                co = self._col("pathology")
                active.add(co)
                cf = {n_: scale * self.loss_weights.get("loss_" + n_, 0.0) / self.all_samples
                      for n_ in ("pathol_ce", "pathol_dice") if n_ in self.loss_names}
                p_ce = slot("pathol_ce") if "pathol_ce" in cf else None
                p_di = slot("pathol_dice") if "pathol_dice" in cf else None
                wsp = torch.empty(lib.bfm_loss_pathol_workspace(), dtype=torch.uint8, device=self.dev)
                keep.extend([tgt, wsp])
                L.check(lib.bfm_loss_pathol(L.ptr(raw), co * raw.stride(0) if rows else co, 1 if rows else n_out, L.ptr(tgt),
                                            nvox, cf.get("pathol_ce", 0.0), cf.get("pathol_dice", 0.0), L.ptr(dRaw), p_ce, p_di,
                                            L.ptr(wsp), wsp.numel(), st), "loss_pathol")
            elif name == "bias_field_log":
                if "bias_field_log" not in sample or "bias_field_log" not in self.tail.row_of:
                    continue
                mask = 1.0 - self._t(target["segmentation"], dims).reshape(-1, D, H, W)[0]
                tgt = self._t(sample["bias_field_log"], dims)
                slot(name)
                active.add(self._col("bias_field_log"))
                dense.append((k - 1, self._col("bias_field_log"), tgt, None, mask, 0.0, self.bias_l2, coef))
            elif name == "seg_ce":
                # CE and Dice share the softmax: one launch covers both names
                r0, ns = self.tail.row_of["segmentation"]
                active.update(range(r0, r0 + ns))
                tgt = self._t(target["segmentation"], dims).reshape(ns, D, H, W)
                c_dice = (scale * self.loss_weights.get("loss_seg_dice", 0.0) / self.all_samples
                          if "seg_dice" in self.loss_names else 0.0)
                seg_loss(r0, ns, tgt, coef, c_dice)
            elif name == "seg_dice":
                if "seg_ce" not in self.loss_names:
                    r0, ns = self.tail.row_of["segmentation"]
                    active.update(range(r0, r0 + ns))
                    tgt = self._t(target["segmentation"], dims).reshape(ns, D, H, W)
                    seg_loss(r0, ns, tgt, 0.0, coef)
        # every l1 / l2 entry in batches of 32 per launch (one pass over raw each); results go to a staging row and from
        # there to their slots
        for b0 in range(0, len(dense), 32):
            batch = dense[b0:b0 + 32]
            n = len(batch)
            cols = (C.c_int32 * n)(*[e[1] for e in batch])
            l2s = (C.c_int32 * n)(*[e[6] for e in batch])
            clamps = (C.c_float * n)(*[e[5] for e in batch])
            coefs = (C.c_float * n)(*[e[7] for e in batch])
            tg = (C.c_void_p * n)(*[e[2].data_ptr() for e in batch])
            wt = (C.c_void_p * n)(*[(e[3].data_ptr() if e[3] is not None else None) for e in batch])
            mk = (C.c_void_p * n)(*[(e[4].data_ptr() if e[4] is not None else None) for e in batch])
            keep.extend([e[2] for e in batch] + [e[3] for e in batch] + [e[4] for e in batch])
            stage = torch.empty(n, dtype=torch.float64, device=self.dev)
            wsm = self._ws_multi
            if rows:
                L.check(lib.bfm_loss_l1_multi_rows(L.ptr(raw), nvox, n_out, nvox, n, cols, l2s, clamps, coefs, tg, wt, mk,
                                                   L.ptr(dRaw), L.ptr(stage), L.ptr(wsm), wsm.numel(), st), "loss_l1_multi_rows")
            else:
                L.check(lib.bfm_loss_l1_multi(L.ptr(raw), n_out, nvox, n, cols, l2s, clamps, coefs, tg, wt, mk, L.ptr(dRaw),
                                              L.ptr(stage), L.ptr(wsm), wsm.numel(), st), "loss_l1_multi")
            idx = torch.tensor([e[0] for e in batch], dtype=torch.int64, device=self.dev)
            vals.index_copy_(0, idx, stage)
        # every gradient-L1 entry in one launch (distinct columns per launch: two entries on one column go to two launches)
        pending = list(grads_l1)
        while pending:
            batch, rest, seen = [], [], set()
            for e in pending:
                if e[1] in seen or len(batch) == 32:
                    rest.append(e)
                else:
                    seen.add(e[1])
                    batch.append(e)
            pending = rest
            n = len(batch)
            cols = (C.c_int32 * n)(*[e[1] for e in batch])
            coefs = (C.c_float * n)(*[e[4] for e in batch])
            tg = (C.c_void_p * n)(*[e[2].data_ptr() for e in batch])
            wt = (C.c_void_p * n)(*[(e[3].data_ptr() if e[3] is not None else None) for e in batch])
            keep.extend([e[2] for e in batch] + [e[3] for e in batch])
            stage = torch.empty(n, dtype=torch.float64, device=self.dev)
            wsm = self._ws_multi
            if rows:
                L.check(lib.bfm_loss_grad_l1_multi_rows(L.ptr(raw), nvox, n_out, n, cols, coefs, tg, wt, D, H, W, L.ptr(dRaw),
                                                        L.ptr(stage), L.ptr(wsm), wsm.numel(), st), "loss_grad_l1_multi_rows")
            else:
                L.check(lib.bfm_loss_grad_l1_multi(L.ptr(raw), n_out, n, cols, coefs, tg, wt, D, H, W, L.ptr(dRaw), L.ptr(stage),
                                                   L.ptr(wsm), wsm.numel(), st), "loss_grad_l1_multi")
            idx = torch.tensor([e[0] for e in batch], dtype=torch.int64, device=self.dev)
            vals.index_copy_(0, idx, stage)
        self._keep = keep
        self._active_rows = active
        return slots, k

    def _finish_losses(self, per_sample, nvox):
        """Host side of SetMultiCriterion: fold channels / samples of the fp64 slots into {loss_<name>: value}."""
        out = OrderedDict()
        wdice = self.wce.double().cpu()
        for slots, vals in per_sample:
            v = vals.cpu()
            for name, lst in slots.items():
                if name == "seg":
                    (k0, n), = lst
                    ns = (n - 1) // 2
                    ce = float(v[k0]) / nvox
                    num, den = v[k0 + 1:k0 + 1 + ns], v[k0 + 1 + ns:k0 + 1 + 2 * ns]
                    dice = float((wdice * (1.0 - 2.0 * num / torch.clamp(den, min=1e-5))).sum())
                    if "seg_ce" in self.loss_names:
                        out["loss_seg_ce"] = out.get("loss_seg_ce", 0.0) + ce / self.all_samples
                    if "seg_dice" in self.loss_names:
                        out["loss_seg_dice"] = out.get("loss_seg_dice", 0.0) + dice / self.all_samples
                else:
                    val = sum(float(v[k0]) for k0, _ in lst) / len(lst)
                    out["loss_" + name] = out.get("loss_" + name, 0.0) + val / self.all_samples
        # keep the criterion's loss order
        return OrderedDict((("loss_" + n), out["loss_" + n]) for n in self.loss_names if ("loss_" + n) in out)

    # ------------------------------------------------------------------ forward + backward
    def _one_sample(self, x, target, sample, scale):
        """Forward, losses and backward of one augmented sample on the current stream.  Returns (grads, slots, vals)."""
        eng, tail, lib = self.eng, self.tail, self.lib
        dims = tuple(x.shape[-3:])
        nvox = dims[0] * dims[1] * dims[2]
        st = L.stream_ptr()
        x_cl = eng.to_cl(x)
        feats, tape = BW.backbone_forward_train(eng, x_cl, dims)
        feat_last = feats[-1][0]
        n_out, cf = tail.n_out, tail.c_feat
        # head outputs as rows of nvox values wherever the one-pass heads backward exists (64 features, <= 96 outputs,
        # <= 64 classes: every shipped head set); BFM_TRAIN_ROWS=0 keeps channels-last
        rows = (os.environ.get("BFM_TRAIN_ROWS", "1") != "0" and cf == 64 and n_out <= 96 and
                tail.row_of.get("segmentation", (0, 0))[1] <= 64)
        raw, fn = tail.run_raw(feat_last, dims, want_feat=True, rows=rows)
        if fn is None:
            fn = feat_last
        dRaw = torch.zeros_like(raw)
        vals = torch.zeros(4 * len(self.loss_names) + 2 * tail.n_out + 8, dtype=torch.float64, device=self.dev)
        slots, _ = self._sample_losses(raw, dims, target, sample, dRaw, vals, scale, rows=rows)
        sink = getattr(eng, "grad_sink", None)
        dW = sink.out("head.weight_all", (n_out, cf)) if sink is not None else torch.empty((n_out, cf), dtype=torch.float32, device=self.dev)
        db = sink.out("head.bias_all", (n_out,)) if sink is not None else torch.empty(n_out, dtype=torch.float32, device=self.dev)
        dFn = torch.empty((nvox, cf), dtype=torch.float32, device=self.dev)
        wsb = torch.empty(lib.bfm_head_bwd_workspace(n_out, cf, nvox), dtype=torch.uint8, device=self.dev)
        if rows:
            L.check(lib.bfm_head_bwd_rows(L.ptr(dRaw), nvox, L.ptr(fn), L.ptr(tail.head_w), n_out, cf, nvox, L.ptr(dW),
                                          L.ptr(db), L.ptr(dFn), L.ptr(wsb), wsb.numel(), st), "head_bwd_rows")
        else:
            L.check(lib.bfm_head_bwd(L.ptr(dRaw), L.ptr(fn), L.ptr(tail.head_w), n_out, cf, nvox, L.ptr(dW), L.ptr(db),
                                     L.ptr(dFn), L.ptr(wsb), wsb.numel(), st), "head_bwd")
        if sink is not None:
            sink.done("head.weight_all")
            sink.done("head.bias_all")
        if eng.unit_feat:
            dfeat = torch.empty_like(dFn)
            L.check(lib.bfm_normalize_bwd(L.ptr(feat_last), L.ptr(dFn), cf, nvox, 1e-12, L.ptr(dfeat), st), "normalize_bwd")
        else:
            dfeat = dFn
        g = BW.backbone_backward(eng, tape, [None] * (len(feats) - 1) + [dfeat.view(dims + (cf,))])
        for task, (r0, n) in tail.row_of.items():
            g["head.final_conv_%s.weight" % task] = dW[r0:r0 + n]
            g["head.final_conv_%s.bias" % task] = db[r0:r0 + n]
        # heads no loss of this sample reached: the reference leaves their .grad None (the all-zero rows above only keep
        # the gradient dictionary's layout fixed for the flat all-reduce); loss_and_grads collects who was reached
        self._touched_heads.update(task for task, (r0, n) in tail.row_of.items()
                                   if any(r in self._active_rows for r in range(r0, r0 + n)))
        return g, slots, vals

    @L.on_device(lambda self, *a, **k: self.dev)
    def loss_and_grads(self, xs, target, samples):
        """xs: list of (1,C,D,H,W) inputs (one per augmented sample); target / samples as the reference's dicts
        (NCDHW tensors).  Returns (loss_dict, total, grads) with grads = d(scale * total)/d(parameter) summed over the
        samples in sample order (scale = the loss scaler's).
        With sample_lanes > 1 (default 2, BFM_TRAIN_LANES) consecutive samples run on separate streams from the second
        iteration on -- the packed weights are all in place by then (apply() rebuilds them eagerly) and the conv variants
        are tuned -- so that one sample's small kernels and launch tails hide under the other's convolutions, as the two
        tiles in flight of the inference path do; the gradients are still added on the caller's stream in sample order,
        so the result is the same bits."""
        eng = self.eng
        scale = self.scaler.scale
        n = len(xs)
        self._touched_heads = set()
        lanes = self.sample_lanes if (n > 1 and self.t >= 1) else 1
        nvox = None
        results = []
        # Where the gradients of the iteration are collected.  With more than one rank -- or with one sample lane -- the LAST
        # sample's kernels write straight into the persistent GradStore and start its buckets' all-reduces as each bucket
        # completes, i.e. under that sample's own backward pass; the earlier samples' sum (in sample order; a + b == b + a
        # bit for bit) is added slot by slot in front of each `done`.  That is DDP with gradient accumulation: no_sync for
        # all samples but the last.  The last sample then runs on the caller's stream after the lanes have joined.  One
        # rank with two lanes keeps every sample on the lanes (nothing to all-reduce).
        import torch.distributed as dist
        group = getattr(self, "_group", None)
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # only inside step(): a bare loss_and_grads() call must neither start collectives nor hand out views of a buffer the
        # next call overwrites
        use_store = (self.__dict__.pop("_store_next", False) and os.environ.get("BFM_GRAD_STORE", "1") != "0" and
                     (lanes <= 1 or world > 1))
        n_head = n - 1 if use_store else n                 # samples that go the ordinary way
        store = None
        if lanes <= 1 or n_head <= 1:
            for x, sample in zip(xs[:n_head], samples[:n_head]):
                results.append(self._one_sample(x, target, sample, scale) + (None,))
        else:
            main = torch.cuda.current_stream(self.dev)
            while len(self._lane_streams) < lanes:
                self._lane_streams.append(torch.cuda.Stream(device=self.dev))
            start = torch.cuda.Event()
            start.record(main)
            seen = eng.pack_count
            for i, (x, sample) in enumerate(zip(xs[:n_head], samples[:n_head])):
                k = i % lanes
                st = self._lane_streams[k]
                st.wait_event(start)
                if eng.pack_count != seen:                 # a packed form was created lazily while issuing an earlier
                    for r in results:                      # sample (new shape): later samples must see it finished
                        st.wait_event(r[3])
                    seen = eng.pack_count
                eng.lane, self._lane = k, k
                try:
                    with torch.cuda.stream(st):
                        g, slots, vals = self._one_sample(x, target, sample, scale)
                        ev = torch.cuda.Event()
                        ev.record(st)
                finally:
                    eng.lane, self._lane = 0, 0
                results.append((g, slots, vals, ev))
            for g, slots, vals, ev in results:
                main.wait_event(ev)
                vals.record_stream(main)
                for t_ in g.values():
                    t_.record_stream(main)
        if use_store:
            store = self.grad_store()
            store.begin(group)
            part = None
            for r in results:
                part = r[0] if part is None else OrderedDict((k_, part[k_] + v_) for k_, v_ in r[0].items())
            eng.grad_sink = _Sink(store, part, self.tail.row_of)
            try:
                results.append(self._one_sample(xs[-1], target, samples[-1], scale) + (None,))
            finally:
                eng.grad_sink = None
        grads = None
        per_sample = []
        self._store_live = store
        for x, (g, slots, vals, _) in zip(xs, results):
            nvox = x.shape[-3] * x.shape[-2] * x.shape[-1]
            per_sample.append((slots, vals))
            if store is not None:
                continue                                   # the last sample's slots already hold the sum
            if grads is None:
                grads = g
            else:
                for k_, v_ in g.items():
                    grads[k_] = grads[k_] + v_
        if store is not None:
            grads = results[-1][0]
        loss_dict = self._finish_losses(per_sample, nvox)
        total = sum(v * self.loss_weights.get(k, 0.0) for k, v in loss_dict.items() if k in self.loss_weights)
        return loss_dict, total, grads

    # ------------------------------------------------------------------ optimiser
    @L.on_device(lambda self, *a, **k: self.dev)
    def apply(self, grads, lr=None, weight_decay=None, grad_div=1.0):
        """unscale -> per-parameter clip (utils/misc.py:1329-1338) -> AdamW -> scaler.update.  Returns
        (stepped, norms): stepped is False when a non-finite gradient made the scaler skip the step.
        grad_div: the gradients hold a SUM over that many ranks (GradStore); the mean's division is folded into the unscale."""
        lib, st = self.lib, L.stream_ptr()
        lr = self.lr if lr is None else float(lr)
        wd = self.wd if weight_decay is None else float(weight_decay)
        params = self.parameters()
        names = [k for k in params if k in grads]
        sums = torch.zeros(len(names), dtype=torch.float64, device=self.dev)
        flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        multi = os.environ.get("BFM_ADAM_MULTI", "1") != "0"
        for k in names:
            grads[k] = grads[k].contiguous()
            if not params[k].is_contiguous():
                raise L.BfmError("parameter %s is not contiguous" % k)
            if k not in self.state:
                self.state[k] = (torch.zeros(params[k].numel(), dtype=torch.float32, device=self.dev),
                                 torch.zeros(params[k].numel(), dtype=torch.float32, device=self.dev))
        if multi:
            # every tensor's sum of squares in one launch (+ a fold): descriptors and chunk map in device memory
            CH = 65536
            desc = np.zeros(len(names), dtype=_ADAM_DESC)
            first, cmap = 0, []
            for i, k in enumerate(names):
                n = params[k].numel()
                m, v = self.state[k]
                desc[i] = (params[k].data_ptr(), grads[k].data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.0, 1.0, 1.0, first, 0)
                nch = (n + CH - 1) // CH
                cmap.extend([i] * nch)
                first += nch
            key = tuple(params[k].numel() for k in names)
            if getattr(self, "_chunk_key", None) != key:
                self._chunk_key = key
                self._chunk_map = torch.tensor(cmap, dtype=torch.int32, device=self.dev)
                self._chunk_ws = torch.empty(len(cmap), dtype=torch.float64, device=self.dev)
            tdev = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(self.dev)
            L.check(lib.bfm_grad_sumsq_multi(L.ptr(tdev), len(names), L.ptr(self._chunk_map), len(cmap), CH, L.ptr(sums),
                                             L.ptr(flag), L.ptr(self._chunk_ws), self._chunk_ws.numel() * 8, st), "grad_sumsq_multi")
        else:
            for i, k in enumerate(names):
                g = grads[k]
                L.check(lib.bfm_grad_sumsq(L.ptr(g), g.numel(), C.c_void_p(sums.data_ptr() + 8 * i), L.ptr(flag),
                                           L.ptr(self._ws), self._ws.numel(), st), "grad_sumsq " + k)
        inv = 1.0 / (self.scaler.scale * grad_div)
        found_inf = bool(flag.item())
        norms = [math.sqrt(v) * inv if math.isfinite(v) else float("inf") for v in sums.cpu().tolist()]
        found_inf = found_inf or any(not math.isfinite(v) for v in norms)
        if found_inf:
            self.scaler.update(True)
            return False, norms
        self.t += 1
        for i, (k, nrm) in enumerate(zip(names, norms)):
            p, g = params[k], grads[k]
            self.steps[k] = self.steps.get(k, 0) + 1        # torch.optim.AdamW keeps one step count per parameter
            coef = inv
            if self.clip > 0:
                c = self.clip / (nrm + 1e-6)
                if c < 1:
                    coef *= c
            m, v = self.state[k]
            if multi:
                # bias corrections in float32 as bfm_adamw_step forms them (powf, sqrtf)
                b1 = np.float32(1.0) - np.power(np.float32(self.betas[0]), np.float32(self.steps[k]), dtype=np.float32)
                b2 = np.sqrt(np.float32(1.0) - np.power(np.float32(self.betas[1]), np.float32(self.steps[k]), dtype=np.float32),
                             dtype=np.float32)
                desc[i]["grad_scale"], desc[i]["bias1"], desc[i]["bias2_sqrt"] = coef, b1, b2
            else:
                L.check(lib.bfm_adamw_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, self.betas[0], self.betas[1],
                                           self.eps, wd, self.steps[k], coef, st), "adamw " + k)
        if multi:
            tdev = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(self.dev)
            L.check(lib.bfm_adamw_step_multi(L.ptr(tdev), len(names), L.ptr(self._chunk_map), int(self._chunk_map.numel()), CH, lr,
                                             self.betas[0], self.betas[1], self.eps, wd, st), "adamw_multi")
            self._keep_desc = tdev
        self._weights_changed()
        self.scaler.update(False)
        return True, norms

    @L.on_device(lambda self, *a, **k: self.dev)
    def step(self, xs, target, samples, lr=None, weight_decay=None, group=None):
        """One full iteration (Trainer/engine.py:96-147).  Returns (loss_dict, total, stepped).  With more than one rank
        the loss dictionary is averaged over the ranks first (utils.reduce_dict, engine.py:124-130) and the skip decision
        is taken on that reduced value, so every rank skips -- or enters the gradient all-reduce -- together."""
        import torch.distributed as dist
        self._group = group
        self._store_next = True
        loss_dict, total, grads = self.loss_and_grads(xs, target, samples)
        store = self.__dict__.pop("_store_live", None)
        touched = set(self._touched_heads)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if store is not None:
            store.finish()                                     # every rank, whatever its loss: the collectives are in flight
            self.allreduce_events = list(store.events) if store.events else []
        if multi:
            world = dist.get_world_size(group)
            tasks = list(self.tail.row_of.keys())
            keys = list(self.loss_names)                       # fixed layout: a rank may lack a loss another one has
            # gloo reduces on the host, RCCL on the device
            rdev = self.dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.tensor([loss_dict.get("loss_" + k, 0.0) for k in keys] +
                             [1.0 if ("loss_" + k) in loss_dict else 0.0 for k in keys] +
                             [1.0 if task in touched else 0.0 for task in tasks], dtype=torch.float64, device=rdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            t = t.cpu().tolist()
            nk = len(keys)
            loss_dict = OrderedDict(("loss_" + k, t[i] / world) for i, k in enumerate(keys) if t[nk + i] > 0)
            total = sum(v * self.loss_weights.get(k, 0.0) for k, v in loss_dict.items() if k in self.loss_weights)
            touched = {task for j, task in enumerate(tasks) if t[2 * nk + j] > 0}     # find_unused_parameters=True
        if not math.isfinite(total):
            return loss_dict, total, False                     # engine.py:129-136: non-finite loss -> skip the iteration
        grad_div = 1.0
        if store is not None:
            grad_div = float(store.world)                      # the buckets hold the SUM over the ranks; the mean's 1 / world
        else:                                                  # goes into the optimiser's gradient scale
            allreduce_mean_(grads, group, self.__dict__.setdefault("allreduce_events", []))
        for task in self.tail.row_of:                          # heads nobody's loss reached: no gradient, no step
            if task not in touched:
                grads.pop("head.final_conv_%s.weight" % task, None)
                grads.pop("head.final_conv_%s.bias" % task, None)
        stepped, _ = self.apply(grads, lr, weight_decay, grad_div=grad_div)
        return loss_dict, total, stepped
