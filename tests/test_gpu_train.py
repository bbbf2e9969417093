"""one training iteration on the HIP kernels against the golden vectors made by running the reference
(model -> processors -> SetMultiCriterion -> backward -> clip_gradients -> AdamW, float64) and against the oracle.
Needs an MI355X: run with `-m gpu`."""
import numpy as np
import pytest
import torch

from conftest import sd_from_npz
from oracle import train_ref as T
from test_oracle_train import load_case

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return torch.device("cuda:0")


def _build(c, scaler=None, clip=None):
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    d = c["d"]
    ga, ta = TU.default_inference_args(f_maps=c["f_maps"], num_levels=c["levels"], left_hemis_only=True,
                                       num_groups=c["groups"])
    s = TU.InferenceSession(ga, ta, _dev(), state_dict=sd_from_npz(d), passes=3)
    step = TR.TrainStep(s.engine, s.model.head.tail(s.engine), c["loss_names"], c["loss_weights"], d["weights_ce"], c["all_samples"],
                        max_surf_distance=c["max_dist"], bias_field_log_type="l2" if c["bias_l2"] else "l1",
                        lr=c["lr"], weight_decay=c["wd"], betas=(c["b1"], c["b2"]), eps=c["eps"],
                        clip_max_norm=c["clip"] if clip is None else clip, scaler=scaler)
    target = {k[7:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("target/")}
    xs = [torch.from_numpy(d["x%d" % i]) for i in range(c["n_samples"])]
    samples = [{"bias_field_log": torch.from_numpy(d["bias_field_log%d" % i]),
                "high_res_residual": torch.from_numpy(d["high_res_residual%d" % i])} for i in range(c["n_samples"])]
    return step, xs, target, samples


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def test_training_iteration_vs_reference_golden():
    """Loss dictionary, every parameter gradient, the per-parameter clipping norms and the parameters after one AdamW
    step, for two augmented samples of a 3-level net with the full demo head set (clamped distance head included)."""
    c = load_case()
    d = c["d"]
    step, xs, target, samples = _build(c)
    loss_dict, total, grads = step.loss_and_grads(xs, target, samples)
    assert list(loss_dict.keys()) == ["loss_" + n for n in c["loss_names"]]
    for k, v in loss_dict.items():
        ref = float(d["loss/" + k])
        assert abs(v - ref) <= 1e-4 * max(abs(ref), 1e-3), (k, v, ref)            # fp32 forward, fp64 reduction
    assert abs(total - float(d["loss_total"])) <= 1e-4 * float(d["loss_total"])
    assert set(grads.keys()) == set(c["names"])
    worst = {k: _rel(grads[k].reshape(d["grad/" + k].shape).cpu().numpy(), d["grad/" + k]) for k in c["names"]}
    print("max rel grad err vs reference fp64: %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))
    # |.|-type losses have sign() gradients: a voxel whose residual rounds across zero in fp32 flips a whole 1/N
    # contribution, so the tolerance is looser than for the smooth backbone test (5e-4)
    bad = {k: v for k, v in worst.items() if v > 2e-3}
    assert not bad, bad
    before = {k: v.clone() for k, v in step.parameters().items()}
    mine = {k: grads[k].double().cpu() for k in c["names"]}
    stepped, norms = step.apply(grads)
    assert stepped and step.t == 1
    assert np.allclose(norms, d["clip_norms"], rtol=2e-3, atol=1e-7)
    after = step.parameters()
    clipped, _ = T.clip_gradients(mine, c["clip"])
    for k in c["names"]:
        ref_delta = d["after/" + k] - d["sd/" + k].astype(np.float64)
        got_delta = (after[k].double() - before[k].double()).reshape(ref_delta.shape).cpu().numpy()
        # step 1 of Adam moves every weight by lr * g/(|g| + eps): ill-conditioned where |g| ~ eps = 1e-8, so the
        # reference's move is compared where its clipped gradient is well above eps ...
        well = np.abs(d["clipped/" + k]) > 1e-6
        assert np.abs(got_delta - ref_delta)[well].max(initial=0.0) <= 2e-2 * c["lr"], k
        # ... and everywhere against the oracle's AdamW applied to THIS path's gradient (clip coefficient, lr, weight
        # decay and bias corrections wired as in the reference), to fp32 rounding of the parameter
        p0 = before[k].double().cpu()
        p1, _, _ = T.adamw_step(p0, clipped[k].reshape(p0.shape), torch.zeros_like(p0), torch.zeros_like(p0), 1, c["lr"],
                                c["b1"], c["b2"], c["eps"], c["wd"])
        assert float((after[k].double().cpu() - p1).abs().max()) <= 2e-3 * c["lr"] + 2e-7 * float(p0.abs().max()), k
    # the next forward must see the new weights (packed-weight caches dropped)
    l2, total2, _ = step.loss_and_grads(xs, target, samples)
    assert total2 != total


def test_losses_match_oracle_on_random_heads():
    """The loss kernels alone: values and d/d(raw) against torch autograd of the restated criterion in float64, on
    random head outputs (no network), with weights, the bias-field mask, the distance clamp and an l1 bias loss."""
    from brainfm_amd import _lib as L
    c = load_case()
    d = c["d"]
    step, xs, target, samples = _build(c)
    step.bias_l2 = 0
    dims = tuple(xs[0].shape[-3:])
    nvox = dims[0] * dims[1] * dims[2]
    tail = step.tail
    g = torch.Generator().manual_seed(5)
    raw = torch.randn((nvox, tail.n_out), generator=g) * 1.5
    raw_d = raw.to(_dev())
    dRaw = torch.zeros_like(raw_d)
    vals = torch.zeros(4 * len(step.loss_names) + 2 * tail.n_out + 8, dtype=torch.float64, device=_dev())
    slots, _ = step._sample_losses(raw_d, dims, target, samples[0], dRaw, vals, 1.0)
    got = step._finish_losses([(slots, vals)], nvox)
    # oracle: same raw values as NCDHW head outputs, float64 autograd
    r64 = raw.double().requires_grad_(True)
    out = {}
    for task, (r0, n) in tail.row_of.items():
        out[task] = r64[:, r0:r0 + n].t().reshape((1, n) + dims)
    out = T.processors(out, c["max_dist"])
    t64 = {k: v.double() for k, v in target.items()}
    s64 = {k: v.double() for k, v in samples[0].items()}
    wce = torch.from_numpy(d["weights_ce"]).double()
    ld = T.multi_criterion([out], t64, [s64], c["loss_names"], wce, c["all_samples"], bias_l2=False)
    tot = sum(ld[k] * c["loss_weights"][k] for k in ld)
    tot.backward()
    for k, v in ld.items():
        v = float(v.detach())
        assert abs(got[k] - v) <= 2e-6 * max(abs(v), 1e-3), (k, got[k], v)
    ref = r64.grad.numpy()
    err = np.abs(dRaw.cpu().numpy().astype(np.float64) - ref)
    # sign() gradients can flip only where the fp32 residual is exactly at a rounding boundary: allow a handful
    tol = 1e-5 * np.abs(ref).max()
    assert (err > tol).mean() < 1e-4, ((err > tol).sum(), err.max(), np.abs(ref).max())


def test_adamw_kernel_matches_torch_optim():
    """bfm_adamw_step over three steps against torch.optim.AdamW (the reference's optimiser, Trainer/models/__init__.py
    :362-366) on the same device."""
    from brainfm_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(10007, generator=g)
    p_ref = torch.nn.Parameter(p0.clone().to(_dev()))
    opt = torch.optim.AdamW([p_ref], lr=3e-3, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8)
    p = p0.clone().to(_dev())
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for t in range(1, 4):
        gr = torch.randn(10007, generator=g).to(_dev()) * (10.0 ** (t - 3))
        p_ref.grad = gr.clone()
        opt.step()
        L.check(lib.bfm_adamw_step(L.ptr(p), L.ptr(gr), L.ptr(m), L.ptr(v), p.numel(), 3e-3, 0.9, 0.999, 1e-8, 0.1, t, 1.0,
                                   L.stream_ptr()), "adamw")
        assert float((p - p_ref.detach()).abs().max()) <= 2e-6, t


def test_loss_scaler_skips_nonfinite_and_unscales():
    """GradScaler semantics: scaled backward gives the same update as the unscaled one (power-of-two scale), an inf
    gradient skips the step and halves the scale."""
    from brainfm_amd import train as TR
    c = load_case()
    a, xs, target, samples = _build(c)
    b, _, _, _ = _build(c, scaler=TR.LossScaler(init_scale=1024.0, growth_interval=1))
    la, ta, ga = a.loss_and_grads(xs, target, samples)
    lb, tb, gb = b.loss_and_grads(xs, target, samples)
    assert ta == tb
    k = "backbone.decoders.1.basic_module.SingleConv2.conv.weight"
    assert _rel((gb[k] / 1024.0).cpu().numpy(), ga[k].cpu().numpy()) <= 1e-5
    a.apply(ga)
    ok, _ = b.apply(gb)
    assert ok and b.scaler.scale == 2048.0                       # growth_interval = 1 clean step
    pa, pb = a.parameters(), b.parameters()
    for key in pa:
        assert float((pa[key] - pb[key]).abs().max()) <= 2e-2 * c["lr"], key
    _, _, gb = b.loss_and_grads(xs, target, samples)
    gb[k][0, 0, 0, 0, 0] = float("inf")
    before = b.parameters()[k].clone()
    ok, _ = b.apply(gb)
    assert not ok and b.scaler.scale == 1024.0 and b.t == 1
    assert torch.equal(before, b.parameters()[k])


def test_generator_feeds_training_steps_and_loss_decreases():
    """BASELINE config 4 at one GPU and toy size: an item of the on-device generator (ShapeID pathology + deformation +
    interpol warp + augmentation chain, tests/test_gpu_synth.py) goes through the reference's collate convention
    (batch dimension added) into TrainStep.step; five AdamW steps on that one item lower the weighted total."""
    import test_gpu_synth as SY
    from brainfm_amd import generator as G
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    from oracle import unet_ref as O
    rs = np.random.RandomState(0)
    shp = (48, 44, 52)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
    ell = (((zz - 24) / 20.) ** 2 + ((yy - 22) / 18.) ** 2 + ((xx - 26) / 22.) ** 2) <= 1
    seeds = rs.rand(30, 3) * np.array(shp)
    lab = np.argmin(((np.stack([zz, yy, xx], -1)[..., None, :] - seeds) ** 2).sum(-1), -1)
    ids = np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13])[lab % 10] * ell
    case = {"name": "toy", "Gen": ids.astype(np.float32), "T1": rs.rand(*shp).astype(np.float32) * ell,
            "segmentation": ids.astype(np.int32),
            "distance": [rs.rand(*shp).astype(np.float32) * 255 for _ in range(4)],
            "registration": [rs.randn(*shp).astype(np.float32) * 500 for _ in range(3)]}
    np.random.seed(3)
    torch.manual_seed(3)
    ga = SY._gen_args()
    ga.task.pathology = False
    ds = G.build_datasets(ga, "cuda:0", cases=[case])["all"]
    _, _, _, target, samples = ds[0]
    target = {k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in target.items()}      # collate: batch dim
    samples = [{k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in s.items()} for s in samples]
    tasks = dict(T1=True, T2=False, FLAIR=False, CT=False, segmentation=True, distance=True, bias_field=True,
                 registration=True, super_resolution=True, surface=False, pathology=False, contrastive=False)
    gi, ti = TU.default_inference_args(f_maps=8, num_levels=3, tasks=tasks, size=(32, 32, 32))
    oc = O.default_out_channels(tasks=[k if k != "bias_field" else "bias_field" for k, v in tasks.items() if v])
    sd = O.random_state_dict(1, 8, 3, out_channels=oc, seed=9)
    s = TU.InferenceSession(gi, ti, _dev(), state_dict=sd, passes=3)
    tail = s.model.head.tail(s.engine)
    names = ["T1", "T1_grad", "seg_ce", "seg_dice", "distance", "bias_field_log", "registration", "registration_grad",
             "SR", "SR_grad"]
    nseg = tail.desc.n_seg
    step = TR.TrainStep(s.engine, tail, names, {"loss_" + n: 1.0 for n in names}, torch.full((nseg,), 1.0 / nseg),
                        all_samples=len(samples), lr=2e-3, scaler=TR.LossScaler(init_scale=256.0))
    totals = []
    for it in range(5):
        ld, tot, ok = step.step([x["input"] for x in samples], target, samples)
        assert ok and np.isfinite(tot) and set(ld) == {"loss_" + n for n in names}
        totals.append(tot)
    print("totals", [round(t, 4) for t in totals])
    assert totals[-1] < totals[0]


def test_two_ranks_training_step_equals_averaged_gradients():
    """DDP semantics at world_size 2 on the real kernels: two gloo ranks share this GPU, each trains on its own sample,
    gradients are averaged by the flat all-reduce; every rank ends with the same parameters, equal to a single-process
    step on the hand-averaged gradients (child processes: see tests/two_rank_train_worker.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "two_rank_train_worker.py")], capture_output=True, text=True,
                       timeout=600)
    if not r.stdout.strip().endswith("OK"):
        err = [ln for ln in r.stderr.splitlines() if "socket.cpp" not in ln]
        raise AssertionError("two-rank training run failed:\n%s\n%s" % (r.stdout[-1500:], "\n".join(err[-40:])))


def test_checkpoint_round_trip_resumes_bit_identically(tmp_path):
    """save_checkpoint writes the reference's {'model', 'optimizer', 'epoch'} file (reference parameter names and shapes,
    torch-AdamW state layout); a fresh session that loads it continues exactly like the one that kept running, and
    torch.optim.AdamW accepts the optimizer part for a module with the same parameter list."""
    c = load_case()
    a, xs, target, samples = _build(c)
    a.step(xs, target, samples)
    path = str(tmp_path / "brainfm_pretrained.pth")
    a.save_checkpoint(path, epoch=3)
    ckp = torch.load(path, map_location="cpu", weights_only=False)
    assert ckp["epoch"] == 3 and set(ckp["model"].keys()) == set(c["names"])
    assert tuple(ckp["model"]["head.final_conv_T1.weight"].shape) == (1, c["f_maps"], 1, 1, 1)
    # torch's own optimiser takes the state (same parameter order and shapes)
    ps = [torch.nn.Parameter(v.clone()) for v in ckp["model"].values()]
    opt = torch.optim.AdamW(ps)
    opt.load_state_dict(ckp["optimizer"])
    assert float(opt.state[ps[0]]["step"]) == 1.0
    b, _, _, _ = _build(c)
    b.load_checkpoint(path)
    assert b.t == 1
    la, ta, _ = a.step(xs, target, samples)
    lb, tb, _ = b.step(xs, target, samples)
    assert ta == tb
    pa, pb = a.parameters(), b.parameters()
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    # the inference loader reads the same file
    from brainfm_amd import models as M
    from brainfm_amd import test_utils as TU
    ga, ta_ = TU.default_inference_args(f_maps=c["f_maps"], num_levels=c["levels"], left_hemis_only=True,
                                        num_groups=c["groups"])
    s = TU.InferenceSession(ga, ta_, _dev(), ckp_path=path, passes=3)
    assert torch.equal(s.engine.enc[0][0].w_raw.cpu(), ckp["model"][s.engine.enc[0][0].name + ".conv.weight"])


def test_sample_lanes_give_the_same_bits_as_serial_samples():
    """From the second iteration on the augmented samples of an iteration run on two streams (packed weights and tuned
    variants are in place by then); gradients are still added in sample order on the caller's stream, so losses,
    gradients and parameters after three iterations equal the strictly serial run bit for bit."""
    c = load_case()
    a, xs, target, samples = _build(c)
    b, _, _, _ = _build(c)
    assert a.sample_lanes == 2
    b.sample_lanes = 1
    xs3, samples3 = xs + [xs[0] * 0.5 + 0.1], samples + [samples[1]]            # three samples: lanes 0, 1, 0
    for it in range(3):
        la, ta, ga = a.loss_and_grads(xs3, target, samples3)
        lb, tb, gb = b.loss_and_grads(xs3, target, samples3)
        assert ta == tb and la == lb, it
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (it, k)
        a.apply(ga)
        b.apply(gb)
    pa, pb = a.parameters(), b.parameters()
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


def test_graph_inference_after_training_sees_the_new_weights():
    """Captured tile graphs bake in the packing exponents; after an optimiser step the session drops them, so graph
    replay equals eager inference on the updated weights."""
    from brainfm_amd import test_utils as TU
    c = load_case()
    from conftest import sd_from_npz
    ga, ta = TU.default_inference_args(f_maps=c["f_maps"], num_levels=c["levels"], left_hemis_only=True, num_groups=c["groups"])
    s = TU.InferenceSession(ga, ta, _dev(), state_dict=sd_from_npz(c["d"]), passes=3)
    from brainfm_amd import train as TR
    step = TR.TrainStep(s.engine, s.model.head.tail(s.engine), c["loss_names"], c["loss_weights"], c["d"]["weights_ce"],
                        c["all_samples"], lr=0.05)
    g = torch.Generator().manual_seed(1)
    vol = torch.rand((1, 1, 24, 20, 28), generator=g).to(_dev())
    TU.prepare_tile_graphs(vol, s, [8] * 3, [16] * 3)
    before, _, _ = TU.tiled_inference(vol, s, [8] * 3, [16] * 3, graphs=True)
    before = {k: v.clone() for k, v in before.items()}
    d = c["d"]
    target = {k[7:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("target/")}
    xs = [torch.from_numpy(d["x0"])]
    samples = [{"bias_field_log": torch.from_numpy(d["bias_field_log0"]), "high_res_residual": torch.from_numpy(d["high_res_residual0"])}]
    step.step(xs, target, samples)
    eager, _, _ = TU.tiled_inference(vol, s, [8] * 3, [16] * 3, graphs=False)
    eager = {k: v.clone() for k, v in eager.items()}
    for rep in range(3):
        got, _, _ = TU.tiled_inference(vol, s, [8] * 3, [16] * 3, graphs=True)
        for k in eager:
            assert torch.equal(got[k], eager[k]), (k, rep)
    assert any(not torch.equal(before[k], eager[k]) for k in eager)


def test_tile_loop_heads_follow_weights_that_grew_in_training():
    """The tile loop's head pass (runs of 64 zero-input voxels skipped) scales the head weights into fp16 range by the
    descriptor's max|w|.  Round 2 kept a private copy of the descriptor for that pass, made before training and never
    refreshed: head weights that grow a hundredfold then overflow fp16 in the tile loop only.  One
    descriptor now serves both; the tile loop on a one-tile volume must give forward_fused's maps."""
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    from conftest import sd_from_npz
    c = load_case()
    ga, ta = TU.default_inference_args(f_maps=c["f_maps"], num_levels=c["levels"], left_hemis_only=True, num_groups=c["groups"])
    s = TU.InferenceSession(ga, ta, _dev(), state_dict=sd_from_npz(c["d"]), passes=3)
    step = TR.TrainStep(s.engine, s.model.head.tail(s.engine), c["loss_names"], c["loss_weights"], c["d"]["weights_ce"],
                        c["all_samples"], lr=0.05)
    g = torch.Generator().manual_seed(2)
    vol = (torch.rand((1, 1, 16, 16, 64), generator=g) + 0.05).to(_dev())
    vol[:, :, :, :, 40:] = 0                                            # runs of zero input: the skipping form is in use
    TU.tiled_inference(vol, s, [16, 16, 64], [16, 16, 64], graphs=False)     # first tile pass, before the weights move
    step.tail.head_w.mul_(100.0)
    step._weights_changed()
    tiled, ranges, _ = TU.tiled_inference(vol, s, [16, 16, 64], [16, 16, 64], graphs=False)
    assert len(ranges) == 1
    fused, _ = s.forward_fused(vol)
    m = (vol[0, 0] != 0)
    for k, v in tiled.items():
        if k not in fused or k == "label":
            continue
        want = fused[k].reshape(m.shape) * m
        assert bool(torch.isfinite(v).all()), k
        err = float((v - want).abs().max()) / max(1e-6, float(want.abs().max()))
        assert err <= 1e-4, (k, err)


@pytest.mark.gpu
@pytest.mark.parametrize("n_out,nvox", [(69, 128 * 77 + 5), (96, 4096), (8, 1000), (33, 64)])
def test_heads_backward_fused_pass_vs_float64(n_out, nvox):
    """bfm_head_bwd at the shipped feature width (C = 64) takes the one-pass kernel of round 4 -- dFn = dRaw . W, dW = dRaw^T . Fn
    and db = column sums of dRaw from LDS tiles on the exact-fp32 matrix core -- instead of three passes over dRaw; both
    paths against float64 (fp32 accumulation over nvox terms: 2e-5 of the largest entry), incl. voxel counts that are not a
    multiple of the 64-voxel tile and head widths on both sides of the 32-row blocks.  C = 16 keeps the separate kernels."""
    import ctypes as C
    from brainfm_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    for cf in (64, 16):
        g = torch.Generator().manual_seed(n_out * 7 + cf)
        dRaw = (torch.randn(nvox, n_out, generator=g) * (torch.rand(nvox, 1, generator=g) > 0.3)).to(dev)
        Fn = torch.randn(nvox, cf, generator=g).to(dev)
        W = torch.randn(n_out, cf, generator=g).to(dev)
        dW = torch.full((n_out, cf), float("nan"), device=dev)
        db = torch.full((n_out,), float("nan"), device=dev)
        dFn = torch.full((nvox, cf), float("nan"), device=dev)
        ws = torch.empty(lib.bfm_head_bwd_workspace(n_out, cf, nvox), dtype=torch.uint8, device=dev)
        L.check(lib.bfm_head_bwd(L.ptr(dRaw), L.ptr(Fn), L.ptr(W), n_out, cf, nvox, L.ptr(dW), L.ptr(db), L.ptr(dFn),
                                 L.ptr(ws), ws.numel(), L.stream_ptr()), "head_bwd")
        r64, f64, w64 = dRaw.double(), Fn.double(), W.double()
        for got, ref, what in ((dFn, r64 @ w64, "dFn"), (dW, r64.t() @ f64, "dW"), (db, r64.sum(0), "db")):
            err = float((got.double() - ref).abs().max()) / max(1e-6, float(ref.abs().max()))
            assert err <= 2e-5, (cf, what, err)
        # the rows layout of dRaw ([n_out] rows, pitch > nvox here) feeds the same tiles: same bits
        pitch = nvox + 24
        rows = torch.full((n_out, pitch), float("nan"), device=dev)
        rows[:, :nvox] = dRaw.t()
        dW2, db2, dFn2 = torch.empty_like(dW), torch.empty_like(db), torch.empty_like(dFn)
        rc = lib.bfm_head_bwd_rows(L.ptr(rows), pitch, L.ptr(Fn), L.ptr(W), n_out, cf, nvox, L.ptr(dW2), L.ptr(db2),
                                   L.ptr(dFn2), L.ptr(ws), ws.numel(), L.stream_ptr())
        if cf == 64:
            L.check(rc, "head_bwd_rows")
            assert torch.equal(dW2, dW) and torch.equal(db2, db) and torch.equal(dFn2, dFn)
        else:
            assert rc == -2                            # rows exist for the one-pass kernel only


def test_rows_layout_of_the_head_outputs_equals_channels_last():
    """Round 4: the training step keeps the head outputs as [n_out] rows of nvox values (bfm_tail_raw_rows, bfm_loss_*_rows,
    bfm_head_bwd_rows).  Per voxel the rows kernels evaluate the channels-last kernels' expressions: logits, probabilities
    and d/d(raw) must be the SAME BITS (transposed), the fp64 loss sums agree to rounding of a different fold order."""
    import ctypes as C
    from brainfm_amd import _lib as L
    c = load_case()
    step, xs, target, samples = _build(c)
    dims = tuple(xs[0].shape[-3:])
    nvox = dims[0] * dims[1] * dims[2]
    tail = step.tail
    dev = _dev()
    # (1) TaskHead.forward in both layouts
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(dims + (tail.c_feat,), generator=g).to(dev)
    raw_cl, fn_cl = tail.run_raw(feat, dims, want_feat=True)
    raw_rows, fn_rows = tail.run_raw(feat, dims, want_feat=True, rows=True)
    assert raw_rows.shape == (tail.n_out, nvox)
    assert torch.equal(raw_rows.t().reshape(raw_cl.shape), raw_cl)
    assert (fn_cl is None and fn_rows is None) or torch.equal(fn_cl, fn_rows)
    # (2) every loss of a sample on random head outputs
    raw = (torch.randn((nvox, tail.n_out), generator=g) * 1.5).to(dev)
    outs = []
    for rows in (False, True):
        r = raw.t().contiguous() if rows else raw
        dRaw = torch.zeros_like(r)
        vals = torch.zeros(4 * len(step.loss_names) + 2 * tail.n_out + 8, dtype=torch.float64, device=dev)
        slots, _ = step._sample_losses(r, dims, target, samples[0], dRaw, vals, 1.0, rows=rows)
        outs.append((step._finish_losses([(slots, vals)], nvox), dRaw.t() if rows else dRaw))
    (la, da), (lb, db) = outs
    assert list(la) == list(lb)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-12 * max(1.0, abs(la[k])), (k, la[k], lb[k])
    assert torch.equal(da, db), float((da - db).abs().max())


def _split_f16(x):
    """hi = fp16(x) (round to nearest even), lo = fp16(x - hi): the two planes of every packed weight form."""
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def test_weight_repack_kernels_against_layout_models():
    """The kernels a training step re-runs on every layer after AdamW (round 4: LDS-tiled with 16-byte loads): max |w|, the
    transposed + tap-mirrored (+ zero-padded) weights of the data-gradient conv, and the two matrix-core weight forms --
    each against a numpy model of its documented layout, bit for bit."""
    import ctypes as C
    from brainfm_amd import _lib as L
    lib = L.load()
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    st = L.stream_ptr()
    # max |x|: contiguous (16-byte path), a column slice (rows with a pitch), an odd length (scalar path)
    for shape, sl in (((64, 48, 27), None), ((64, 48, 27), 16), ((7, 5, 27), None), ((33, 3, 27), 1)):
        w = torch.randn(shape, generator=g).to(dev)
        t = w if sl is None else w[:, sl:]
        out = torch.zeros(1, device=dev)
        rows, ln, stride = (1, t.numel(), t.numel()) if t.is_contiguous() else (t.shape[0], t[0].numel(), t.stride(0))
        L.check(lib.bfm_absmax_f32(L.ptr(t), rows, ln, stride, L.ptr(out), st), "absmax")
        assert float(out) == float(t.abs().max()), (shape, sl)
    # transposed, mirrored, padded
    for cout, cin, pad in ((64, 32, 64), (96, 48, 64), (32, 16, 16), (8, 16, 64), (64, 24, 24)):
        w = torch.randn(cout, cin, 27, generator=g).to(dev)
        out = torch.full((pad, cout, 27), float("nan"), device=dev)
        L.check(lib.bfm_transpose_mirror_weights(L.ptr(w), cout, cin, pad, L.ptr(out), st), "transpose_mirror")
        ref = torch.zeros(pad, cout, 27, device=dev)
        ref[:cin] = w.permute(1, 0, 2).flip(2)
        assert torch.equal(out, ref), (cout, cin, pad)
    # matrix-core forms
    for cout, cin in ((64, 16), (128, 48)):
        w = (torch.randn(cout, cin, 27, generator=g) * 0.1)
        wmax = float(w.abs().max())
        wd = w.to(dev)
        wexp = C.c_int(0)
        nt, kcn = cout // 64, cin // 16
        # (a) packed[ntile][kc][tap][nb][hl][lane][8]: w[ntile*64 + nb*32 + (l&31)][kc*16 + 8*(l>>5) + j][tap]
        buf = torch.zeros(nt * kcn * 27 * 2 * 2 * 64 * 8, dtype=torch.float16, device=dev)
        L.check(lib.bfm_pack_conv_weights_mfma(L.ptr(wd), cin, cout, wmax, L.ptr(buf), C.byref(wexp), st), "pack_mfma")
        x = (w.numpy() * np.float32(2.0 ** wexp.value)).astype(np.float32)
        hi, lo = _split_f16(x)
        ref = np.zeros((nt, kcn, 27, 2, 2, 64, 8), np.float16)
        l = np.arange(64)
        for n in range(nt):
            for kc in range(kcn):
                for nb in range(2):
                    co = n * 64 + nb * 32 + (l & 31)
                    for j in range(8):
                        ci = kc * 16 + 8 * (l >> 5) + j
                        ref[n, kc, :, nb, 0, :, j] = hi[co, ci, :].T
                        ref[n, kc, :, nb, 1, :, j] = lo[co, ci, :].T
        assert np.array_equal(buf.cpu().numpy().view(np.uint16), ref.reshape(-1).view(np.uint16)), ("mfma", cout, cin)
        # (b) packed16[ntile][kc][pair 14][cb 4][hl][lane][8]: w[ntile*64 + cb*16 + (l&15)][kc*16 + 8*(kg&1) + j][2*pair + (kg>>1)]
        nbytes = lib.bfm_pack_conv_weights_mfma16_bytes(cin, cout)
        assert nbytes == nt * kcn * 14 * 4 * 2 * 64 * 16
        buf = torch.zeros(nbytes // 2, dtype=torch.float16, device=dev)
        L.check(lib.bfm_pack_conv_weights_mfma16(L.ptr(wd), cin, cout, wmax, L.ptr(buf), C.byref(wexp), st), "pack_mfma16")
        x = (w.numpy() * np.float32(2.0 ** wexp.value)).astype(np.float32)
        hi, lo = _split_f16(np.concatenate([x, np.zeros((cout, cin, 1), np.float32)], axis=2))      # tap 27: zeros
        ref = np.zeros((nt, kcn, 14, 4, 2, 64, 8), np.float16)
        kg = l >> 4
        for n in range(nt):
            for kc in range(kcn):
                for pair in range(14):
                    tap = 2 * pair + (kg >> 1)
                    for cb in range(4):
                        co = n * 64 + cb * 16 + (l & 15)
                        for j in range(8):
                            ci = kc * 16 + 8 * (kg & 1) + j
                            ref[n, kc, pair, cb, 0, :, j] = hi[co, ci, tap]
                            ref[n, kc, pair, cb, 1, :, j] = lo[co, ci, tap]
        assert np.array_equal(buf.cpu().numpy().view(np.uint16), ref.reshape(-1).view(np.uint16)), ("mfma16", cout, cin)


@pytest.mark.parametrize("ca,cb,cout,dims", [(32, 64, 64, (8, 16, 32)), (64, 32, 128, (6, 8, 96)), (32, 32, 64, (8, 12, 40)),
                                             (32, 64, 64, (4, 10, 36)), (32, 32, 64, (6, 14, 24))])
def test_weight_gradient_of_a_decoder_join_vs_float64(ca, cb, cout, dims):
    """bfm_conv3x3x3_wgrad_ex on cat(skip, nearest_up2(low)): for an exact 2x join whose low-res rows tile by 4 x 16 the
    upsampled channels take the folded kernel of round 4 (64 products per low-res voxel on the low-res tensor instead of
    27 per high-res voxel, conv_wgrad_up_f16_kernel) and the skip channels their own kernel; low-res sizes that do not
    divide by the 4 x 16 tile -- (8, 12, 40) -> 6 x 20, (4, 10, 36) -> 5 x 18 -- shift their last tile back inside and zero the
    shared voxels; (6, 14, 24) -> 7 x 12 is narrower than a tile and keeps the 27-tap kernel on all channels.  Against the float64 correlation of the GroupNorm-applied,
    zero-padded input with dP: split-fp16 products with fp32 accumulation, 2e-5 of the largest entry."""
    import ctypes as C
    from brainfm_amd import _lib as L
    from brainfm_amd.engine import nearest_index_map
    lib = L.load()
    dev = _dev()
    D, H, W = dims
    lo = (D // 2, H // 2, W // 2)
    g = torch.Generator().manual_seed(ca + cb)
    A = torch.randn(dims + (ca,), generator=g).to(dev)
    B = torch.randn(lo + (cb,), generator=g).to(dev)
    dP = (torch.randn(dims + (cout,), generator=g) * 0.01).to(dev)
    cin = ca + cb
    scale = (torch.rand(cin, generator=g) + 0.5).to(dev)
    shift = (torch.randn(cin, generator=g) * 0.1).to(dev)
    maps = [nearest_index_map(lo[a], dims[a]) for a in range(3)]
    reps = [np.bincount(maps[a], minlength=lo[a]).astype(np.int32) for a in range(3)]
    tens = [torch.from_numpy(m).to(dev) for m in maps + reps]
    up = L.Upsample(lo[0], lo[1], lo[2], *[t.data_ptr() for t in tens])
    ix = [torch.from_numpy(m).long().to(dev) for m in maps]
    X = torch.cat([A, B[ix[0]][:, ix[1]][:, :, ix[2]]], dim=-1) * scale + shift
    bnd, xb = dP.abs().max().reshape(1), X.abs().max().reshape(1)
    ws = torch.empty(lib.bfm_conv3x3x3_wgrad_workspace(cin, cout, D, H, W), dtype=torch.uint8, device=dev)
    dW = torch.full((cout, cin, 27), float("nan"), dtype=torch.float32, device=dev)
    L.check(lib.bfm_conv3x3x3_wgrad_ex(L.ptr(dP), cout, L.ptr(A), ca, L.ptr(B), cb, D, H, W, C.byref(up), L.ptr(scale),
                                       L.ptr(shift), L.ptr(bnd), L.ptr(xb), 1, 3, L.ptr(dW), L.ptr(ws), ws.numel(),
                                       L.stream_ptr()), "wgrad")
    Xp = torch.nn.functional.pad(X.double().permute(3, 0, 1, 2)[None], (1, 1, 1, 1, 1, 1))[0]
    d64 = dP.double().reshape(-1, cout)
    ref = torch.empty((cout, cin, 27), dtype=torch.float64, device=dev)
    for t in range(27):
        kd, kh, kw = t // 9, (t // 3) % 3, t % 3
        ref[:, :, t] = (Xp[:, kd:kd + D, kh:kh + H, kw:kw + W].reshape(cin, -1) @ d64).t()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-5, err


def test_multi_tensor_clip_norms_and_adamw_equal_the_per_tensor_kernels(monkeypatch):
    """TrainStep.apply with all parameters in one launch each (bfm_grad_sumsq_multi, bfm_adamw_step_multi; round 4) against
    the per-tensor kernels (BFM_ADAM_MULTI=0): same clipping norms to fp64 rounding of a different fold order, same
    parameters after two steps to one float32 rounding of the bias corrections."""
    c = load_case()
    outs = []
    for multi in ("1", "0"):
        monkeypatch.setenv("BFM_ADAM_MULTI", multi)
        step, xs, target, samples = _build(c)
        norms_all = []
        for _ in range(2):
            _, _, grads = step.loss_and_grads(xs, target, samples)
            ok, norms = step.apply(grads)
            assert ok
            norms_all.append(norms)
        outs.append((norms_all, {k: v.clone() for k, v in step.parameters().items()}))
    (na, pa), (nb, pb) = outs
    assert np.allclose(na[0], nb[0], rtol=1e-12, atol=0)
    assert np.allclose(na[1], nb[1], rtol=1e-5, atol=1e-12)
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-6 * max(1.0, float(pb[k].abs().max())), k


def test_surface_loss_matches_oracle():
    """criterion.py:175-176 `loss_surface` (round 6: one of the losses outside brain_id.yaml that used to raise): the L1 of
    the 8-channel surface head (Trainer/models/__init__.py:103-106, 235-237) against its target, value and d/d(raw) against
    float64 autograd of the restated criterion, next to the clamped distance loss in the same launch."""
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    from oracle import unet_ref as O
    tasks = dict(T1=True, T2=False, FLAIR=False, CT=False, segmentation=False, distance=True, bias_field=False,
                 registration=False, super_resolution=False, surface=True, pathology=False, contrastive=False)
    gi, ti = TU.default_inference_args(f_maps=8, num_levels=2, tasks=tasks, size=(16, 16, 16))
    oc = O.default_out_channels(tasks=[k for k, v in tasks.items() if v])
    assert oc.get("surface") == 8
    sd = O.random_state_dict(1, 8, 2, out_channels=oc, seed=3)
    s = TU.InferenceSession(gi, ti, _dev(), state_dict=sd, passes=3)
    tail = s.model.head.tail(s.engine)
    names = ["T1", "distance", "surface"]
    weights = {"loss_T1": 1.0, "loss_distance": 0.7, "loss_surface": 1.3}
    step = TR.TrainStep(s.engine, tail, names, weights, torch.ones(1), all_samples=1, max_surf_distance=3.0)
    dims = (16, 16, 16)
    nvox = 16 ** 3
    g = torch.Generator().manual_seed(8)
    target = {"T1": torch.rand((1, 1) + dims, generator=g), "distance": torch.randn((1, oc["distance"]) + dims, generator=g) * 2,
              "surface": torch.randn((1, 8) + dims, generator=g)}
    raw = torch.randn((nvox, tail.n_out), generator=g) * 1.5
    raw_d = raw.to(_dev())
    dRaw = torch.zeros_like(raw_d)
    vals = torch.zeros(4 * len(names) + 2 * tail.n_out + 8, dtype=torch.float64, device=_dev())
    slots, _ = step._sample_losses(raw_d, dims, target, {}, dRaw, vals, 1.0)
    got = step._finish_losses([(slots, vals)], nvox)
    r64 = raw.double().requires_grad_(True)
    out = {task: r64[:, r0:r0 + n].t().reshape((1, n) + dims) for task, (r0, n) in tail.row_of.items()}
    out = T.processors(out, 3.0)
    ld = T.multi_criterion([out], {k: v.double() for k, v in target.items()}, [{}], names, torch.ones(1).double(), 1)
    tot = sum(ld[k] * weights[k] for k in ld)
    tot.backward()
    assert set(got) == set(ld)
    for k, v in ld.items():
        v = float(v.detach())
        assert abs(got[k] - v) <= 2e-6 * max(abs(v), 1e-3), (k, got[k], v)
    ref = r64.grad.numpy()
    err = np.abs(dRaw.cpu().numpy().astype(np.float64) - ref)
    assert (err > 1e-5 * np.abs(ref).max()).mean() < 1e-4
    r0, n = tail.row_of["surface"]
    assert n == 8 and float(np.abs(ref[:, r0:r0 + n]).max()) > 0


@pytest.mark.parametrize("rows", [False, True])
def test_pathology_losses_match_oracle(rows):
    """criterion.py:193-212 `loss_pathol_ce` / `loss_pathol_dice` on the sigmoid of the one-channel pathology head
    (PatholProcessor, joiner.py:79-87; round 6: bfm_loss_pathol): values and d/d(raw) against float64 autograd of the
    restated criterion, for the channels-last and the rows layout of the head outputs, with a target that is mostly zero
    (a lesion) so that the Dice denominator is small but not clamped."""
    from brainfm_amd import test_utils as TU
    from brainfm_amd import train as TR
    from oracle import unet_ref as O
    tasks = dict(T1=True, T2=False, FLAIR=False, CT=False, segmentation=False, distance=False, bias_field=False,
                 registration=False, super_resolution=False, surface=False, pathology=True, contrastive=False)
    gi, ti = TU.default_inference_args(f_maps=8, num_levels=2, tasks=tasks, size=(16, 16, 16))
    oc = O.default_out_channels(tasks=[k for k, v in tasks.items() if v])
    sd = O.random_state_dict(1, 8, 2, out_channels=oc, seed=4)
    s = TU.InferenceSession(gi, ti, _dev(), state_dict=sd, passes=3)
    tail = s.model.head.tail(s.engine)
    names = ["T1", "pathol_ce", "pathol_dice"]
    weights = {"loss_T1": 1.0, "loss_pathol_ce": 0.8, "loss_pathol_dice": 1.7}
    step = TR.TrainStep(s.engine, tail, names, weights, torch.ones(1), all_samples=2)
    dims = (16, 16, 16)
    nvox = 16 ** 3
    g = torch.Generator().manual_seed(9)
    lesion = (torch.rand((1, 1) + dims, generator=g) > 0.9).float()
    target = {"T1": torch.rand((1, 1) + dims, generator=g), "pathology": lesion}
    raw = torch.randn((nvox, tail.n_out), generator=g) * 2.0
    raw[:7, tail.row_of["pathology"][0]] = -20.0                       # p < 1e-5: the clamp of the cross entropy
    lesion.reshape(-1)[:7] = 1.0
    raw_d = (raw.t().contiguous() if rows else raw).to(_dev())
    dRaw = torch.zeros_like(raw_d)
    vals = torch.zeros(4 * len(names) + 2 * tail.n_out + 8, dtype=torch.float64, device=_dev())
    slots, _ = step._sample_losses(raw_d, dims, target, {}, dRaw, vals, 1.0, rows=rows)
    got = step._finish_losses([(slots, vals)], nvox)
    r64 = raw.double().requires_grad_(True)
    out = {task: r64[:, r0:r0 + n].t().reshape((1, n) + dims) for task, (r0, n) in tail.row_of.items()}
    out = T.processors(out, 3.0)
    ld = T.multi_criterion([out], {k: v.double() for k, v in target.items()}, [{}], names, torch.ones(1).double(), 2)
    tot = sum(ld[k] * weights[k] for k in ld)
    tot.backward()
    assert list(got) == ["loss_" + n for n in names]
    for k, v in ld.items():
        v = float(v.detach())
        assert abs(got[k] - v) <= 5e-6 * max(abs(v), 1e-3), (k, got[k], v)
    ref = r64.grad.numpy()
    have = dRaw.cpu().numpy().astype(np.float64)
    if rows:
        have = have.T
    err = np.abs(have - ref)
    assert (err > 2e-5 * np.abs(ref).max()).mean() < 1e-4, (err.max(), np.abs(ref).max())
    c = tail.row_of["pathology"][0]
    assert np.abs(ref[:, c]).max() > 0 and np.abs(have[:7, c]).max() < 1e-6 * np.abs(ref[:, c]).max() + 1e-12
