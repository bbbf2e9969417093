"""Pins oracle/train_ref.py (the restated losses, clipping and AdamW) against tests/golden/train_step.npz, which was made
by running the reference's model, processors, SetMultiCriterion, clip_gradients and torch.optim.AdamW in float64
(tests/golden/make_golden_train.py).  CPU only."""
import numpy as np
import torch

from conftest import load_npz
from oracle import train_ref as T
from oracle import unet_ref as O


def load_case():
    d = load_npz("train_step.npz")
    f_maps, levels, groups = (int(v) for v in d["cfg"])
    names = [str(s) for s in d["param_names"]]
    hyper = d["hyper"]
    case = dict(
        d=d, f_maps=f_maps, levels=levels, groups=groups, names=names,
        lr=float(hyper[0]), wd=float(hyper[1]), clip=float(hyper[2]), b1=float(hyper[3]), b2=float(hyper[4]),
        eps=float(hyper[5]), all_samples=float(hyper[6]), max_dist=float(hyper[7]),
        loss_names=[str(s) for s in d["loss_names"]],
        loss_weights={str(k): float(v) for k, v in zip(d["loss_weight_names"], d["loss_weights"])},
        bias_l2=str(d["bias_field_log_type"]) == "l2",
        out_channels=O.default_out_channels(left_hemis_only=True),
    )
    case["n_samples"] = sum(1 for k in d if k.startswith("x") and k[1:].isdigit())
    return case


def test_oracle_iteration_matches_reference_fp64():
    c = load_case()
    d = c["d"]
    params = {k[3:]: torch.from_numpy(v).double().requires_grad_(True) for k, v in d.items() if k.startswith("sd/")}
    target = {k[7:]: torch.from_numpy(v).double() for k, v in d.items() if k.startswith("target/")}
    xs, samples = [], []
    for i in range(c["n_samples"]):
        xs.append(torch.from_numpy(d["x%d" % i]).double())
        samples.append({"bias_field_log": torch.from_numpy(d["bias_field_log%d" % i]).double(),
                        "high_res_residual": torch.from_numpy(d["high_res_residual%d" % i]).double()})
    wce = torch.from_numpy(d["weights_ce"]).double()
    total, ld = T.iteration_loss(xs, params, target, samples, c["out_channels"], c["loss_names"], c["loss_weights"], wce,
                                 c["all_samples"], c["f_maps"], c["levels"], c["groups"], c["max_dist"], c["bias_l2"])
    for k, v in ld.items():
        ref = float(d["loss/" + k])
        assert abs(float(v.detach()) - ref) <= 1e-9 * max(1.0, abs(ref)), (k, float(v.detach()), ref)
    assert abs(float(total.detach()) - float(d["loss_total"])) <= 1e-9 * abs(float(d["loss_total"]))
    total.backward()
    grads = {}
    for k in c["names"]:
        ref = d["grad/" + k]
        got = params[k].grad.numpy()
        assert np.abs(got - ref).max() <= 1e-9 * max(1e-12, np.abs(ref).max()) + 1e-14, k
        grads[k] = params[k].grad
    clipped, norms = T.clip_gradients(grads, c["clip"])
    assert np.allclose(norms, d["clip_norms"], rtol=1e-9, atol=1e-12)          # clip_gradients rounds nothing
    for k in c["names"]:
        assert np.abs(clipped[k].numpy() - d["clipped/" + k]).max() <= 1e-9 * max(1e-12, np.abs(d["clipped/" + k]).max()) + 1e-14
        p0 = params[k].detach()
        p1, _, _ = T.adamw_step(p0, clipped[k], torch.zeros_like(p0), torch.zeros_like(p0), 1, c["lr"], c["b1"], c["b2"],
                                c["eps"], c["wd"])
        assert np.abs(p1.numpy() - d["after/" + k]).max() <= 1e-12, k


def test_forward_difference_conventions():
    """GradientLoss.gradient: x is the fastest axis ('back'), zero on the last slice."""
    x = torch.arange(2 * 3 * 4, dtype=torch.float64).reshape(1, 1, 2, 3, 4) ** 2
    dx, dy, dz = T.forward_diffs(x)
    assert torch.all(dx[..., -1] == 0) and torch.all(dy[..., -1, :] == 0) and torch.all(dz[..., -1, :, :] == 0)
    assert float(dx[0, 0, 0, 0, 0]) == 1.0 and float(dy[0, 0, 0, 0, 0]) == 16.0 and float(dz[0, 0, 0, 0, 0]) == 144.0
