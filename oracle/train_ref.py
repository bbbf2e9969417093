"""CPU oracle for one training iteration.  TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Restates, on plain torch CPU tensors of whatever dtype the caller passes (the tests use float64):
  * the processors applied before the criterion          Trainer/models/joiner.py:69-77 (softmax), :149-157 (clamp)
  * SetCriterion / SetMultiCriterion                      Trainer/models/criterion.py:111-124, 178-186, 215-294, 330-353
  * l1_loss / GradientLoss                                Trainer/models/losses.py:10-11, 27-74
  * the weighted total and the iteration                  Trainer/engine.py:114-147
  * utils.misc.clip_gradients                             utils/misc.py:1329-1338
  * torch.optim.AdamW (single tensor, amsgrad off)        as called by Trainer/models/__init__.py:362-366
Pinned against tests/golden/train_step.npz (made by running the reference: tests/golden/make_golden_train.py) in
tests/test_oracle_train.py.
"""
import torch

from oracle import unet_ref as O

IMAGE_KEYS = ("T1", "T2", "FLAIR", "CT")


def l1(o, t, w=1.0):
    return torch.mean((o - t).abs() * w)


def forward_diffs(x):
    """losses.py:39-49: forward differences along the last three axes, zero on the last slice of each."""
    dx = torch.zeros_like(x)
    dy = torch.zeros_like(x)
    dz = torch.zeros_like(x)
    dx[..., :-1] = x[..., 1:] - x[..., :-1]
    dy[..., :-1, :] = x[..., 1:, :] - x[..., :-1, :]
    dz[..., :-1, :, :] = x[..., 1:, :, :] - x[..., :-1, :, :]
    return dx, dy, dz


def grad_l1(o, t, w=1.0):
    a, b = forward_diffs(o), forward_diffs(t)
    return l1(a[0], b[0], w) + l1(a[1], b[1], w) + l1(a[2], b[2], w)


def processors(out, max_surf_distance=3.0):
    out = dict(out)
    if "segmentation" in out:
        out["segmentation"] = torch.softmax(out["segmentation"], dim=1)
    if "distance" in out:
        out["distance"] = torch.clamp(out["distance"], -max_surf_distance, max_surf_distance)
    if "pathology" in out:                                        # PatholProcessor, Trainer/models/joiner.py:79-87
        out["pathology"] = torch.sigmoid(out["pathology"])
    return out


def sample_losses(out, target, sample, loss_names, weights_ce, bias_l2=True):
    """One sample's {loss_<name>: value} (SetCriterion.forward)."""
    res = {}
    wce = weights_ce.reshape(1, -1, 1, 1, 1)
    wdice = weights_ce.reshape(1, -1)
    for name in loss_names:
        if name in IMAGE_KEYS or (name.endswith("_grad") and name[:-5] in IMAGE_KEYS):
            key = name[:-5] if name.endswith("_grad") else name
            w = (1.0 - target[key + "_DM"]) if (key + "_DM") in target else 1.0
            v = grad_l1(out[key], target[key], w) if name.endswith("_grad") else l1(out[key], target[key], w)
        elif name == "SR":
            v = l1(out["high_res_residual"], sample["high_res_residual"])
        elif name == "SR_grad":
            v = grad_l1(out["high_res_residual"], sample["high_res_residual"])
        elif name in ("distance", "registration", "surface"):
            v = l1(out[name], target[name])
        elif name == "registration_grad":
            v = grad_l1(out["registration"], target["registration"])
        elif name == "bias_field_log":
            m = 1.0 - target["segmentation"][:, 0]
            a, b = out["bias_field_log"] * m, sample["bias_field_log"] * m
            v = torch.mean((a - b) ** 2) if bias_l2 else torch.mean((a - b).abs())
        elif name == "pathol_ce":                                 # criterion.py:193-201
            p, t = out["pathology"], target["pathology"]
            v = torch.mean(-torch.sum(torch.log(torch.clamp(p, min=1e-5)) * t, dim=1))
        elif name == "pathol_dice":                               # criterion.py:203-212
            p, t = out["pathology"], target["pathology"]
            v = torch.sum(1.0 - 2.0 * (p * t).sum(dim=[2, 3, 4]) / torch.clamp((p + t).sum(dim=[2, 3, 4]), min=1e-5))
        elif name == "seg_ce":
            p, t = out["segmentation"], target["segmentation"]
            v = torch.mean(-torch.sum(torch.log(torch.clamp(p, min=1e-5)) * wce * t, dim=1))
        elif name == "seg_dice":
            p, t = out["segmentation"], target["segmentation"]
            v = torch.sum(wdice * (1.0 - 2.0 * (p * t).sum(dim=[2, 3, 4]) / torch.clamp((p + t).sum(dim=[2, 3, 4]), min=1e-5)))
        else:
            raise KeyError("loss '%s' is outside the restated set" % name)
        res["loss_" + name] = v
    return res


def multi_criterion(outs, target, samples, loss_names, weights_ce, all_samples, bias_l2=True):
    """SetMultiCriterion.forward: per-loss sum over the samples, divided by gen_args.generator.all_samples."""
    tot = {}
    for out, sample in zip(outs, samples):
        for k, v in sample_losses(out, target, sample, loss_names, weights_ce, bias_l2).items():
            tot[k] = tot.get(k, 0.0) + v
    return {k: v / all_samples for k, v in tot.items()}


def model_outputs(x, params, out_channels, f_maps, num_levels, num_groups=8, max_surf_distance=3.0):
    feats = O.get_feature(x, params, f_maps=f_maps, num_levels=num_levels, num_groups=num_groups, unit_feat=True)
    out = O.task_heads(feats[-1], params, out_channels)
    return processors(out, max_surf_distance)


def iteration_loss(xs, params, target, samples, out_channels, loss_names, loss_weights, weights_ce, all_samples,
                   f_maps, num_levels, num_groups=8, max_surf_distance=3.0, bias_l2=True):
    outs = [model_outputs(x, params, out_channels, f_maps, num_levels, num_groups, max_surf_distance) for x in xs]
    ld = multi_criterion(outs, target, samples, loss_names, weights_ce, all_samples, bias_l2)
    total = sum(ld[k] * loss_weights[k] for k in ld if k in loss_weights)
    return total, ld


def clip_gradients(grads, clip):
    """utils/misc.py:1329-1338: each parameter's gradient is scaled to norm <= clip on its own."""
    norms, out = [], {}
    for k, g in grads.items():
        n = g.norm(2)
        norms.append(float(n))
        c = clip / (n + 1e-6)
        out[k] = g * c if c < 1 else g
    return out, norms


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / (bc2 ** 0.5) + eps)
    return p, m, v
