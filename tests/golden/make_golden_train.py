"""Generate golden vectors for one training iteration by RUNNING the reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_train.py
Writes tests/golden/train_step.npz: seeded inputs and targets, the weights the reference modules drew from torch's
RNG, and what the reference computes for them -- the loss dictionary of SetMultiCriterion, the gradient of the weighted
total w.r.t. every parameter (torch autograd, model in float64 so that the vectors are a truth and not torch-fp32's
rounding), the per-parameter clipping of utils.misc.clip_gradients and the parameters after one torch.optim.AdamW step.
The iteration follows Trainer/engine.py:96-147 (model -> processors -> criterion -> weighted sum -> backward -> clip ->
step) without autocast / GradScaler (their scale cancels exactly in fp64).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_infer as M  # noqa: E402  (sets up the reference import harness)

R = M.R
import torch  # noqa: E402


def main():
    import utils.misc as um
    from Trainer.models import build_model
    gen_args = um.preprocess_cfg([R + "/cfgs/generator/default.yaml", R + "/cfgs/generator/test/demo_test.yaml"],
                                 cfg_dir="")
    train_args = um.preprocess_cfg([R + "/cfgs/trainer/default_train.yaml", R + "/cfgs/trainer/default_val.yaml",
                                    R + "/cfgs/trainer/test/demo_test.yaml"], cfg_dir="")
    f_maps, levels = 8, 3
    train_args.f_maps = f_maps
    train_args.num_levels = levels
    train_args.task_f_maps = [f_maps]

    def hemis(g):
        g.generator.left_hemis_only = True          # 18 classes, 2 distance channels: small fixture
    hemis(gen_args)
    torch.manual_seed(21)
    gen_args, train_args, model, processors, criterion, post = build_model(gen_args, train_args, "cpu")
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if "groupnorm.weight" in k:
                v.copy_(1.0 + 0.4 * (torch.rand(v.shape, generator=g) - 0.5))
            if "groupnorm.bias" in k:
                v.copy_(0.4 * (torch.rand(v.shape, generator=g) - 0.5))
    sd32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.double()
    model.train()
    criterion.train()
    criterion.weights_ce = criterion.weights_ce.double()
    criterion.weights_dice = criterion.weights_dice.double()
    # unequal loss weights so that a swapped weight shows
    wd = criterion.weight_dict
    for i, k in enumerate(sorted(wd)):
        wd[k] = float(0.5 + 0.25 * (i % 5))

    dims = (12, 16, 10)
    n_seg = gen_args.n_labels
    n_dist = 2
    n_samples = 2
    d = {}
    samples, x32 = [], []
    for i in range(n_samples):
        x = torch.rand((1, 1) + dims, generator=g)
        x32.append(x)
        bf = 0.3 * torch.randn((1, 1) + dims, generator=g)
        hr = 0.2 * torch.randn((1, 1) + dims, generator=g)
        samples.append({"input": x.double(), "bias_field_log": bf.double(), "high_res_residual": hr.double()})
        d["x%d" % i] = x.numpy()
        d["bias_field_log%d" % i] = bf.numpy()
        d["high_res_residual%d" % i] = hr.numpy()
    lab = torch.randint(0, n_seg, (1,) + dims, generator=g)
    onehot = torch.nn.functional.one_hot(lab, n_seg).permute(0, 4, 1, 2, 3).float().contiguous()
    target = {"segmentation": onehot}
    for k in ("T1", "T2", "FLAIR", "CT"):
        target[k] = torch.rand((1, 1) + dims, generator=g)
    target["T1_DM"] = (torch.rand((1, 1) + dims, generator=g) > 0.8).float()       # loss_T1 / loss_T1_grad weights
    target["distance"] = torch.clamp(2.5 * torch.randn((1, n_dist) + dims, generator=g), -3, 3)
    target["registration"] = torch.randn((1, 3) + dims, generator=g)
    for k, v in target.items():
        d["target/" + k] = v.numpy()
    target64 = {k: v.double() for k, v in target.items()}

    # make the distance head large enough that the DistProcessor clamp is active on part of the volume
    with torch.no_grad():
        model.head.final_conv_distance.weight.mul_(8.0)
        sd32["head.final_conv_distance.weight"] = sd32["head.final_conv_distance.weight"] * 8.0

    lr, wdecay, clip = 1e-3, 0.04, 0.05
    opt = torch.optim.AdamW([{"params": [p for p in model.parameters() if p.requires_grad]}])
    for gr in opt.param_groups:
        gr["lr"] = lr
        gr["weight_decay"] = wdecay
    opt.zero_grad()
    outputs, _ = model(samples)
    for p in processors:
        outputs = p(outputs, target64, "synth")
    loss_dict = criterion(outputs, target64, samples)
    losses = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)
    losses.backward()
    names = [n for n, _ in model.named_parameters()]
    for n, p in model.named_parameters():
        d["grad/" + n] = p.grad.detach().numpy().copy()
    norms = um.clip_gradients(model, clip)
    d["clip_norms"] = np.array(norms, dtype=np.float64)
    for n, p in model.named_parameters():
        d["clipped/" + n] = p.grad.detach().numpy().copy()
    opt.step()
    for n, p in model.named_parameters():
        d["after/" + n] = p.detach().numpy().copy()
    for k, v in loss_dict.items():
        d["loss/" + k] = np.float64(float(v.detach()))
    d["loss_total"] = np.float64(float(losses.detach()))
    d["loss_weight_names"] = np.array(sorted(wd))
    d["loss_weights"] = np.array([wd[k] for k in sorted(wd)], dtype=np.float64)
    d["loss_names"] = np.array(list(criterion.loss_names))
    d["param_names"] = np.array(names)
    d["hyper"] = np.array([lr, wdecay, clip, 0.9, 0.999, 1e-8, float(gen_args.generator.all_samples),
                           float(gen_args.max_surf_distance)], dtype=np.float64)
    d["bias_field_log_type"] = np.array(str(train_args.losses.bias_field_log_type))
    d["weights_ce"] = criterion.weights_ce.reshape(-1).numpy()
    d["cfg"] = np.array([f_maps, levels, 8])
    for k, v in sd32.items():
        d["sd/" + k] = v.numpy()
    # a float32 run of the same iteration: how far torch-fp32 itself sits from the float64 truth (reported by the test)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **d)
    print("train_step:", {k: float(v) for k, v in loss_dict.items()})
    print("total", float(losses), "params", len(names), "clip norms", np.round(norms[:6], 4))
    frac = float((outputs[0]["distance"].abs() >= 3).double().mean())
    print("distance clamp active on %.1f%% of voxels" % (100 * frac))


if __name__ == "__main__":
    main()
