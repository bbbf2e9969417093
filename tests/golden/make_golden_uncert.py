"""Golden vectors for the `losses.uncertainty` head set, by RUNNING the reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_uncert.py
Writes tests/golden/infer_uncert.npz: the reference's build_model with train_args.losses.uncertainty = 'gaussian'
(Trainer/models/__init__.py:57-111: T1/T2/FLAIR/CT/bias_field_log/high_res_residual heads get a second, sigma channel;
joiner.py:238-241 puts UncertaintyProcessor first), a small net, model -> processors -> postprocessor on a seeded
volume.  Data only: input, the weights the reference drew, every output of the reference's dict.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

R = ref_import.setup()
import torch  # noqa: E402

torch.set_num_threads(8)


def main():
    import utils.misc as um
    from Trainer.models import build_model
    gen_args = um.preprocess_cfg([R + "/cfgs/generator/default.yaml", R + "/cfgs/generator/test/demo_test.yaml"],
                                 cfg_dir="")
    train_args = um.preprocess_cfg([R + "/cfgs/trainer/default_train.yaml", R + "/cfgs/trainer/default_val.yaml",
                                    R + "/cfgs/trainer/test/demo_test.yaml"], cfg_dir="")
    f_maps, levels = 8, 3
    train_args.f_maps, train_args.num_levels, train_args.task_f_maps = f_maps, levels, [f_maps]
    train_args.losses.uncertainty = "gaussian"
    torch.manual_seed(7)
    gen_args, train_args, model, processors, criterion, post = build_model(gen_args, train_args, "cpu")
    g = torch.Generator().manual_seed(107)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if "groupnorm.weight" in k:
                v.copy_(1.0 + 0.4 * (torch.rand(v.shape, generator=g) - 0.5))
            if "groupnorm.bias" in k:
                v.copy_(0.4 * (torch.rand(v.shape, generator=g) - 0.5))
    model.eval()
    torch.manual_seed(8)
    x = torch.rand(1, 1, 16, 12, 20)
    with torch.no_grad():
        samples = [{"input": x}]
        outs, _ = model(samples)
        for p in processors:
            outs = p(outs, samples)
        outs, _, _ = post(gen_args, train_args, outs, samples, target=None, feats=None, tasks=gen_args.tasks)
    o = outs[0]
    d = {"sd/" + k: v.detach().numpy() for k, v in model.state_dict().items()}
    d["x"] = x.numpy()
    d["cfg"] = np.array([f_maps, levels, 8])
    d["processors"] = np.array([type(p).__name__ for p in processors])
    d["out_channels"] = np.array(["%s=%d" % kv for kv in train_args.out_channels.items()])
    d["output_names"] = np.array(list(train_args.output_names))
    d["aux_output_names"] = np.array(list(train_args.aux_output_names))
    for k, v in o.items():
        if k == "feat":
            d["feat_last"] = v[-1].numpy()
        else:
            d["out/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "infer_uncert.npz"), **d)
    print({k: v.shape for k, v in d.items() if not k.startswith("sd/")})
    print(list(d["processors"]), list(d["out_channels"]))


if __name__ == "__main__":
    main()
