"""Generate golden vectors for the inference path by RUNNING the reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_infer.py
Writes tests/golden/infer_small.npz, infer_layers.npz, infer_tiled.npz,
tiling_ranges.npz.  The fixtures are data only: seeded inputs, the weights the
reference modules drew from torch's RNG, and the reference's outputs.
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

R = ref_import.setup()
import torch  # noqa: E402

torch.set_num_threads(8)


def np_sd(sd):
    return {"sd/" + k: v.detach().cpu().numpy() for k, v in sd.items()}


def load_ref_functions(path, names):
    """Compile selected top-level functions of a reference file without importing
    the module (utils/test_utils.py reads an atlas through nibabel at import)."""
    src = open(path).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"torch": torch, "np": np, "print": lambda *a, **k: None}
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def build(f_maps, num_levels, seed, gen_overrides=None):
    import utils.misc as um
    from Trainer.models import build_model
    gen_args = um.preprocess_cfg([R + "/cfgs/generator/default.yaml", R + "/cfgs/generator/test/demo_test.yaml"],
                                 cfg_dir="")
    train_args = um.preprocess_cfg([R + "/cfgs/trainer/default_train.yaml", R + "/cfgs/trainer/default_val.yaml",
                                    R + "/cfgs/trainer/test/demo_test.yaml"], cfg_dir="")
    train_args.f_maps = f_maps
    train_args.num_levels = num_levels
    train_args.task_f_maps = [f_maps]
    if gen_overrides:
        gen_overrides(gen_args)
    torch.manual_seed(seed)
    gen_args, train_args, model, processors, criterion, post = build_model(gen_args, train_args, "cpu")
    # move GroupNorm affine away from (1, 0) so gamma/beta are exercised
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if "groupnorm.weight" in k:
                v.copy_(1.0 + 0.4 * (torch.rand(v.shape, generator=g) - 0.5))
            if "groupnorm.bias" in k:
                v.copy_(0.4 * (torch.rand(v.shape, generator=g) - 0.5))
    model.eval()
    return gen_args, train_args, model, processors, post


@torch.no_grad()
def run(gen_args, train_args, model, processors, post, x):
    samples = [{"input": x}]
    outs, _ = model(samples)
    for p in processors:
        outs = p(outs, samples)
    outs, _, _ = post(gen_args, train_args, outs, samples, target=None, feats=None, tasks=gen_args.tasks)
    return outs[0]


def small():
    gen_args, train_args, model, processors, post = build(8, 4, seed=1)
    torch.manual_seed(0)
    x = torch.rand(1, 1, 20, 18, 22)
    o = run(gen_args, train_args, model, processors, post, x)
    d = np_sd(model.state_dict())
    d["x"] = x.numpy()
    for i, f in enumerate(o["feat"]):
        d["feat%d" % i] = f.numpy()
    for k, v in o.items():
        if k != "feat":
            d["out/" + k] = v.numpy()
    d["cfg"] = np.array([8, 4, 8])  # f_maps, num_levels, num_groups
    np.savez_compressed(os.path.join(HERE, "infer_small.npz"), **d)
    print("infer_small", {k: v.shape for k, v in d.items() if not k.startswith("sd/")})

    # left-hemisphere head set (18-class LUT, 2 distance channels) -- cfgs/generator/test/demo_test_hemis.yaml
    def hemis(g):
        g.generator.left_hemis_only = True
    gen_args, train_args, model, processors, post = build(8, 3, seed=3, gen_overrides=hemis)
    torch.manual_seed(5)
    x = torch.rand(1, 1, 12, 16, 8)
    o = run(gen_args, train_args, model, processors, post, x)
    d = np_sd(model.state_dict())
    d["x"] = x.numpy()
    for k, v in o.items():
        if k != "feat":
            d["out/" + k] = v.numpy()
    d["feat_last"] = o["feat"][-1].numpy()
    d["cfg"] = np.array([8, 3, 8])
    np.savez_compressed(os.path.join(HERE, "infer_hemis.npz"), **d)
    print("infer_hemis", sorted(k for k in d if k.startswith("out/")))


@torch.no_grad()
def layers():
    """Full-width single blocks through the reference Encoder / Decoder modules."""
    from Trainer.models.unet3d.buildingblocks import Encoder, Decoder, DoubleConv
    d = {}
    torch.manual_seed(11)
    enc0 = Encoder(1, 64, apply_pooling=False, basic_module=DoubleConv, conv_layer_order="gcl", num_groups=8).eval()
    enc1 = Encoder(64, 128, basic_module=DoubleConv, conv_layer_order="gcl", num_groups=8).eval()
    dec = Decoder(128 + 64, 64, basic_module=DoubleConv, conv_layer_order="gcl", num_groups=8).eval()
    g = torch.Generator().manual_seed(12)
    for m in (enc0, enc1, dec):
        for k, v in m.state_dict().items():
            if "groupnorm.weight" in k:
                v.copy_(1.0 + 0.4 * (torch.rand(v.shape, generator=g) - 0.5))
            if "groupnorm.bias" in k:
                v.copy_(0.4 * (torch.rand(v.shape, generator=g) - 0.5))
    x = torch.rand(1, 1, 12, 10, 14, generator=g)
    e0 = enc0(x)
    e1 = enc1(e0)
    y = dec(e0, e1)
    for name, m in (("enc0", enc0), ("enc1", enc1), ("dec", dec)):
        for k, v in m.state_dict().items():
            d["%s/%s" % (name, k)] = v.numpy()
    d.update(x=x.numpy(), e0=e0.numpy(), e1=e1.numpy(), y=y.numpy())
    # odd sizes: pool 5->2 and nearest 2->5, 3->7
    p = torch.rand(1, 3, 5, 7, 3, generator=g)
    d["pool_in"] = p.numpy()
    d["pool_out"] = torch.nn.functional.max_pool3d(p, 2).numpy()
    u = torch.rand(1, 2, 2, 3, 5, generator=g)
    d["up_in"] = u.numpy()
    d["up_out"] = torch.nn.functional.interpolate(u, size=(5, 7, 10), mode="nearest").numpy()
    np.savez_compressed(os.path.join(HERE, "infer_layers.npz"), **d)
    print("infer_layers", {k: v.shape for k, v in d.items() if "/" not in k})


@torch.no_grad()
def tiled():
    tiling, zero_crop = load_ref_functions(R + "/utils/test_utils.py", ["tiling", "zero_crop"])
    globals_ = tiling.__globals__
    globals_["zero_crop"] = zero_crop
    d = {}
    # interval lists / cnt for the BASELINE shapes (scripts/demo_test.py:126 uses stride 80, win 160)
    for n in (160, 200, 256, 512):
        if n == 512:
            img = torch.zeros(1, 1, n, 8, 8)  # the x interval list alone (round-1 key, kept)
            lst, cnt = tiling(img, stride=[80, 80, 80], win_size=[160, 160, 160])
            d["ranges_%d_x" % n] = np.array(sorted(set(tuple(r[0]) for _, r in lst)))
        img = torch.zeros(1, 1, n, n, n)
        lst, cnt = tiling(img, stride=[80, 80, 80], win_size=[160, 160, 160])
        d["ranges_%d" % n] = np.array([r for _, r in lst])
        d["cnt_%d_hist" % n] = np.bincount(cnt.numpy().astype(np.int64).ravel(), minlength=9)
        d["cnt_%d_diag" % n] = cnt.numpy()[np.arange(n), np.arange(n), np.arange(n)]
    np.savez_compressed(os.path.join(HERE, "tiling_ranges.npz"), **d)
    print("tiling_ranges", {k: v.shape for k, v in d.items()})

    # toy tiled inference + stitch, emulating scripts/demo_test.py:75-119 in memory
    gen_args, train_args, model, processors, post = build(8, 3, seed=7)
    torch.manual_seed(2)
    D, H, W = 40, 36, 44
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    ell = (((zz - D / 2 + .5) / 17.) ** 2 + ((yy - H / 2 + .5) / 15.) ** 2 + ((xx - W / 2 + .5) / 19.) ** 2) <= 1
    full = torch.rand(1, 1, D, H, W) * ell[None, None]
    im_list, cnt = tiling(full, stride=[12, 12, 12], win_size=[24, 24, 24])
    # the reference's own get_deformed_atlas (utils/test_utils.py:45-57) on a small synthetic atlas: its module
    # globals MNI / A (:38-43, read from files/gca.mgz at import) are set here the way the module sets them
    (get_deformed_atlas,) = load_ref_functions(R + "/utils/test_utils.py", ["get_deformed_atlas"])
    from Generator.utils import fast_3D_interp_torch
    ai, aj, ak = torch.meshgrid(torch.arange(26.), torch.arange(30.), torch.arange(22.), indexing="ij")
    atlas = 100. + 60. * torch.sin(ai / 3.1) * torch.cos(aj / 4.3) + 40. * torch.sin(ak / 2.7 + 0.5)   # smooth, like an MRI atlas
    c, s_ = np.cos(0.3), np.sin(0.3)
    aff2 = np.array([[-9. * c, 9. * s_, 0., 110.], [0., 0., 8., -95.], [-9. * s_, -9. * c, 0., 120.], [0., 0., 0., 1.]])
    get_deformed_atlas.__globals__.update(MNI=atlas.to(torch.float32), fast_3D_interp_torch=fast_3D_interp_torch,
                                          A=torch.tensor(np.linalg.inv(aff2), dtype=torch.float32))
    keys = None
    acc = {}
    inside = 0
    for im, rng in im_list:
        o = run(gen_args, train_args, model, processors, post, im.clone())
        mask = im.clone()
        mask[im != 0.] = 1.
        # scripts/demo_test.py:102-104,108: computed per tile from the unmasked registration maps, saved * mask, and
        # appended to the keys the stitch loop walks
        o["deformed_atlas"] = get_deformed_atlas(torch.squeeze(mask), torch.squeeze(o["regx"]), torch.squeeze(o["regy"]),
                                                 torch.squeeze(o["regz"]))
        inside += int((o["deformed_atlas"] != 0).sum())
        if keys is None:
            keys = [k for k in o if "feat" not in k and "segmentation" not in k]
            acc = {k: torch.zeros_like(torch.squeeze(full)) for k in keys}
        (x0, x1), (y0, y1), (z0, z1) = rng
        for k in keys:
            v = torch.squeeze(o[k] * mask)
            if "label" in k:  # read_image(..., is_label=True) -> torch.int
                v = v.to(torch.int)
            acc[k][x0:x1, y0:y1, z0:z1] += v
    out = np_sd(model.state_dict())
    out["full"] = full.numpy()
    out["cnt"] = cnt.numpy()
    out["ranges"] = np.array([r for _, r in im_list])
    out["atlas"] = atlas.numpy()
    out["atlas_aff"] = aff2
    print("deformed_atlas: %d tile voxels sampled inside the atlas" % inside)
    for k in keys:
        out["stitched/" + k] = (acc[k] / cnt).numpy()
    out["cfg"] = np.array([8, 3, 8, 12, 24])
    np.savez_compressed(os.path.join(HERE, "infer_tiled.npz"), **out)
    print("infer_tiled", len(im_list), "tiles", keys)


@torch.no_grad()
def wide():
    """The full pipeline of the reference on a 64-wide, 2-level net -- every conv except the stem is a matrix-core layer
    in the build (Winograd, up-folded decoder conv, split-fp16 MFMA), so this pins THOSE kernels to the reference itself
    (labels included), not only to the oracle: single-volume outputs (all heads, int64 labels, features) and the tiled
    flow of scripts/demo_test.py:66-119 with the 17 stitched keys."""
    tiling, get_deformed_atlas = load_ref_functions(R + "/utils/test_utils.py", ["tiling", "get_deformed_atlas"])
    from Generator.utils import fast_3D_interp_torch
    gen_args, train_args, model, processors, post = build(64, 2, seed=13)
    torch.manual_seed(6)
    x = torch.rand(1, 1, 24, 20, 28)
    x[:, :, :, :3] = 0
    o = run(gen_args, train_args, model, processors, post, x)
    d = np_sd(model.state_dict())
    d["x"] = x.numpy()
    for i, f in enumerate(o["feat"]):
        d["feat%d" % i] = f.numpy()
    for k, v in o.items():
        if k != "feat":
            d["out/" + k] = v.numpy()
    d["cfg"] = np.array([64, 2, 8, 12, 24])               # f_maps, num_levels, num_groups, stride, win
    # tiled flow with the deformed atlas
    D, H, W = 40, 36, 44
    zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
    ell = (((zz - D / 2 + .5) / 17.) ** 2 + ((yy - H / 2 + .5) / 15.) ** 2 + ((xx - W / 2 + .5) / 19.) ** 2) <= 1
    full = torch.rand(1, 1, D, H, W) * ell[None, None]
    ai, aj, ak = torch.meshgrid(torch.arange(26.), torch.arange(30.), torch.arange(22.), indexing="ij")
    atlas = 100. + 60. * torch.sin(ai / 3.1) * torch.cos(aj / 4.3) + 40. * torch.sin(ak / 2.7 + 0.5)
    c, s_ = np.cos(0.3), np.sin(0.3)
    aff2 = np.array([[-9. * c, 9. * s_, 0., 110.], [0., 0., 8., -95.], [-9. * s_, -9. * c, 0., 120.], [0., 0., 0., 1.]])
    get_deformed_atlas.__globals__.update(MNI=atlas.to(torch.float32), fast_3D_interp_torch=fast_3D_interp_torch,
                                          A=torch.tensor(np.linalg.inv(aff2), dtype=torch.float32))
    im_list, cnt = tiling(full, stride=[12, 12, 12], win_size=[24, 24, 24])
    keys, acc = None, {}
    for im, rng in im_list:
        o = run(gen_args, train_args, model, processors, post, im.clone())
        mask = im.clone()
        mask[im != 0.] = 1.
        o["deformed_atlas"] = get_deformed_atlas(torch.squeeze(mask), torch.squeeze(o["regx"]), torch.squeeze(o["regy"]),
                                                 torch.squeeze(o["regz"]))
        if keys is None:
            keys = [k for k in o if "feat" not in k and "segmentation" not in k]
            acc = {k: torch.zeros_like(torch.squeeze(full)) for k in keys}
        (x0, x1), (y0, y1), (z0, z1) = rng
        for k in keys:
            v = torch.squeeze(o[k] * mask)
            if "label" in k:
                v = v.to(torch.int)
            acc[k][x0:x1, y0:y1, z0:z1] += v
    d["full"] = full.numpy()
    d["cnt"] = cnt.numpy()
    d["atlas"] = atlas.numpy()
    d["atlas_aff"] = aff2
    for k in keys:
        d["stitched/" + k] = (acc[k] / cnt).numpy()
    np.savez_compressed(os.path.join(HERE, "infer_wide.npz"), **d)
    print("infer_wide", len(im_list), "tiles", keys, os.path.getsize(os.path.join(HERE, "infer_wide.npz")), "bytes")


if __name__ == "__main__":
    small()
    layers()
    tiled()
    wide()
    print("sizes:", {f: os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz")})
