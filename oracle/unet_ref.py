"""Oracle for the inference path: functional CPU
restatement of the BrainFM 3D U-Net, task heads, processors, post-processor,
tiling and stitching.  Test infrastructure only (see oracle/__init__.py).

All tensors are NCDHW fp32 like the reference.  Weights come in as a plain
``dict`` keyed with the reference's state-dict names
(``backbone.encoders.0.basic_module.SingleConv1.groupnorm.weight`` ...).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# Trainer/models/__init__.py:16-24
LABELS_LEFT = [0, 1, 2, 3, 4, 7, 8, 9, 10, 14, 15, 17, 31, 34, 36, 38, 40, 42]
LABELS_FULL = [0, 11, 12, 13, 16, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46,
               1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 17, 47, 49, 51, 53, 55,
               18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 48, 50, 52, 54, 56]


def f_maps_per_level(f_maps, num_levels):
    """Trainer/models/unet3d/utils.py:109-110: f_maps * 2**k."""
    return [f_maps * 2 ** k for k in range(num_levels)]


def double_conv_channels(cin, cout, encoder):
    """DoubleConv channel rule, buildingblocks.py:131-141."""
    if encoder:
        c1 = cout // 2
        if c1 < cin:
            c1 = cin
        return [(cin, c1), (c1, cout)]
    return [(cin, cout), (cout, cout)]


def layer_table(in_channels, f_maps, num_levels):
    """All 3x3x3 conv layers in execution order.

    Returns a list of dicts {name, cin, cout, kind} where name is the state-dict
    prefix (``backbone.encoders.i.basic_module.SingleConvj``).  Follows
    create_encoders/create_decoders, buildingblocks.py:279-329.
    """
    fm = f_maps_per_level(f_maps, num_levels) if isinstance(f_maps, int) else list(f_maps)
    layers = []
    for i, co in enumerate(fm):
        ci = in_channels if i == 0 else fm[i - 1]
        for j, (a, b) in enumerate(double_conv_channels(ci, co, True)):
            layers.append(dict(name="backbone.encoders.%d.basic_module.SingleConv%d" % (i, j + 1),
                               cin=a, cout=b, kind="enc", level=i, idx=j))
    rev = list(reversed(fm))
    for i in range(len(rev) - 1):
        ci, co = rev[i] + rev[i + 1], rev[i + 1]
        for j, (a, b) in enumerate(double_conv_channels(ci, co, False)):
            layers.append(dict(name="backbone.decoders.%d.basic_module.SingleConv%d" % (i, j + 1),
                               cin=a, cout=b, kind="dec", level=i, idx=j))
    return layers


def single_conv(x, sd, prefix, num_groups=8):
    """'gcl' SingleConv: GroupNorm -> Conv3d(k3,p1,no bias) -> LeakyReLU(0.01).

    buildingblocks.py:31-60 (create_conv).  GroupNorm falls back to one group
    when the channel count is below num_groups (:56-57); eps is nn.GroupNorm's
    default 1e-5; LeakyReLU slope is nn.LeakyReLU's default 0.01 (:36).
    """
    c = x.shape[1]
    g = num_groups if c >= num_groups else 1
    x = F.group_norm(x, g, sd[prefix + ".groupnorm.weight"], sd[prefix + ".groupnorm.bias"], eps=1e-5)
    x = F.conv3d(x, sd[prefix + ".conv.weight"], None, padding=1)
    return F.leaky_relu(x, 0.01)


def get_feature(x, sd, in_channels=1, f_maps=64, num_levels=6, num_groups=8, unit_feat=True):
    """AbstractUNet.get_feature, unet3d/model.py:195-209.

    Encoder i>0 starts with MaxPool3d(2) (buildingblocks.py:185-186, :205-209);
    decoder = nearest interpolate to the skip's size, cat((skip, x), 1), DoubleConv
    (:265-276, :361-363).  Returns the list of decoder feature maps, deepest
    first, the last one L2-normalised over channels when unit_feat (model.py:207).
    """
    fm = f_maps_per_level(f_maps, num_levels) if isinstance(f_maps, int) else list(f_maps)
    skips = []
    for i in range(len(fm)):
        if i > 0:
            x = F.max_pool3d(x, 2)
        for j in (1, 2):
            x = single_conv(x, sd, "backbone.encoders.%d.basic_module.SingleConv%d" % (i, j), num_groups)
        skips.insert(0, x)
    skips = skips[1:]
    feats = [x]
    for i, skip in enumerate(skips):
        x = F.interpolate(x, size=skip.shape[2:], mode="nearest")
        x = torch.cat((skip, x), dim=1)
        for j in (1, 2):
            x = single_conv(x, sd, "backbone.decoders.%d.basic_module.SingleConv%d" % (i, j), num_groups)
        feats.append(x)
    if unit_feat:
        feats[-1] = F.normalize(feats[-1], dim=1)
    return feats


def task_heads(feat_last, sd, out_channels):
    """TaskHead.forward with task_f_maps=[64]: one 1x1x1 conv + bias per task.

    head.py:38-40,52-59.  ``out_channels`` is the ordered {task: n} mapping
    produced by process_args (Trainer/models/__init__.py:57-110).
    """
    out = OrderedDict()
    for name, n in out_channels.items():
        out[name] = F.conv3d(feat_last, sd["head.final_conv_%s.weight" % name], sd["head.final_conv_%s.bias" % name])
    return out


def default_out_channels(left_hemis_only=False, uncertainty=False, tasks=None):
    """process_args, Trainer/models/__init__.py:37-125, for the demo_test task set."""
    if tasks is None:
        tasks = ["T1", "T2", "FLAIR", "CT", "segmentation", "distance", "bias_field", "registration",
                 "super_resolution"]
    r = 2 if uncertainty else 1
    oc = OrderedDict()
    for t in ("T1", "T2", "FLAIR", "CT"):
        if t in tasks:
            oc[t] = r
    if "bias_field" in tasks:
        oc["bias_field_log"] = r
    if "segmentation" in tasks:
        oc["segmentation"] = len(LABELS_LEFT) if left_hemis_only else len(LABELS_FULL)
    if "distance" in tasks:
        oc["distance"] = 2 if left_hemis_only else 4
    if "registration" in tasks:
        oc["registration"] = 3
    if "surface" in tasks:                        # Trainer/models/__init__.py:103-106
        oc["surface"] = 8
    if "super_resolution" in tasks:
        oc["high_res_residual"] = r
    if "pathology" in tasks:
        oc["pathology"] = 1
    return oc


def processors_and_post(out, x_input, tasks, left_hemis_only=False, max_surf_distance=3.0):
    """SegProcessor + DistProcessor (joiner.py:69-77,149-157) followed by
    get_postprocessor (Trainer/models/__init__.py:272-354), single sample.

    ``out`` is mutated/returned like the reference does.
    """
    if "segmentation" in tasks:
        out["segmentation"] = torch.softmax(out["segmentation"], dim=1)
    if "distance" in tasks:
        out["distance"] = torch.clamp(out["distance"], min=-max_surf_distance, max=max_surf_distance)
    if "pathology" in tasks:
        out["pathology"] = torch.sigmoid(out["pathology"])
    if "super_resolution" in tasks:
        out["high_res"] = out["high_res_residual"] + x_input
    if "bias_field" in tasks:
        out["bias_field"] = torch.exp(out["bias_field_log"])
        del out["bias_field_log"]
    if "distance" in tasks:
        a = 2
        d = out["distance"]
        out["lp"], out["lw"] = d[:, 0][:, None], d[:, 1][:, None]
        fake = 70 * (1 - (torch.tanh(a * (out["lw"] + 0.3)) + 1) / 2) + 40 * (1 - (torch.tanh(a * out["lp"]) + 1) / 2)
        if not left_hemis_only:
            out["rp"], out["rw"] = d[:, 2][:, None], d[:, 3][:, None]
            fake_r = 70 * (1 - (torch.tanh(a * (out["rw"] + 0.3)) + 1) / 2) + \
                40 * (1 - (torch.tanh(a * out["rp"]) + 1) / 2)
            fake = fake + fake_r
        out["fake_cortical"] = fake
        del out["distance"]
    if "registration" in tasks:
        r = out["registration"]
        out["regx"], out["regy"], out["regz"] = r[:, 0][:, None], r[:, 1][:, None], r[:, 2][:, None]
        del out["registration"]
    if "segmentation" in tasks:
        lut = torch.tensor(LABELS_LEFT if left_hemis_only else LABELS_FULL)
        out["label"] = lut[torch.argmax(out["segmentation"], 1, keepdim=True)]
    if "CT" in tasks:
        out["CT"] = out["CT"] * 1000
    return out


def forward_all(x, sd, tasks=None, in_channels=1, f_maps=64, num_levels=6, num_groups=8, unit_feat=True,
                left_hemis_only=False, max_surf_distance=3.0, uncertainty=False):
    """evaluate_image minus config/checkpoint handling, utils/test_utils.py:289-312.

    uncertainty: train_args.losses.uncertainty is set (Trainer/models/__init__.py:57-111): the regression heads have
    two channels.  The reference's UncertaintyProcessor (joiner.py:45-56) splits `name + '_sigma'` off only for output
    names that contain 'image' -- none of process_args' names does -- so both channels stay in one tensor and go
    through the post-processor together (pinned by tests/golden/infer_uncert.npz)."""
    if tasks is None:
        tasks = ["T1", "T2", "FLAIR", "CT", "segmentation", "distance", "bias_field", "registration",
                 "super_resolution"]
    feats = get_feature(x, sd, in_channels, f_maps, num_levels, num_groups, unit_feat)
    out = OrderedDict(feat=feats)
    out.update(task_heads(feats[-1], sd, default_out_channels(left_hemis_only, uncertainty, tasks)))
    return processors_and_post(out, x, tasks, left_hemis_only, max_surf_distance)


# --------------------------------------------------------------------------
# tiling / stitching  (utils/test_utils.py:93-137, scripts/demo_test.py:66-119)
# --------------------------------------------------------------------------

def axis_intervals(n, win, stride):
    """One axis of `tiling`: first window is `win` wide, every later one is
    `stride` wide and the last is pulled back to end at n (test_utils.py:105-124)."""
    start, end = 0, min(win, n)
    out = [(start, end)]
    while end < n:
        start = min(end, n - stride)
        end = min(start + stride, n)
        out.append((start, end))
    return out


def tiling_ranges(shape, stride, win_size):
    xs = axis_intervals(shape[0], win_size[0], stride[0])
    ys = axis_intervals(shape[1], win_size[1], stride[1])
    zs = axis_intervals(shape[2], win_size[2], stride[2])
    ranges = [[x, y, z] for x in xs for y in ys for z in zs]
    cnt = np.zeros(shape, dtype=np.float32)
    for (x0, x1), (y0, y1), (z0, z1) in ranges:
        cnt[x0:x1, y0:y1, z0:z1] += 1
    return ranges, cnt


def tile_mask(im):
    """mask = im.clone(); mask[im != 0] = 1  (scripts/demo_test.py:88-89)."""
    m = im.clone()
    m[im != 0.] = 1.
    return m


def stitch(tile_outputs, ranges, cnt, shape):
    """full[range] += tile (tile order), full /= cnt  (scripts/demo_test.py:108-119).

    ``tile_outputs`` are already multiplied by the tile mask; labels are summed
    as floats like the reference does (quirk Q5).
    """
    full = torch.zeros(shape, dtype=torch.float32)
    for t, ((x0, x1), (y0, y1), (z0, z1)) in zip(tile_outputs, ranges):
        full[x0:x1, y0:y1, z0:z1] += t.to(torch.float32)
    return full / torch.as_tensor(cnt)


STITCH_KEYS = ["T1", "T2", "FLAIR", "CT", "high_res_residual", "high_res", "bias_field", "lp", "lw", "rp", "rw",
               "fake_cortical", "regx", "regy", "regz", "label", "deformed_atlas"]


def tiled_inference(full_im, sd, stride, win_size, atlas=None, **net_kw):
    """test_tile restated without disk round trips (scripts/demo_test.py:66-119).

    full_im: (1,1,D,H,W).  Returns {key: (D,H,W) fp32} for the 16 non-feat,
    non-segmentation keys, plus 'deformed_atlas' (:102-104, stitched like the
    others :108-119) when atlas = (MNI volume, its vox2ras affine) is given
    (the reference reads both from files/gca.mgz, utils/test_utils.py:38-43).
    """
    shape = tuple(full_im.shape[2:])
    ranges, cnt = tiling_ranges(shape, stride, win_size)
    per_key = {k: [] for k in STITCH_KEYS}
    if atlas is not None:
        from . import synth_ref
        MNI = np.asarray(atlas[0], dtype=np.float32)
        A = torch.tensor(np.linalg.inv(np.asarray(atlas[1], dtype=np.float64)), dtype=torch.float32).numpy()
    for (x0, x1), (y0, y1), (z0, z1) in ranges:
        im = full_im[:, :, x0:x1, y0:y1, z0:z1]
        outs = forward_all(im, sd, **net_kw)
        m = tile_mask(im)
        if atlas is not None:
            sq = lambda t: torch.squeeze(t).numpy()
            DEF = synth_ref.deformed_atlas(sq(m), sq(outs["regx"]), sq(outs["regy"]), sq(outs["regz"]), MNI, A)
            outs["deformed_atlas"] = torch.from_numpy(DEF)[None, None]
        for k in STITCH_KEYS:
            if k in outs:
                per_key[k].append(torch.squeeze(outs[k] * m))
    return {k: stitch(v, ranges, cnt, shape) for k, v in per_key.items() if v}, ranges, cnt


def random_state_dict(in_channels=1, f_maps=64, num_levels=6, out_channels=None, seed=1, affine_jitter=True):
    """Seeded random weights with the reference's key names and torch's default
    Conv3d init scale (kaiming-uniform, bound 1/sqrt(fan_in)).  GroupNorm affine
    is jittered away from (1,0) so that gamma/beta handling is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for L in layer_table(in_channels, f_maps, num_levels):
        ci, co = L["cin"], L["cout"]
        bound = 1.0 / np.sqrt(ci * 27)
        if affine_jitter:
            sd[L["name"] + ".groupnorm.weight"] = 1.0 + 0.2 * (torch.rand(ci, generator=g) - 0.5)
            sd[L["name"] + ".groupnorm.bias"] = 0.2 * (torch.rand(ci, generator=g) - 0.5)
        else:
            sd[L["name"] + ".groupnorm.weight"] = torch.ones(ci)
            sd[L["name"] + ".groupnorm.bias"] = torch.zeros(ci)
        sd[L["name"] + ".conv.weight"] = (torch.rand(co, ci, 3, 3, 3, generator=g) * 2 - 1) * bound
    fm = f_maps_per_level(f_maps, num_levels) if isinstance(f_maps, int) else list(f_maps)
    c0 = fm[0]
    if out_channels is None:
        out_channels = default_out_channels()
    for name, n in out_channels.items():
        bound = 1.0 / np.sqrt(c0)
        sd["head.final_conv_%s.weight" % name] = (torch.rand(n, c0, 1, 1, 1, generator=g) * 2 - 1) * bound
        sd["head.final_conv_%s.bias" % name] = (torch.rand(n, generator=g) * 2 - 1) * bound
    return sd
