"""The CPU oracle against golden vectors produced by the real reference
(tests/golden/make_golden_infer.py).  CPU only."""
import numpy as np
import torch

from conftest import load_npz, sd_from_npz
from oracle import unet_ref as O

RTOL = 2e-5  # same ATen kernels, but thread count / op order may differ slightly


def _close(a, b, tol=RTOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert err <= tol, err


def test_small_net_all_outputs():
    d = load_npz("infer_small.npz")
    f_maps, levels, groups = [int(v) for v in d["cfg"]]
    sd = sd_from_npz(d)
    x = torch.from_numpy(d["x"])
    out = O.forward_all(x, sd, f_maps=f_maps, num_levels=levels, num_groups=groups)
    for i, f in enumerate(out["feat"]):
        _close(f.numpy(), d["feat%d" % i])
    keys = [k[4:] for k in d if k.startswith("out/")]
    assert sorted(keys) == sorted(k for k in out if k != "feat")
    for k in keys:
        if k == "label":
            assert out[k].dtype == torch.int64
            assert np.array_equal(out[k].numpy(), d["out/label"])
        else:
            _close(out[k].numpy(), d["out/" + k])


def test_left_hemis_head_set():
    d = load_npz("infer_hemis.npz")
    f_maps, levels, groups = [int(v) for v in d["cfg"]]
    sd = sd_from_npz(d)
    x = torch.from_numpy(d["x"])
    out = O.forward_all(x, sd, f_maps=f_maps, num_levels=levels, num_groups=groups, left_hemis_only=True)
    assert "rp" not in out and out["segmentation"].shape[1] == 18
    for k in [k[4:] for k in d if k.startswith("out/")]:
        if k == "label":
            assert np.array_equal(out[k].numpy(), d["out/label"])
        else:
            _close(out[k].numpy(), d["out/" + k])


def test_full_width_blocks():
    d = load_npz("infer_layers.npz")
    sd = {}
    for blk, pre in (("enc0", "backbone.encoders.0."), ("enc1", "backbone.encoders.1."), ("dec", "backbone.decoders.0.")):
        for k, v in d.items():
            if k.startswith(blk + "/"):
                sd[pre + k[len(blk) + 1:]] = torch.from_numpy(v)
    x = torch.from_numpy(d["x"])
    e0 = O.single_conv(O.single_conv(x, sd, "backbone.encoders.0.basic_module.SingleConv1"), sd,
                       "backbone.encoders.0.basic_module.SingleConv2")
    _close(e0.numpy(), d["e0"])
    p = torch.nn.functional.max_pool3d(e0, 2)
    e1 = O.single_conv(O.single_conv(p, sd, "backbone.encoders.1.basic_module.SingleConv1"), sd,
                       "backbone.encoders.1.basic_module.SingleConv2")
    _close(e1.numpy(), d["e1"])
    up = torch.nn.functional.interpolate(e1, size=e0.shape[2:], mode="nearest")
    y = torch.cat((e0, up), 1)
    y = O.single_conv(O.single_conv(y, sd, "backbone.decoders.0.basic_module.SingleConv1"), sd,
                      "backbone.decoders.0.basic_module.SingleConv2")
    _close(y.numpy(), d["y"])


def test_tiling_ranges_match_reference():
    d = load_npz("tiling_ranges.npz")
    for n in (160, 200, 256, 512):
        ranges, cnt = O.tiling_ranges((n, n, n), [80] * 3, [160] * 3)
        assert np.array_equal(np.array(ranges), d["ranges_%d" % n])
        assert np.array_equal(np.bincount(cnt.astype(np.int64).ravel(), minlength=9), d["cnt_%d_hist" % n])
        assert np.array_equal(cnt[np.arange(n), np.arange(n), np.arange(n)], d["cnt_%d_diag" % n])
    assert np.array_equal(np.array(O.axis_intervals(512, 160, 80)), d["ranges_512_x"])
    # 256 -> (0,160),(160,240),(176,256)
    assert O.axis_intervals(256, 160, 80) == [(0, 160), (160, 240), (176, 256)]


def test_tiled_stitch_toy():
    d = load_npz("infer_tiled.npz")
    f_maps, levels, groups, stride, win = [int(v) for v in d["cfg"]]
    sd = sd_from_npz(d)
    full = torch.from_numpy(d["full"])
    out, ranges, cnt = O.tiled_inference(full, sd, [stride] * 3, [win] * 3, atlas=(d["atlas"], d["atlas_aff"]),
                                         f_maps=f_maps, num_levels=levels, num_groups=groups)
    assert np.array_equal(np.array(ranges), d["ranges"])
    assert np.array_equal(cnt, d["cnt"])
    keys = [k[9:] for k in d if k.startswith("stitched/")]
    assert len(keys) == 17 and keys[-1] == "deformed_atlas" and list(out.keys()) == keys    # scripts/demo_test.py:107-119
    for k in keys:
        _close(out[k].numpy(), d["stitched/" + k], 5e-5)


def test_wide_net_all_outputs_and_tiled_stitch():
    """The 64-wide 2-level net of infer_wide.npz (the fixture that pins the build's matrix-core kernels to the reference):
    the oracle reproduces the reference's single-volume outputs, labels included, and the 17 stitched keys."""
    d = load_npz("infer_wide.npz")
    f_maps, levels, groups, stride, win = [int(v) for v in d["cfg"]]
    sd = sd_from_npz(d)
    out = O.forward_all(torch.from_numpy(d["x"]), sd, f_maps=f_maps, num_levels=levels, num_groups=groups)
    for i, f in enumerate(out["feat"]):
        _close(f.numpy(), d["feat%d" % i])
    for k in [k[4:] for k in d if k.startswith("out/")]:
        if k == "label":
            assert np.array_equal(out[k].numpy(), d["out/label"])
        else:
            _close(out[k].numpy(), d["out/" + k])
    st, ranges, cnt = O.tiled_inference(torch.from_numpy(d["full"]), sd, [stride] * 3, [win] * 3,
                                        atlas=(d["atlas"], d["atlas_aff"]), f_maps=f_maps, num_levels=levels,
                                        num_groups=groups)
    keys = [k[9:] for k in d if k.startswith("stitched/")]
    assert len(keys) == 17 and list(st.keys()) == keys
    for k in keys:
        _close(st[k].numpy(), d["stitched/" + k], 5e-5)


def test_uncertainty_head_set():
    """`losses.uncertainty` (Trainer/models/__init__.py:57-111): two channels per regression head, both kept in one
    tensor by the reference (its UncertaintyProcessor matches no output name) and post-processed together."""
    d = load_npz("infer_uncert.npz")
    f_maps, levels, groups = [int(v) for v in d["cfg"]]
    assert list(d["processors"]) == ["UncertaintyProcessor", "SegProcessor", "DistProcessor"]
    out = O.forward_all(torch.from_numpy(d["x"]), sd_from_npz(d), f_maps=f_maps, num_levels=levels, num_groups=groups,
                        uncertainty=True)
    keys = [k[4:] for k in d if k.startswith("out/")]
    assert sorted(keys) == sorted(k for k in out if k != "feat")
    for k in ("T1", "T2", "FLAIR", "CT", "bias_field", "high_res_residual", "high_res"):
        assert out[k].shape[1] == 2 and d["out/" + k].shape[1] == 2
    for k in keys:
        assert tuple(out[k].shape) == d["out/" + k].shape, k
        if k == "label":
            assert np.array_equal(out[k].numpy(), d["out/label"])
        else:
            _close(out[k].numpy(), d["out/" + k])
    _close(out["feat"][-1].numpy(), d["feat_last"])
