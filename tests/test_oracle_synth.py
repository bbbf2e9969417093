"""The synthesis oracle (oracle/synth_ref.py) against golden vectors produced by the real reference
(tests/golden/make_golden_synth.py).  CPU only."""
import numpy as np

from conftest import load_npz
from oracle import synth_ref as S


def _close(a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    e = float(np.abs(a - b).max()) / max(1e-6, float(np.abs(b).max()))
    assert e <= tol, e


def test_interp_linear_and_nearest_bit_exact():
    d = load_npz("synth_interp.npz")
    assert np.array_equal(S.interp3d_linear(d["X1"], d["II"], d["JJ"], d["KK"]), d["lin1"])
    assert np.array_equal(S.interp3d_linear(d["X1"], d["II"], d["JJ"], d["KK"], 7.5), d["lin1_def"])
    assert np.array_equal(S.interp3d_linear(d["X3"], d["II"], d["JJ"], d["KK"]), d["lin3"])
    assert np.array_equal(S.interp3d_linear(d["X1"], d["G3i"], d["G3j"], d["G3k"]), d["lin_grid"])
    assert np.array_equal(S.interp3d_nearest(d["S"], d["N3i"], d["N3j"], d["N3k"]), d["near_i"])
    assert np.array_equal(S.interp3d_nearest(d["X3"], d["N3i"], d["N3j"], d["N3k"]), d["near_f"])
    # SURVEY appendix C
    Xc = (np.arange(27).reshape(3, 3, 3) + 1).astype(np.float32)
    p = d["appc_pts"]
    lin = S.interp3d_linear(Xc, p[:, 0], p[:, 1], p[:, 2])
    assert np.array_equal(lin, d["appc_lin"])
    assert lin[0] == 0 and lin[2] == 9.5 and lin[3] == 27 and lin[4] == 0 and lin[6] == 20.5
    nn = np.array([0.5, 1.5, 2.5, -0.6], np.float32).reshape(4, 1, 1)
    z = np.zeros((4, 1, 1), np.float32)
    assert S.interp3d_nearest(Xc, nn, z, z).ravel().tolist() == [1, 19, 19, 1]


def test_zoom_blur_augmentations():
    d = load_npz("synth_zoom_blur_aug.npz")
    assert np.array_equal(S.myzoom(d["zx"], d["zf"]), d["zy"])
    assert np.array_equal(S.myzoom(d["zx2"], d["zf2"]), d["zy2"])
    assert np.array_equal(S.myzoom(d["zx3"], d["zf3"]), d["zy3"])
    z = S.myzoom(np.arange(4, dtype=np.float32).reshape(4, 1, 1), np.array([2.5, 1, 1]))
    assert np.array_equal(z, d["appc_zoom"])
    _close(S.make_gaussian_kernel(1.0), d["gk1"], 1e-6)
    _close(S.gaussian_blur_3d(d["bI"], d["bstd"]), d["bO"], 2e-6)
    _close(S.gamma_transform(d["aug_I"], float(d["aug_gamma"])), d["aug_Ig"], 2e-6)
    bflog = S.myzoom(d["bf_small"], np.array([40, 40, 40]) / np.array(d["bf_small"].shape))
    assert np.array_equal(bflog, d["bf_log"])
    _close(S.apply_bias_field(d["aug_I"], bflog), d["bf_I"], 1e-6)
    assert np.array_equal(S.add_noise(d["aug_I"] - 100, d["noise_std"][0], d["noise_randn"]), d["noise_out"])
    # resample_resolution: blur, trilinear down-sampling on the reference's coordinate grid, zoom back
    I = d["aug_I"]
    blur = S.gaussian_blur_3d(I, d["rs_stds"])
    size = np.array([40, 40, 40]); new = d["rs_small"].shape
    fac = d["rs_factors"]
    delta = (1.0 - fac) / (2.0 * fac)
    v = [np.arange(delta[a], delta[a] + new[a] / fac[a], 1 / fac[a])[:new[a]] for a in range(3)]
    II, JJ, KK = np.meshgrid(*v, sparse=False, indexing="ij")
    small = S.interp3d_linear(blur, II.astype(np.float32), JJ.astype(np.float32), KK.astype(np.float32))
    _close(small, d["rs_small"], 3e-6)
    assert np.array_equal(S.myzoom(d["rs_small"], 1 / fac), d["rs_back"])


def test_perlin_curl_bit_exact_and_percentile():
    d = load_npz("synth_perlin_pde.npz")
    shape, res = tuple(d["p_shape"]), tuple(d["p_res"])
    g = S.perlin_gradients(d["p_theta"], d["p_phi"], (True, False, False))
    assert np.array_equal(S.perlin_noise_3d(shape, res, g), d["p_noise"])
    g = S.perlin_gradients(d["pm_theta"], d["pm_phi"], (True, False, False))
    nm, m, thr = S.percentile_mask(S.perlin_noise_3d(shape, res, g), float(d["pm_pct"]))
    assert np.array_equal(nm, d["pm_noise"]) and np.array_equal(m, d["pm_mask"])
    g = S.perlin_gradients(d["p2_theta"], d["p2_phi"])
    assert np.array_equal(S.perlin_noise_3d((12, 12, 18), (3, 2, 3), g), d["p2_noise"])
    pots = [S.perlin_noise_3d(shape, res, S.perlin_gradients(d["v_theta_" + n], d["v_phi_" + n], (True, False, False)))
            for n in "abc"]
    Vx, Vy, Vz = S.stream_3d(*pots, multiplier=500)
    assert np.array_equal(Vx, d["Vx"]) and np.array_equal(Vy, d["Vy"]) and np.array_equal(Vz, d["Vz"])


def test_advection_rhs_bit_exact():
    d = load_npz("synth_perlin_pde.npz")
    V = (d["Vx40"], d["Vy40"], d["Vz40"])
    assert np.array_equal(S.advect_rhs(d["C32"][0], *V), d["rhs32"][0])
    assert np.array_equal(S.advect_rhs(d["C64"][0], *V), d["rhs64"][0])


def test_dopri5_matches_reference_solver():
    d = load_npz("synth_perlin_pde.npz")
    V = (d["Vx40"], d["Vy40"], d["Vz40"])
    t = np.arange(10) * 0.1
    for tag, nt in (("ode64", 6), ("ode32", 4)):
        st = {}
        sol = S.dopri5_integrate(lambda y: S.advect_rhs(y, *V), d[tag + "_y0"][0], t[:nt], stats=st)
        assert sol.dtype == d[tag + "_sol"].dtype
        assert st["nfe"] == int(d[tag + "_nfe"]), (st, int(d[tag + "_nfe"]))
        _close(sol, d[tag + "_sol"][:, 0], 1e-6 if tag == "ode64" else 2e-5)   # fp32 state: rounding-order noise


def test_grid_pull_all_bounds():
    d = load_npz("synth_grid_pull.npz")
    for b in ["zero", "replicate", "dct1", "dct2", "dst1", "dst2", "dft"]:
        for ex in (0, 1):
            out = S.grid_pull_linear(d["vol"], d["grid"], b, bool(ex))
            _close(out, d["out_%s_%d" % (b, ex)], 1e-6)
    X = (np.arange(27).reshape(1, 1, 3, 3, 3) + 1).astype(np.float32)
    z0 = S.grid_pull_linear(X, d["appc_pts"], "zero", False).ravel()
    _close(z0, d["appc_zero_0"].ravel(), 1e-6)
    assert z0[0] == 5 and z0[5] == 11 and z0[7] == 0          # differs from fast_3D_interp_torch
    _close(S.grid_pull_linear(X, d["appc_pts"], "zero", True).ravel(), d["appc_zero_1"].ravel(), 1e-6)
    _close(S.grid_pull_linear(X, d["appc_pts"], "dct2", True).ravel(), d["appc_dct2_1"].ravel(), 1e-6)


def test_deform_grid_atlas_contrast_onehot():
    d = load_npz("synth_deform_atlas.npz")
    F = S.myzoom(d["dg_Fsmall"], np.array(d["dg_size"]) / np.array([3, 3, 3]))
    assert np.array_equal(F, d["dg_F"])
    xx, yy, zz, lo, hi = S.deform_grid(list(d["dg_size"]), list(d["dg_shp"]), d["dg_A"], d["dg_c2"], F)
    assert lo == list(d["dg_lo"]) and hi == list(d["dg_hi"])
    _close(xx, d["dg_xx"], 1e-6); _close(yy, d["dg_yy"], 1e-6); _close(zz, d["dg_zz"], 1e-6)
    out = S.deformed_atlas(d["at_mask"], d["at_rx"], d["at_ry"], d["at_rz"], d["at_MNI"], d["at_A"])
    _close(out, d["at_out"], 1e-6)
    assert np.array_equal(S.synth_from_labels(d["cs_G"], d["cs_mus"], d["cs_sigmas"], d["cs_randn"]), d["cs_out"])
    lut = np.zeros(10000, np.int64); lut[:64] = d["oh_lut"]
    assert np.array_equal(S.onehot_lut(d["oh_S"], lut, 56), d["oh_out"])


def test_cubic_bspline_resize_oracle_vs_reference_golden():
    """The oracle's separable fp64 restatement of interpol.resize(interpolation=3, bound='dct2', prefilter=True)
    against the reference's own (fp32, 64-tap) evaluation: prefilter coefficients and resized volumes."""
    d = load_npz("interpol_resize.npz")
    for name, anchor in (("up", "e"), ("down", "e"), ("centers", "c")):
        x = d[name + "/x"]
        coeff = S.bspline3_prefilter_dct2(x)
        assert np.abs(coeff - d[name + "/coeff"]).max() <= 2e-5 * np.abs(d[name + "/coeff"]).max()
        y = S.resize_cubic_ref(x, [int(v) for v in d[name + "/shape"]], anchor)
        assert y.shape == d[name + "/y"].shape
        assert np.abs(y - d[name + "/y"]).max() <= 2e-5 * np.abs(d[name + "/y"]).max(), name


def test_grid_push_grad_oracle_vs_reference_golden():
    """fp64 restatements of iso1.push3d / grad3d against the vendored torch-interpol (fp32) for all seven boundary
    conditions, with and without extrapolation."""
    d = load_npz("interpol_pushgrad.npz")
    vol, grid, src = d["vol"], d["grid"], d["src"]
    for bound in range(7):
        for ext in (0, 1):
            k = "b%d_e%d/" % (bound, ext)
            push = S.grid_push_linear(src, grid, vol.shape[2:], bound, bool(ext))
            assert np.abs(push - d[k + "push"]).max() <= 2e-5 * max(1.0, np.abs(d[k + "push"]).max()), k
            grad = S.grid_grad_linear(vol, grid, bound, bool(ext))
            assert np.abs(grad - d[k + "grad"]).max() <= 2e-5 * max(1.0, np.abs(d[k + "grad"]).max()), k
