"""The fused / device-resident kernels of the generator item (brainfm_amd/csrc/synth_item.hip) against the unfused chains
of entry points that the golden vectors of the real reference pin (tests/test_gpu_synth.py): every one must give the
chain's bits.  Needs an MI355X: run with `-m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def N(t):
    return t.detach().cpu().numpy()


def test_randn_philox_is_a_repeatable_standard_normal_stream():
    from scipy import stats
    from brainfm_amd import _lib as L
    lib = L.load()

    def draw(n, seed, off, scale=1.0):
        out = torch.full((n,), float("nan"), device=DEV)
        L.check(lib.bfm_randn_philox(L.ptr(out), n, C.c_uint64(seed), C.c_uint64(off), scale, L.stream_ptr()), "randn")
        return N(out)

    a = draw(1 << 20, 7, 0)
    assert np.isfinite(a).all()
    assert abs(a.mean()) < 4e-3 and abs(a.std() - 1) < 4e-3
    assert abs(stats.skew(a)) < 1e-2 and abs(stats.kurtosis(a)) < 3e-2
    assert stats.kstest(a[:200000], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5e-3                 # neighbours (same counter, next counter)
    assert np.array_equal(a, draw(1 << 20, 7, 0))                        # same (seed, offset): same field
    b, c = draw(1 << 20, 7, 1), draw(1 << 20, 8, 0)
    assert abs(np.corrcoef(a, b)[0, 1]) < 5e-3 and abs(np.corrcoef(a, c)[0, 1]) < 5e-3
    for n in (1, 2, 3, 5, 1023):                                         # tails that are not a multiple of 4
        assert np.array_equal(draw(n, 7, 0), a[:n])
    assert np.array_equal(draw(4096, 7, 0, 2.5), a[:4096] * np.float32(2.5))
    # the generator's draws: torch.manual_seed makes them repeatable, consecutive calls differ
    from brainfm_amd import generator_utils as GU
    d = GU.DeviceDraws()
    torch.manual_seed(5)
    x1, x2 = N(d.randn((4, 5, 6), DEV)), N(d.randn((4, 5, 6), DEV))
    torch.rand(2)                                                        # the host stream moves on, as inside an item
    x3 = N(d.randn((4, 5, 6), DEV))
    torch.manual_seed(5)
    y1, y2 = N(d.randn((4, 5, 6), DEV)), N(d.randn((4, 5, 6), DEV))
    assert np.array_equal(x1, y1) and np.array_equal(x2, y2) and not np.array_equal(x1, x2) and not np.array_equal(x1, x3)
    d.reseed()                                                           # same state, count restarted by hand
    assert np.array_equal(N(d.randn((4, 5, 6), DEV)), x1)
    # ... and consume nothing from the host stream: the host draws that follow are those of a run without fields
    torch.manual_seed(9)
    h0 = torch.rand(3)
    torch.manual_seed(9)
    d.randn((2, 2, 2), DEV)
    assert torch.equal(torch.rand(3), h0)


@pytest.mark.parametrize("photo", [False, True])
def test_deform_grid_zoomed_equals_zoom_then_deform_bitwise(photo):
    from brainfm_amd import generator as G, generator_utils as GU
    rs = np.random.RandomState(3)
    gen = object.__new__(G.BaseGen)
    gen.device = torch.device(DEV)
    gen.size = [40, 36, 44]
    small = (4, 3, 5)
    Fsmall = T(rs.randn(*small, 3).astype(np.float32) * 3)
    factor = np.array(gen.size) / np.array(small)
    A = GU.make_affine_matrix(np.array([0.1, -0.2, 0.15]), np.array([0.05, -0.1, 0.02]), np.array([1.1, 0.9, 1.05]))
    shp = [64, 60, 56]
    c2 = ((np.array(shp) - 1) / 2).astype(np.float32)
    F = GU.myzoom_torch(Fsmall, factor)
    if photo:
        F[:, :, :, 1] = 0
    ref = gen.deform_grid(shp, A.astype(np.float32), c2, F)
    F2, got = gen.deform_grid_zoomed(shp, A.astype(np.float32), c2, Fsmall, factor, photo)
    assert torch.equal(F2, F)
    for a, b in zip(ref[:3], got[:3]):
        assert torch.equal(a, b)
    assert list(ref[3:]) == list(got[3:])


def _chain_read_and_deform(vol, box, II, JJ, KK, mean, scale, default_max, post_div, clamp, sign, flip):
    """The unfused chain of round 3: host crop -> nan_to_num -> (x - mean) / scale -> max -> fast_3D_interp_torch -> / -> clamp."""
    from brainfm_amd import generator_utils as GU, _lib as L
    x1, y1, z1, x2, y2, z2 = box
    I = torch.nan_to_num(T(vol[x1:x2, y1:y2, z1:z2]))
    if mean != 0. or scale != 1.:
        I = GU.ew_unary(L.EW_SUB_DIV, I, mean, scale)
    dv = GU.tensor_max(I) if default_max else 0.
    out = GU.fast_3D_interp_torch(I, II, JJ, KK, "linear", dv)
    if post_div:
        out = GU.ew_unary(L.EW_DIV, out, post_div)
    if clamp is not None:
        out = GU.ew_unary(L.EW_CLAMP, out, clamp[0], clamp[1])
    if flip:
        out = torch.flip(out, [0])
    if sign:
        out = GU.ew_unary(L.EW_AFFINE, out.contiguous(), sign, 0.0)
    return out


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("prepared", [False, True])
def test_gather_targets_equals_the_crop_chain_bitwise(flip, prepared):
    """bfm_gather_targets over resident full volumes == crop + nan_to_num + (x-mean)/scale + max + interp + post per volume;
    `prepared`: the transform applied to the resident volume beforehand (what the generator keeps in HBM), else per texel."""
    from brainfm_amd import generator_utils as GU, _lib as L
    lib = L.load()
    rs = np.random.RandomState(11)
    shp = (50, 46, 54)
    out_shape = (32, 30, 34)
    box = (3, 2, 5, 47, 50, 52)                       # y2 beyond the volume: clipped like a NumPy slice
    cn = (44, 44, 47)
    II = T((rs.rand(*out_shape) * (cn[0] + 2) - 1).astype(np.float32))      # some coordinates outside the crop
    JJ = T((rs.rand(*out_shape) * (cn[1] + 2) - 1).astype(np.float32))
    KK = T((rs.rand(*out_shape) * (cn[2] + 2) - 1).astype(np.float32))
    II.view(-1)[:50] = T(np.arange(50, dtype=np.float32) % cn[0])            # exact integers, 0 and n-1 among them
    JJ.view(-1)[:50] = 0.0
    specs = [dict(mean=0., scale=1., default_max=False, post_div=0., clamp=None, sign=0.),
             dict(mean=128., scale=20., default_max=True, post_div=1.07, clamp=(-3., 3.), sign=0.),
             dict(mean=0., scale=10000., default_max=False, post_div=0., clamp=None, sign=-1.),
             dict(mean=128., scale=20., default_max=True, post_div=0.93, clamp=(-3., 3.), sign=0.)]
    vols = []
    for k in range(len(specs)):
        v = (rs.randn(*shp) * (100 if k != 2 else 5000) + (128 if k in (1, 3) else 0)).astype(np.float32)
        v[rs.rand(*shp) < 0.01] = np.nan
        v[rs.rand(*shp) < 0.002] = np.inf
        vols.append(v)
    outs = [torch.full(out_shape, float("nan"), device=DEV) for _ in specs]
    jobs = (L.GatherJob * len(specs))()
    keep = []
    for k, (v, sp) in enumerate(zip(vols, specs)):
        dv = T(v)
        pre, mean, scale = (2 if (sp["mean"] != 0. or sp["scale"] != 1.) else 1), sp["mean"], sp["scale"]
        if prepared:
            dv = GU.ew_unary(L.EW_NAN_TO_NUM, dv)
            if pre == 2:
                dv = GU.ew_unary(L.EW_SUB_DIV, dv, mean, scale)
            pre, mean, scale = 0, 0., 1.
        keep.append(dv)
        lo, hi = sp["clamp"] if sp["clamp"] else (0., 0.)
        jobs[k] = L.GatherJob(dv.data_ptr(), outs[k].data_ptr(), mean, scale, pre, int(sp["default_max"]), sp["post_div"],
                              int(sp["clamp"] is not None), lo, hi, sp["sign"], int(k == 0))
    scal = torch.zeros(3 * L.GATHER_MAX_JOBS, dtype=torch.float64, device=DEV)
    ws = torch.empty(lib.bfm_gather_targets_workspace(), dtype=torch.uint8, device=DEV)
    boxc = (C.c_int * 6)(*box)
    L.check(lib.bfm_gather_targets(jobs, len(specs), shp[0], shp[1], shp[2], boxc, L.ptr(II), L.ptr(JJ), L.ptr(KK),
                                   out_shape[0], out_shape[1], out_shape[2], int(flip), L.ptr(scal), L.ptr(ws), ws.numel(),
                                   L.stream_ptr()), "gather_targets")
    for k, (v, sp) in enumerate(zip(vols, specs)):
        ref = _chain_read_and_deform(v, box, II, JJ, KK, flip=flip, **sp)
        assert torch.equal(outs[k], ref), k
    assert float(scal[L.GATHER_MAX_JOBS]) == float(outs[0].min()) and float(scal[L.GATHER_MAX_JOBS + 1]) == float(outs[0].max())
    # read_and_deform_image's normalisation with the extrema on the device
    ref = outs[0].clone()
    ref -= ref.min()
    ref /= ref.max()
    L.check(lib.bfm_minmax_normalise(L.ptr(outs[0]), outs[0].numel(), C.c_void_p(scal.data_ptr() + 8 * L.GATHER_MAX_JOBS),
                                     L.stream_ptr()), "minmax_normalise")
    assert torch.equal(outs[0], ref)


@pytest.mark.parametrize("flip", [False, True])
def test_gather_onehot_equals_nearest_plus_onehot_bitwise(flip):
    from brainfm_amd import generator_utils as GU, _lib as L
    lib = L.load()
    rs = np.random.RandomState(5)
    shp, out_shape, box = (40, 44, 38), (24, 28, 26), (2, 0, 3, 39, 44, 36)
    cn = (37, 44, 33)
    S = rs.randint(0, 60, size=shp).astype(np.int32)
    lut_h = np.zeros(10000, np.int32)
    lut_h[:60] = rs.randint(0, 12, size=60)
    nl = 12
    vflip_h = np.array([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7], np.int32)
    II = T((rs.rand(*out_shape) * (cn[0] + 1) - 0.5).astype(np.float32))
    JJ = T((rs.rand(*out_shape) * (cn[1] + 1) - 0.5).astype(np.float32))
    KK = T((rs.rand(*out_shape) * (cn[2] + 1) - 0.5).astype(np.float32))
    II.view(-1)[:8] = T(np.array([0.5, 1.5, 2.5, 3.5, -0.5, 36.5, 35.5, 0.49999], np.float32))      # ties: half to even
    St, lut, vflip = T(S), T(lut_h), T(vflip_h)
    out = torch.full(out_shape + (nl,), float("nan"), device=DEV)
    L.check(lib.bfm_gather_onehot(L.ptr(St), shp[0], shp[1], shp[2], (C.c_int * 6)(*box), L.ptr(II), L.ptr(JJ), L.ptr(KK),
                                  out_shape[0], out_shape[1], out_shape[2], int(flip), L.ptr(lut), lut.numel(), nl,
                                  L.ptr(vflip) if flip else None, L.ptr(out), L.stream_ptr()), "gather_onehot")
    crop = T(S[box[0]:box[3], box[1]:box[4], box[2]:box[5]])
    Sdef = GU.fast_3D_interp_torch(crop, II, JJ, KK, "nearest").contiguous()
    ref = torch.empty(out_shape + (nl,), device=DEV)
    L.check(lib.bfm_onehot_lut(L.ptr(Sdef), L.ptr(lut), lut.numel(), nl, Sdef.numel(), L.ptr(ref), L.stream_ptr()), "onehot")
    if flip:
        ref = torch.flip(ref, [0])[:, :, :, vflip.long()]
    assert torch.equal(out, ref)
    # the class-major form the generator hands to the criterion: the same values in the order of the permuted view
    rows = torch.full((nl,) + out_shape, float("nan"), device=DEV)
    L.check(lib.bfm_gather_onehot_rows(L.ptr(St), shp[0], shp[1], shp[2], (C.c_int * 6)(*box), L.ptr(II), L.ptr(JJ), L.ptr(KK),
                                       out_shape[0], out_shape[1], out_shape[2], int(flip), L.ptr(lut), lut.numel(), nl,
                                       L.ptr(vflip) if flip else None, L.ptr(rows), L.stream_ptr()), "gather_onehot_rows")
    assert torch.equal(rows, ref.permute(3, 0, 1, 2))


def test_percentile_on_the_device_equals_numpy():
    from brainfm_amd import shapeid as SH
    rs = np.random.RandomState(2)
    cases = [rs.randn(200001),
             np.concatenate([rs.randn(5000), np.zeros(700), -np.zeros(300), -np.ones(300)]),
             np.array([3.0]), np.array([2.0, 1.0]), np.full(1000, -7.25),
             rs.rand(4099) * 1e-300, np.concatenate([-rs.rand(2000) * 1e300, rs.rand(3000)])]
    for x in cases:
        xt = T(x)
        for q in (0.0, 12.5, 50.0, 73.5, 85.0, 99.0, 99.9, 100.0):
            got = SH.percentile_dev(xt, q)
            assert float(got[0]) == float(np.percentile(x, q)), (len(x), q)
    # a Perlin field, the operand it is used on
    np.random.seed(4)
    noise = SH.generate_perlin_noise_3d((32, 32, 32), (2, 2, 2), tileable=(True, False, False), device=DEV)
    for q in (85.0, 91.7, 99.0):
        assert SH.percentile_linear(noise, q) == float(np.percentile(N(noise), q))
    # threshold + binarize with every scalar on the device == the two-step host form
    from brainfm_amd import generator_utils as GU
    masked, mask, mx = SH.threshold_at_percentile(noise, 91.7, want_max=True)
    n_np = N(noise)
    thr = np.percentile(n_np, 91.7)
    assert np.array_equal(N(mask), (n_np >= thr).astype(np.float64)) and np.array_equal(N(masked), n_np * (n_np >= thr))
    assert float(mx) == (n_np * (n_np >= thr)).max()
    P, psum = GU.binarize_dev(masked, 0.2, mx)
    ref = (N(masked) >= 0.2 * N(masked).max()).astype(np.float64)
    assert np.array_equal(N(P), ref) and float(psum) == ref.sum()
    assert torch.equal(GU.binarize(masked, 0.2), P)
    x32 = T(rs.rand(33, 17, 9).astype(np.float32))
    P32, s32 = GU.binarize_dev(x32, 0.37)
    ref32 = (N(x32) >= np.float32(0.37) * N(x32).max()).astype(np.float32)
    assert P32.dtype == torch.float32 and np.array_equal(N(P32), ref32) and float(s32) == ref32.sum()


@pytest.mark.parametrize("f64", [True, False])
def test_pathology_mask_and_encode_with_device_scalars(f64):
    from brainfm_amd import _lib as L, generator_utils as GU
    lib = L.load()
    rs = np.random.RandomState(9)
    shape = (20, 22, 24)
    dt = np.float64 if f64 else np.float32
    P_h = (rs.rand(*shape) > 0.7).astype(dt)
    Pp_h = (rs.rand(*shape) * P_h).astype(dt)
    cer_h = (rs.rand(*shape) * (rs.rand(*shape) > 0.2)).astype(np.float32)
    I_h = (rs.rand(*shape) * 200).astype(np.float32)
    rn_h = rs.randn(*shape).astype(np.float32)
    P, Pp, cer, I, rn = T(P_h), T(Pp_h), T(cer_h), T(I_h), T(rn_h)
    psum = torch.empty(1, dtype=torch.float64, device=DEV)
    ws = GU.workspace(torch.device(DEV))
    L.check(lib.bfm_pathology_mask(L.ptr(P), L.ptr(Pp), int(f64), L.ptr(cer), P.numel(), L.ptr(psum), L.ptr(ws), ws.numel(),
                                   L.stream_ptr()), "mask")
    Pm, Ppm = P_h * (cer_h != 0), Pp_h * (cer_h != 0)
    assert np.array_equal(N(P), Pm) and np.array_equal(N(Pp), Ppm) and float(psum) == Pm.sum()
    u = np.array([0.3, 0.8, 0.55, 0.1], np.float32)
    stats = T(np.array([50.0, 10.0, 90.0, 10.0]))                       # wm mean 5, gm mean 9 -> direction True
    for direction, want in ((1, True), (0, False), (-1, True)):
        out = torch.empty_like(I)
        dotsum = torch.empty(2, dtype=torch.float64, device=DEV)
        L.check(lib.bfm_pathology_encode_dev(L.ptr(I), L.ptr(P), L.ptr(Pp), int(f64), L.ptr(rn), (C.c_float * 4)(*u),
                                             direction, L.ptr(stats), I.numel(), L.ptr(out), L.ptr(dotsum), L.ptr(ws),
                                             ws.numel(), L.stream_ptr()), "encode")
        if f64:
            dot, tot = (I_h.astype(np.float64) * Pm).sum(), Pm.sum()
        else:
            dot, tot = (I_h * Pm).astype(np.float64).sum(), Pm.astype(np.float64).sum()
        assert abs(float(dotsum[0]) - dot) <= 1e-9 * abs(dot) and float(dotsum[1]) == tot
        I_mu = np.float32(float(dotsum[0]) / float(dotsum[1]))
        mus = np.float32(3) * I_mu / np.float32(4) + I_mu / np.float32(4) * u[:2]
        mus = mus if want else -mus
        sig = I_mu / np.float32(4) * u[2:]
        one = np.rint(Pm) >= 1
        g = (np.where(one, mus[1], mus[0]) + np.where(one, sig[1], sig[0]) * rn_h).astype(np.float32)
        if f64:
            ref = (I_h.astype(np.float64) + Ppm * g.astype(np.float64)).astype(np.float32)
        else:
            ref = I_h + Ppm * g
        ref = np.where(ref < 0, np.float32(0), ref)
        assert np.array_equal(N(out), ref), direction


def test_interp_axes_equals_the_meshgrid_sample_and_sample_finalize():
    from brainfm_amd import generator_utils as GU, _lib as L
    lib = L.load()
    rs = np.random.RandomState(4)
    X = T(rs.rand(30, 34, 28).astype(np.float32))
    v = [np.arange(-0.4, 30.2, 1.37), np.arange(0.0, 33.5, 2.11), np.arange(0.25, 27.01, 0.93)]
    II, JJ, KK = np.meshgrid(*v, sparse=False, indexing="ij")
    ref = GU.fast_3D_interp_torch(X, torch.tensor(II, dtype=torch.float, device=DEV),
                                  torch.tensor(JJ, dtype=torch.float, device=DEV),
                                  torch.tensor(KK, dtype=torch.float, device=DEV))
    tabs = [T(t.astype(np.float32)) for t in v]
    out = torch.empty(tuple(len(t) for t in v), device=DEV)
    L.check(lib.bfm_interp3d_linear_axes(L.ptr(X), 30, 34, 28, L.ptr(tabs[0]), L.ptr(tabs[1]), L.ptr(tabs[2]), out.shape[0],
                                         out.shape[1], out.shape[2], 0.0, L.ptr(out), L.stream_ptr()), "axes")
    assert torch.equal(out, ref)
    # resample_resolution end to end is pinned by test_resample_resolution_chain_with_reference_rng (golden)
    I = T((rs.rand(12, 14, 10) * 90).astype(np.float32))
    hr = T((rs.rand(12, 14, 10) * 110).astype(np.float32))
    mx = GU.reduce_dev(1, I)
    assert float(mx) == float(I.max())
    for flip in (False, True):
        a, b = torch.empty_like(I), torch.empty_like(I)
        L.check(lib.bfm_sample_finalize(L.ptr(I), L.ptr(hr), 12, 14, 10, L.ptr(mx), int(flip), L.ptr(a), L.ptr(b),
                                        L.stream_ptr()), "finalize")
        m = float(I.max())
        fin = GU.ew_unary(L.EW_DIV, I, m)
        res = GU.ew_binary(L.EW_AXPY, GU.ew_unary(L.EW_DIV, hr, m), fin, -1.0)
        if flip:
            fin, res = torch.flip(fin, [0]), torch.flip(res, [0])
        assert torch.equal(a, fin) and torch.equal(b, res)


def test_generator_item_is_repeatable_and_its_volumes_stay_resident():
    """Two datasets under the same seeds give the same item bit for bit (own Philox stream, fixed reduction orders), the
    second item of a dataset uploads nothing (case volumes resident in HBM), and the LRU budget is honoured."""
    from brainfm_amd import generator as G
    import test_gpu_synth as SY
    rs = np.random.RandomState(0)
    shp = (48, 44, 52)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
    ell = (((zz - 24) / 20.) ** 2 + ((yy - 22) / 18.) ** 2 + ((xx - 26) / 22.) ** 2) <= 1
    lab = ((zz // 8) * 7 + (yy // 8) * 3 + (xx // 8)) % 10
    ids = np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13])[lab] * ell
    case = {"name": "toy", "Gen": ids.astype(np.float32), "T1": rs.rand(*shp).astype(np.float32) * ell,
            "segmentation": ids.astype(np.int32),
            "distance": [rs.rand(*shp).astype(np.float32) * 255 for _ in range(4)],
            "registration": [rs.randn(*shp).astype(np.float32) * 500 for _ in range(3)]}

    def run():
        import random
        np.random.seed(3); torch.manual_seed(3); random.seed(3)
        ds = G.build_datasets(SY._gen_args(), DEV, cases=[case])["all"]
        return ds, ds[0]

    ds1, (_, _, _, t1, s1) = run()
    ds2, (_, _, _, t2, s2) = run()
    for k in t1:
        if isinstance(t1[k], torch.Tensor):
            assert torch.equal(t1[k], t2[k]), k
    for a, b in zip(s1, s2):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    n_resident, nbytes = len(ds1.volumes.items), ds1.volumes.bytes
    n_tensors = len({v[0].data_ptr() for v in ds1.volumes.items.values()})
    # Gen, T1, seg, 4 distance, 3 registration: ten resident tensors (an identity 'prep' request aliases the 'f32' copy)
    assert n_tensors == 10 and nbytes == 10 * int(np.prod(shp)) * 4, (n_resident, n_tensors, nbytes)
    ds1[0]
    assert len(ds1.volumes.items) == n_resident and ds1.volumes.bytes == nbytes
    small = G.DeviceVolumes(torch.device(DEV), budget=3 * int(np.prod(shp)) * 4)
    for v in case["distance"]:
        small.get(G.ArrayVolume(v), "f32")
    assert len(small.items) == 3 and small.bytes == 3 * int(np.prod(shp)) * 4
    small.forget(G.ArrayVolume(case["distance"][3]))
    assert len(small.items) == 2
    small.forget()
    assert len(small.items) == 0 and small.bytes == 0
    # an array re-filled in place is re-uploaded (content stamp), not served stale; an identity 'prep' copy of a
    # finite volume is the 'f32' copy itself
    src = rs.rand(*shp).astype(np.float32)
    vol = G.ArrayVolume(src)
    a = small.get(vol, "f32").clone()
    assert small.get(vol, "prep") is small.get(vol, "f32") and small.bytes == int(np.prod(shp)) * 4
    src[...] = rs.rand(*shp).astype(np.float32)
    b = small.get(vol, "f32")
    assert not torch.equal(a, b) and np.array_equal(b.cpu().numpy(), src)
    small.invalidate()
    assert len(small.items) == 0 and small.bytes == 0


@pytest.mark.parametrize("tag,nt", [("ode64", 6), ("ode32", 4)])
def test_dopri5_with_the_controller_on_the_device_equals_the_host_loop(tag, nt, monkeypatch):
    """bfm_dopri5_advect_* (one kernel per stage, error norm / accept / next step / dense output on the device) against
    the host-controlled loop of rounds 1-3 over the same kernels' expressions: same number of steps and RHS evaluations,
    solutions equal to the last bits (pow() in the step-size rule is the one operation evaluated by another library)."""
    from conftest import load_npz
    from brainfm_amd import shapeid as SH
    d = load_npz("synth_perlin_pde.npz")
    V = {"Vx": T(d["Vx40"]), "Vy": T(d["Vy40"]), "Vz": T(d["Vz40"])}
    t = torch.from_numpy(np.arange(10) * 0.1)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BFM_ODE_DEVICE", mode)
        pde = SH.AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict=V,
                            BC="neumann", dt=0.1, device=DEV)
        sol = SH.odeint_adjoint(pde, T(d[tag + "_y0"]), t[:nt], 0.1, method="dopri5")
        res[mode] = (N(sol), pde.nfe)
    assert res["1"][1] == res["0"][1] == int(d[tag + "_nfe"])
    a, b = res["1"][0], res["0"][0]
    assert a.shape == b.shape and a.dtype == b.dtype
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= (1e-12 if tag == "ode64" else 1e-6) * scale, np.abs(a - b).max() / scale
    # stiffer fields: step sizes clamped from both sides (x6), and a state that diverges (x40: the error norm turns NaN and
    # the rule must then do what Python's min / max do with a NaN) -- same step sequence on both paths
    for mult in (6, 40):
        Vbig = {k: v * mult for k, v in V.items()}
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("BFM_ODE_DEVICE", mode)
            pde = SH.AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict=Vbig,
                                BC="neumann", dt=0.1, device=DEV)
            sol = SH.odeint_adjoint(pde, T(d[tag + "_y0"]), t[:nt], 0.1, method="dopri5")
            out[mode] = (N(sol), pde.nfe)
        assert out["1"][1] == out["0"][1], mult
        if mult == 6:
            assert np.isfinite(out["0"][0]).all()
            assert np.abs(out["1"][0] - out["0"][0]).max() <= (1e-10 if tag == "ode64" else 1e-5) * np.abs(out["0"][0]).max()


def test_generator_item_from_volume_files_equals_the_in_memory_case(tmp_path):
    """A case whose volumes are NIfTI / MGH files opened with brainfm_amd.volio.load (nibabel's interface: shape, affine,
    dataobj, get_fdata -- what the reference's nib.load hands to Generator/utils.py:296-305) gives the item of the same
    volumes passed as arrays, bit for bit: the resident copy is made from dataobj / get_fdata once, in the dtype the
    reference converts to."""
    import random
    from brainfm_amd import generator as G, volio
    import test_gpu_synth as SY
    rs = np.random.RandomState(1)
    shp = (44, 48, 40)
    zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
    ell = (((zz - 22) / 18.) ** 2 + ((yy - 24) / 20.) ** 2 + ((xx - 20) / 17.) ** 2) <= 1
    lab = ((zz // 8) * 5 + (yy // 8) * 3 + (xx // 8)) % 10
    ids = (np.array([2, 3, 4, 41, 42, 17, 10, 11, 12, 13])[lab] * ell)
    mem = {"name": "c", "Gen": ids.astype(np.float32), "T1": (rs.rand(*shp) * ell).astype(np.float32),
           "segmentation": ids.astype(np.int32),
           "distance": [(rs.rand(*shp) * 255).astype(np.float32) for _ in range(4)],
           "registration": [(rs.randn(*shp) * 500).astype(np.float32) for _ in range(3)]}
    aff = np.diag([1.0, 1.0, 1.0, 1.0])

    def on_disk(name, arr, ext):
        path = str(tmp_path / (name + ext))
        volio.MRIwrite(arr, aff, path)
        return volio.load(path)

    files = {"name": "c", "Gen": on_disk("gen", mem["Gen"], ".nii.gz"), "T1": on_disk("t1", mem["T1"], ".mgz"),
             "segmentation": on_disk("seg", mem["segmentation"], ".nii"),
             "distance": [on_disk("d%d" % i, v, ".nii.gz") for i, v in enumerate(mem["distance"])],
             "registration": [on_disk("r%d" % i, v, ".nii") for i, v in enumerate(mem["registration"])]}

    def run(case):
        np.random.seed(5); torch.manual_seed(5); random.seed(5)
        ds = G.build_datasets(SY._gen_args(), DEV, cases=[case])["all"]
        return ds[0]

    _, _, _, t1, s1 = run(mem)
    _, _, _, t2, s2 = run(files)
    for k in t1:
        if isinstance(t1[k], torch.Tensor):
            assert torch.equal(t1[k], t2[k]), k
    for a, b in zip(s1, s2):
        for k in a:
            assert torch.equal(a[k], b[k]), k
