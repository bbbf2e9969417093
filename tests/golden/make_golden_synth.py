"""Generate golden vectors for the synthesis kernels by RUNNING the reference (build container only).

    python tests/golden/make_golden_synth.py

Random draws the reference makes internally (np.random / torch) are reproduced by re-seeding and
stored as explicit inputs, since RNG-stream parity across devices is not a goal.
"""
import ast
import os
import sys
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

R = ref_import.setup()
import torch  # noqa: E402

torch.set_num_threads(4)
import Generator.utils as GU  # noqa: E402
from ShapeID import perlin3d as P3  # noqa: E402
from ShapeID.DiffEqs.pde import AdvDiffPDE  # noqa: E402
from ShapeID.DiffEqs.adjoint import odeint_adjoint  # noqa: E402
import utils.interpol as interpol  # noqa: E402


def save(name, **d):
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, {k: getattr(v, "shape", None) for k, v in d.items()})


def interp():
    g = torch.Generator().manual_seed(0)
    X1 = torch.rand(9, 10, 11, generator=g)
    X3 = torch.rand(9, 10, 11, 3, generator=g)
    n = 4000
    II = torch.rand(n, generator=g) * 11 - 1
    JJ = torch.rand(n, generator=g) * 12 - 1
    KK = torch.rand(n, generator=g) * 13 - 1
    # edge cases: exact 0, exact n-1, integers, tiny positives, halves
    edge = torch.tensor([[0, 1, 1], [1e-6, 1, 1], [0.5, 1, 1], [8, 9, 10], [8.0001, 1, 1], [1, 0, 1], [1.5, 1.5, 1.5],
                         [-0.2, 1, 1], [8, 9, 10.0001], [3, 4, 5], [2.5, 3.5, 4.5], [7.999, 8.999, 9.999]])
    II = torch.cat([II, edge[:, 0]]); JJ = torch.cat([JJ, edge[:, 1]]); KK = torch.cat([KK, edge[:, 2]])
    d = dict(X1=X1.numpy(), X3=X3.numpy(), II=II.numpy(), JJ=JJ.numpy(), KK=KK.numpy())
    d["lin1"] = GU.fast_3D_interp_torch(X1, II, JJ, KK, "linear").numpy()
    d["lin1_def"] = GU.fast_3D_interp_torch(X1, II, JJ, KK, "linear", 7.5).numpy()
    d["lin3"] = GU.fast_3D_interp_torch(X3, II, JJ, KK, "linear").numpy()
    S = torch.randint(0, 60, (9, 10, 11), generator=g, dtype=torch.int32)
    d["S"] = S.numpy()
    # nearest mode indexes Y.shape[3]: the reference only works with 3-D coordinate grids there
    N3 = [(torch.rand(12, 13, 14, generator=g) * s - 1.0) for s in (11., 12., 13.)]
    N3[0][0, 0, :6] = torch.tensor([0.5, 1.5, 2.5, -0.6, 8.5, 7.5])
    d["N3i"], d["N3j"], d["N3k"] = [t.numpy() for t in N3]
    d["near_i"] = GU.fast_3D_interp_torch(S, N3[0], N3[1], N3[2], "nearest").numpy()
    d["near_f"] = GU.fast_3D_interp_torch(X3, N3[0], N3[1], N3[2], "nearest").numpy()
    # Appendix C of SURVEY: hand-checkable answers
    Xc = (torch.arange(27).view(3, 3, 3) + 1).float()
    pts = torch.tensor([[0, 1, 1], [1e-6, 1, 1], [0.5, 1, 1], [2, 2, 2], [2.0001, 1, 1], [1, 0, 1], [1.5, 1.5, 1.5],
                        [-0.2, 1, 1]])
    d["appc_pts"] = pts.numpy()
    d["appc_lin"] = GU.fast_3D_interp_torch(Xc, pts[:, 0], pts[:, 1], pts[:, 2], "linear").numpy()
    nn = torch.tensor([0.5, 1.5, 2.5, -0.6]).view(4, 1, 1)
    d["appc_near"] = GU.fast_3D_interp_torch(Xc, nn, torch.zeros(4, 1, 1), torch.zeros(4, 1, 1), "nearest").numpy()
    # 3-D coordinate grids (the shape the generator uses)
    G3 = [torch.rand(6, 7, 8, generator=g) * s - 0.5 for s in (9.5, 10.5, 11.5)]
    d["G3i"], d["G3j"], d["G3k"] = [t.numpy() for t in G3]
    d["lin_grid"] = GU.fast_3D_interp_torch(X1, G3[0], G3[1], G3[2], "linear").numpy()
    save("synth_interp.npz", **d)


def zoom_blur_aug():
    g = torch.Generator().manual_seed(1)
    d = {}
    X = torch.randn(5, 6, 7, 3, generator=g)
    f = np.array([32 / 5, 30 / 6, 28 / 7])
    d["zx"] = X.numpy(); d["zf"] = f
    d["zy"] = GU.myzoom_torch(X, f).numpy()
    X2 = torch.randn(4, 5, 3, generator=g)
    f2 = np.array([2.5, 1.0, 3.3])
    d["zx2"] = X2.numpy(); d["zf2"] = f2
    d["zy2"] = GU.myzoom_torch(X2, f2).numpy()
    X3 = torch.rand(20, 18, 22, generator=g)
    f3 = np.array([0.4, 0.55, 1 / 3.0])
    d["zx3"] = X3.numpy(); d["zf3"] = f3
    d["zy3"] = GU.myzoom_torch(X3, f3).numpy()
    d["appc_zoom"] = GU.myzoom_torch(torch.arange(4).float().view(4, 1, 1), np.array([2.5, 1, 1])).numpy()
    # gaussian blur
    d["gk1"] = GU.make_gaussian_kernel(1.0, "cpu").numpy()
    I = torch.rand(20, 18, 22, generator=g)
    stds = np.array([1.2, 0.0, 2.3])
    d["bI"] = I.numpy(); d["bstd"] = stds
    d["bO"] = GU.gaussian_blur_3d(I, stds, "cpu").numpy()
    # augmentation chain pieces with the reference's own RNG (captured by re-seeding)
    cfg = Namespace(gamma_std=0.1, bf_scale_min=0.02, bf_scale_max=0.04, bf_std_min=0.1, bf_std_max=0.6,
                    noise_std_min=0.05, noise_std_max=1.0)
    Iimg = torch.rand(40, 40, 40, generator=g) * 200
    np.random.seed(5)
    Ig, _ = GU.add_gamma_transform(Iimg.clone(), {}, cfg, "cpu")
    np.random.seed(5)
    gamma = np.exp(cfg.gamma_std * np.random.randn(1)[0])
    d["aug_I"] = Iimg.numpy(); d["aug_gamma"] = np.array(gamma); d["aug_Ig"] = Ig.numpy()
    np.random.seed(6); torch.manual_seed(6)
    setups = {"photo_mode": False, "spac": None}
    Ibf, aux = GU.add_bias_field(Iimg.clone(), {}, cfg, "synth", setups, [40, 40, 40], "cpu")
    np.random.seed(6); torch.manual_seed(6)
    bf_scale = cfg.bf_scale_min + np.random.rand(1) * (cfg.bf_scale_max - cfg.bf_scale_min)
    size_small = np.round(bf_scale * np.array([40, 40, 40])).astype(int).tolist()
    BFsmall = torch.tensor(cfg.bf_std_min + (cfg.bf_std_max - cfg.bf_std_min) * np.random.rand(1), dtype=torch.float) * \
        torch.randn(size_small, dtype=torch.float)
    d["bf_small"] = BFsmall.numpy(); d["bf_log"] = aux["BFlog"].numpy(); d["bf_I"] = Ibf.numpy()
    np.random.seed(7); torch.manual_seed(7)
    In, _ = GU.add_noise(Iimg.clone() - 100, {}, cfg, "cpu")
    np.random.seed(7); torch.manual_seed(7)
    nstd = cfg.noise_std_min + (cfg.noise_std_max - cfg.noise_std_min) * np.random.rand(1)
    rn = torch.randn(Iimg.shape, dtype=torch.float)
    d["noise_std"] = np.array(nstd, dtype=np.float32); d["noise_randn"] = rn.numpy(); d["noise_out"] = In.numpy()
    # resample_resolution: blur + trilinear down-sampling + zoom back (augment_sample tail)
    np.random.seed(8)
    setups = {"thickness": np.array([1.0, 4.2, 1.0]), "resolution": np.array([1.0, 3.5, 1.0])}
    Ismall, aux = GU.resample_resolution(Iimg.clone(), {}, setups, np.array([1.0, 1.0, 1.0]), [40, 40, 40], "cpu")
    np.random.seed(8)
    stds = (0.85 + 0.3 * np.random.rand()) * np.log(5) / np.pi * setups["thickness"] / np.array([1.0, 1.0, 1.0])
    stds[setups["thickness"] <= 1.0] = 0.0
    d["rs_stds"] = stds; d["rs_small"] = Ismall.numpy(); d["rs_factors"] = aux["factors"]
    d["rs_back"] = GU.myzoom_torch(Ismall, 1 / aux["factors"]).numpy()
    save("synth_zoom_blur_aug.npz", **d)


def perlin_pde():
    d = {}
    shape, res = (16, 12, 20), [2, 2, 2]
    np.random.seed(11)
    noise = P3.generate_perlin_noise_3d(shape, res, tileable=(True, False, False))
    np.random.seed(11)
    theta = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)
    phi = 2 * np.pi * np.random.rand(res[0] + 1, res[1] + 1, res[2] + 1)
    d["p_shape"] = np.array(shape); d["p_res"] = np.array(res)
    d["p_theta"], d["p_phi"], d["p_noise"] = theta, phi, noise
    np.random.seed(12)
    pm, m = P3.generate_perlin_noise_3d(shape, res, tileable=(True, False, False), percentile=73.5)
    np.random.seed(12)
    d["pm_theta"] = 2 * np.pi * np.random.rand(3, 3, 3); d["pm_phi"] = 2 * np.pi * np.random.rand(3, 3, 3)
    d["pm_noise"], d["pm_mask"], d["pm_pct"] = pm, m, np.array(73.5)
    # non-cubic resolution, no tiling
    np.random.seed(13)
    n2 = P3.generate_perlin_noise_3d((12, 12, 18), [3, 2, 3])
    np.random.seed(13)
    d["p2_theta"] = 2 * np.pi * np.random.rand(4, 3, 4); d["p2_phi"] = 2 * np.pi * np.random.rand(4, 3, 4)
    d["p2_noise"] = n2
    # velocity (3 potentials + curl) x V_multiplier
    np.random.seed(14)
    V = P3.generate_velocity_3d(shape, res, 500, "cpu")
    np.random.seed(14)
    for nm in "abc":
        d["v_theta_" + nm] = 2 * np.pi * np.random.rand(3, 3, 3)
        d["v_phi_" + nm] = 2 * np.pi * np.random.rand(3, 3, 3)
    d["Vx"], d["Vy"], d["Vz"] = V["Vx"].numpy(), V["Vy"].numpy(), V["Vz"].numpy()
    # advection RHS, fp32 and fp64 state
    # the fixture volume is 10x smaller than the generator's 160^3, so the shipped V_multiplier=500
    # would violate the CFL limit of the clamped step (dt >= 0.02) and blow up: use 40 for the PDE part
    np.random.seed(14)
    V = P3.generate_velocity_3d(shape, res, 40, "cpu")
    d["Vx40"], d["Vy40"], d["Vz40"] = V["Vx"].numpy(), V["Vy"].numpy(), V["Vz"].numpy()
    pde = AdvDiffPDE(data_spacing=[1., 1., 1.], perf_pattern="adv", V_type="vector_div_free", V_dict=V, BC="neumann",
                     dt=0.1, device="cpu")
    g = torch.Generator().manual_seed(3)
    C32 = torch.rand(1, *shape, generator=g)
    C64 = torch.rand(1, *shape, generator=g, dtype=torch.float64)
    d["C32"], d["C64"] = C32.numpy(), C64.numpy()
    with torch.no_grad():
        d["rhs32"] = pde(torch.tensor(0.), C32).numpy()
        d["rhs64"] = pde(torch.tensor(0.), C64).numpy()
        # dopri5 as the generator drives it (Generator/utils.py:549-554)
        t = torch.from_numpy(np.arange(10) * 0.1)
        nfe = [0]
        orig = pde.forward

        def counted(tt, y):
            nfe[0] += 1
            return orig(tt, y)
        pde.forward = counted
        y64 = torch.from_numpy(pm)[None]                       # generate_shape_3d path: fp64 state
        sol64 = odeint_adjoint(pde, y64, t[:6], 0.1, method="dopri5")
        d["ode64_y0"], d["ode64_sol"], d["ode64_nfe"] = y64.numpy(), sol64.numpy(), np.array(nfe[0])
        nfe[0] = 0
        y32 = torch.from_numpy(pm.astype(np.float32))[None]     # read-from-file path: fp32 state
        sol32 = odeint_adjoint(pde, y32, t[:4], 0.1, method="dopri5")
        d["ode32_y0"], d["ode32_sol"], d["ode32_nfe"] = y32.numpy(), sol32.numpy(), np.array(nfe[0])
    save("synth_perlin_pde.npz", **d)


def grid_pull():
    d = {}
    g = torch.Generator().manual_seed(21)
    vol = torch.randn(2, 2, 5, 6, 7, generator=g)
    ident = torch.stack(torch.meshgrid(torch.arange(4.), torch.arange(5.), torch.arange(6.), indexing="ij"), -1)
    grid = ident[None] * torch.tensor([1.3, 1.25, 1.2]) + torch.randn(2, 4, 5, 6, 3, generator=g) * 1.5 - 0.7
    d["vol"], d["grid"] = vol.numpy(), grid.numpy()
    for b in ["zero", "replicate", "dct1", "dct2", "dst1", "dst2", "dft"]:
        for ex in (False, True):
            out = interpol.grid_pull(vol, grid, interpolation="linear", bound=b, extrapolate=ex, prefilter=False)
            d["out_%s_%d" % (b, int(ex))] = out.numpy()
    X = (torch.arange(27).view(1, 1, 3, 3, 3) + 1).float()
    pts = torch.tensor([[0, 1, 1], [1e-6, 1, 1], [0.5, 1, 1], [2, 2, 2], [2.0001, 1, 1], [1, 0, 1], [1.5, 1.5, 1.5],
                        [-0.2, 1, 1]]).view(1, 8, 1, 1, 3)
    d["appc_pts"] = pts.numpy()
    d["appc_zero_0"] = interpol.grid_pull(X, pts, bound="zero", extrapolate=False).numpy()
    d["appc_zero_1"] = interpol.grid_pull(X, pts, bound="zero", extrapolate=True).numpy()
    d["appc_dct2_1"] = interpol.grid_pull(X, pts, bound="dct2", extrapolate=True).numpy()
    save("synth_grid_pull.npz", **d)


def deform_and_atlas():
    from Generator.datasets import BaseGen
    d = {}
    gen = object.__new__(BaseGen)
    gen.device = "cpu"
    gen.synth_args = Namespace(size=[24, 20, 28])
    gen.prepare_grid()
    np.random.seed(31); torch.manual_seed(31)
    A = torch.tensor(GU.make_affine_matrix(np.array([0.1, -0.2, 0.15]), np.array([0.05, -0.1, 0.02]),
                                           np.array([1.1, 0.9, 1.05])), dtype=torch.float)
    shp = (40, 36, 44)
    c2 = torch.tensor((np.array(shp) - 1) / 2, dtype=torch.float)
    Fsmall = 2.0 * torch.randn(3, 3, 3, 3)
    F = GU.myzoom_torch(Fsmall, np.array(gen.size) / np.array([3, 3, 3]))
    xx2, yy2, zz2, x1, y1, z1, x2, y2, z2 = gen.deform_grid(shp, A, c2, F)
    d.update(dg_A=A.numpy(), dg_c2=c2.numpy(), dg_F=F.numpy(), dg_size=np.array(gen.size), dg_shp=np.array(shp),
             dg_xx=xx2.numpy(), dg_yy=yy2.numpy(), dg_zz=zz2.numpy(), dg_lo=np.array([x1, y1, z1]),
             dg_hi=np.array([x2, y2, z2]), dg_Fsmall=Fsmall.numpy())
    d["affine_mat"] = GU.make_affine_matrix(np.array([0.1, -0.2, 0.15]), np.array([0.05, -0.1, 0.02]),
                                            np.array([1.1, 0.9, 1.05]))
    # get_deformed_atlas (utils/test_utils.py:45-57) without importing the module (it reads files/gca.mgz via nibabel)
    src = open(R + "/utils/test_utils.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_deformed_atlas"]
    g = torch.Generator().manual_seed(32)
    MNI = torch.rand(30, 32, 28, generator=g)
    Aaff = torch.tensor([[0.9, 0.05, 0.0, 14.0], [0.0, 1.1, -0.05, 15.0], [0.02, 0.0, 1.0, 13.0], [0, 0, 0, 1.0]])
    ns = {"torch": torch, "A": Aaff, "MNI": MNI, "fast_3D_interp_torch": GU.fast_3D_interp_torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "x", "exec"), ns)
    mask = (torch.rand(16, 14, 18, generator=g) > 0.3).float()
    reg = [torch.randn(16, 14, 18, generator=g) * 0.06 for _ in range(3)]
    out = ns["get_deformed_atlas"](mask, *reg)
    d.update(at_MNI=MNI.numpy(), at_A=Aaff.numpy(), at_mask=mask.numpy(), at_rx=reg[0].numpy(), at_ry=reg[1].numpy(),
             at_rz=reg[2].numpy(), at_out=out.numpy())
    # contrast synthesis + one-hot (Generator/datasets.py:366-372, Generator/utils.py:408-411)
    gen.synth_args = Namespace(size=[24, 20, 28], ct_prob=0, left_hemis_only=False)
    gen.prepare_one_hot()
    G = torch.randint(0, 256, (12, 10, 14), generator=g).float()
    G[0, 0, :5] = 77
    torch.manual_seed(33); np.random.seed(33)
    mus, sigmas = gen.get_contrast(False)
    torch.manual_seed(34)
    Gr = torch.round(torch.where(G == 77, torch.tensor(2.), G)).long()
    rn = torch.randn(Gr.shape, dtype=torch.float)
    SYN = mus[Gr] + sigmas[Gr] * rn
    SYN[SYN < 0] = 0
    d.update(cs_G=G.numpy(), cs_mus=mus.numpy(), cs_sigmas=sigmas.numpy(), cs_randn=rn.numpy(), cs_out=SYN.numpy())
    S = torch.tensor(np.random.choice(np.array(sorted(set(range(57)) - {45})), size=(6, 5, 7))).int()
    d["oh_S"] = S.numpy(); d["oh_lut"] = gen.lut.numpy()[:64]
    d["oh_out"] = gen.onehotmatrix[gen.lut[S.long()]].numpy()
    d["vflip"] = gen.vflip
    save("synth_deform_atlas.npz", **d)


if __name__ == "__main__":
    interp()
    zoom_blur_aug()
    perlin_pde()
    grid_pull()
    deform_and_atlas()
    print({f: os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.startswith("synth_")})
