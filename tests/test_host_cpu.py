"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, host-side
tiling logic matches the reference's golden interval lists, and the multi-rank tile sharding
(gloo, world_size 2) reproduces the single-process stitched result bit for bit."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_npz


def _header_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(bfm_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    from brainfm_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from brainfm_amd import build
        build.build(verbose=False)
    lib = L.load()
    declared = _header_symbols()
    assert len(declared) >= 20
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    unbound = [n for n in declared if n not in L.SIGNATURES]
    assert not unbound, "declared in the header but not bound in _lib.py: %s" % unbound
    assert b"gfx950" in lib.bfm_version()


def test_product_path_has_no_cpu_fallback():
    from brainfm_amd import _lib as L, test_utils as TU
    from brainfm_amd.engine import UNetEngine
    with pytest.raises(L.BfmError):
        UNetEngine({}, device="cpu")
    with pytest.raises(L.BfmError):
        TU.evaluate_image(torch.zeros(1, 1, 8, 8, 8), None, device="cpu")


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "brainfm_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_tiling_matches_reference_golden():
    from brainfm_amd import test_utils as TU
    d = load_npz("tiling_ranges.npz")
    for n in (160, 200, 256, 512):                              # 512: BASELINE config 4 (216 tiles)
        img = torch.zeros(1, 1, n, n, n)
        lst, cnt = TU.tiling(img, stride=[80] * 3, win_size=[160] * 3)
        assert np.array_equal(np.array([r for _, r in lst]), d["ranges_%d" % n])
        c = cnt.numpy()
        assert np.array_equal(np.bincount(c.astype(np.int64).ravel(), minlength=9), d["cnt_%d_hist" % n])
        assert np.array_equal(c[np.arange(n), np.arange(n), np.arange(n)], d["cnt_%d_diag" % n])
        for t, ((x0, x1), (y0, y1), (z0, z1)) in lst[:3]:
            assert tuple(t.shape[2:]) == (x1 - x0, y1 - y0, z1 - z0)
    assert np.array_equal(np.array(TU.axis_intervals(512, 160, 80)), d["ranges_512_x"])
    assert len(TU.tiling_ranges((256,) * 3, [80] * 3, [160] * 3)) == 27
    assert len(TU.tiling_ranges((512,) * 3, [80] * 3, [160] * 3)) == 216
    # ragged / tiny volumes
    assert TU.axis_intervals(100, 160, 80) == [(0, 100)]
    assert TU.axis_intervals(161, 160, 80) == [(0, 160), (81, 161)]


def test_zero_crop_and_center_crop():
    from brainfm_amd import test_utils as TU
    v = torch.zeros(10, 12, 14)
    v[2:7, 3:9, 4:5] = 1
    assert tuple(TU.zero_crop(v).shape) == (5, 6, 1)
    img, start, shp, aff = TU.center_crop(torch.rand(30, 20, 10), [16, 16, 16], aff=np.eye(4))
    assert tuple(img.shape) == (1, 1, 16, 16, 10) and start == [7, 2, 0] and aff[0, 3] == 7


def test_lpt_assignment_balances_tiles():
    from brainfm_amd import test_utils as TU
    ranges = TU.tiling_ranges((256,) * 3, [80] * 3, [160] * 3)
    for world in (1, 2, 4, 8):
        owner = TU.assign_tiles(ranges, world)
        assert sorted(set(owner)) == list(range(world))
        assert sum(TU.tile_cost(r) for r in ranges) == 32768000
        vox = [sum(TU.tile_cost(r) for r, o in zip(ranges, owner) if o == k) for k in range(world)]
        assert max(vox) == min(vox) == 32768000 // world     # the 8 shape groups carry 4 096 000 tile voxels each
        # tiles of one shape stay together as far as the balance allows: at 8 ranks every rank holds ONE shape group
        # (its deep levels run as one batch, the weights are read once per rank)
        shapes_of = [{tuple(b - a for a, b in r) for r, o in zip(ranges, owner) if o == k} for k in range(world)]
        assert all(len(s_) == 8 // world for s_ in shapes_of), shapes_of
    assert TU.assign_tiles(ranges, 8) == TU.assign_tiles(ranges, 8)
    # batches: same shape, at most BFM_GROUP_MAX tiles, every tile exactly once, largest work first
    b = TU.tile_batches(ranges)
    assert sorted(i for x in b for i in x) == list(range(27)) and [len(x) for x in b] == [8, 4, 4, 4, 2, 2, 2, 1]
    big = TU.tiling_ranges((512,) * 3, [80] * 3, [160] * 3)
    bb = TU.tile_batches(big)
    assert sorted(i for x in bb for i in x) == list(range(216)) and max(len(x) for x in bb) <= TU.GROUP_MAX
    assert all(len({tuple(q - p for p, q in big[i]) for i in x}) == 1 for x in bb)
    own = TU.assign_tiles(big, 8)
    cost = [sum(TU.tile_time(r) for r, o in zip(big, own) if o == k) for k in range(8)]
    assert max(cost) <= 1.25 * sum(cost) / 8, cost


def test_default_args_and_process_args():
    from brainfm_amd import test_utils as TU, models as M
    ga, ta = TU.default_inference_args()
    ga, ta = M.process_args(ga, ta, ga.task)
    assert list(ta.out_channels.items()) == [("T1", 1), ("T2", 1), ("FLAIR", 1), ("CT", 1), ("bias_field_log", 1),
                                             ("segmentation", 56), ("distance", 4), ("registration", 3),
                                             ("high_res_residual", 1)]
    ga, ta = TU.default_inference_args(left_hemis_only=True)
    ga, ta = M.process_args(ga, ta, ga.task)
    assert ta.out_channels["segmentation"] == 18 and ta.out_channels["distance"] == 2


def test_state_dict_names_match_reference():
    """The parameter tree must carry the reference's 84 state-dict names."""
    from brainfm_amd import test_utils as TU, models as M
    d = load_npz("infer_small.npz")
    ga, ta = TU.default_inference_args(f_maps=8, num_levels=4)
    _, _, model, _, _, _ = M.build_model(ga, ta, "cpu")
    ref_keys = sorted(k[3:] for k in d if k.startswith("sd/"))
    assert sorted(model.state_dict().keys()) == ref_keys
    sd = {"module." + k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}   # DDP-style prefix
    M.load_state_dict_by_suffix(model, sd)
    k = "backbone.encoders.1.basic_module.SingleConv2.conv.weight"
    assert torch.equal(model.state_dict()[k], sd["module." + k])


# ----------------------------------------------------------------------------- gloo, world_size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostOps:
    """Stand-in per-tile evaluator and accumulator so the sharding / gather / ordering logic can run
    without a GPU: the 'network' is a cheap deterministic function of the tile."""
    keys = ["T1", "regx", "label"]

    def run_tile(self, im):
        x = im.reshape(-1).to(torch.float32)
        m = (x != 0).float()
        rows = [torch.sin(x * 3) * m * 1.37, (x * x + 0.1) * m, torch.floor(x * 50) * m]
        return self.keys, torch.stack(rows, 0)

    def add(self, acc, rows_j, rng, shape):
        (x0, x1), (y0, y1), (z0, z1) = rng
        acc[x0:x1, y0:y1, z0:z1] += rows_j.reshape(x1 - x0, y1 - y0, z1 - z0)

    def finalize(self, acc, cnt):
        acc /= cnt


def _worker(rank, world, port, q, shape=(40, 36, 44)):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from brainfm_amd import test_utils as TU
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(0)
    full = torch.rand(1, 1, *shape)
    full[:, :, :6] = 0
    acc, ranges, cnt = TU.tiled_inference_distributed(full, None, [12] * 3, [24] * 3, ops=_HostOps())   # round-wise gathers
    acc1, _, _ = TU.tiled_inference_distributed(full, None, [12] * 3, [24] * 3, ops=_HostOps(), rounds=False)
    # only rank 0 holds the volume (the reference reads one file in one process): it travels by broadcast inside the call,
    # with the shape known to the peers and with the 5-number header in front of it
    st = {}
    acc2, _, _ = TU.tiled_inference_distributed(full if rank == 0 else None, None, [12] * 3, [24] * 3, ops=_HostOps(),
                                                shape=shape, stats=st, broadcast=True)
    acc3, _, _ = TU.tiled_inference_distributed(full if rank == 0 else None, None, [12] * 3, [24] * 3, ops=_HostOps(),
                                                broadcast=True)
    assert st["world"] == world and st["broadcast_bytes"] == 4 * full.numel() and len(st["round_bytes_per_peer"]) == st["rounds"]
    assert st["bytes_sent_per_peer"] == sum(st["round_bytes_per_peer"]) > 0
    if rank == 0:
        for k in acc:
            assert torch.equal(acc[k], acc1[k]), k            # one gather at the end gives the same bits
            assert torch.equal(acc[k], acc2[k]) and torch.equal(acc[k], acc3[k]), k
        q.put({k: v.numpy() for k, v in acc.items()})
    else:
        assert acc2 is None and acc3 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (40, 36, 44)), (3, (40, 36, 44)), (3, (20, 22, 24))])
def test_distributed_tiling_two_ranks_bitwise_equals_single(world, shape):
    """world 3: the ranks hold different numbers of tiles (padding rounds), rank 0's own tiles stay local; a volume of
    a single tile leaves the other ranks without work (bench.py --size 160 --gpus N)."""
    import torch.multiprocessing as mp
    from brainfm_amd import test_utils as TU
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, shape)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process result with the same ops, reference tile order
    torch.manual_seed(0)
    full = torch.rand(1, 1, *shape)
    full[:, :, :6] = 0
    ops = _HostOps()
    shape = tuple(full.shape[2:])
    ranges = TU.tiling_ranges(shape, [12] * 3, [24] * 3)
    cnt = TU.count_volume(shape, ranges, "cpu")
    acc = {k: torch.zeros(shape) for k in ops.keys}
    for rng in ranges:
        (x0, x1), (y0, y1), (z0, z1) = rng
        _, rows = ops.run_tile(full[:, :, x0:x1, y0:y1, z0:z1])
        for j, k in enumerate(ops.keys):
            ops.add(acc[k], rows[j], rng, shape)
    for k in ops.keys:
        acc[k] /= cnt
        assert np.array_equal(res[k], acc[k].numpy()), k


def _worker_real_lists(rank, world, port, q, shape, stride, win):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from brainfm_amd import test_utils as TU
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    torch.manual_seed(1)
    full = torch.rand(1, 1, *shape)
    full[:, :, :, : shape[1] // 5] = 0                         # background: tiles that keep nothing, tiles that keep a part
    st = {}
    acc, ranges, cnt = TU.tiled_inference_distributed(full if rank == 0 else None, None, [stride] * 3, [win] * 3, ops=_HostOps(),
                                                      shape=shape, stats=st, broadcast=True)
    if rank == 0:
        q.put(({k: v.numpy() for k, v in acc.items()}, len(ranges), st["rounds"], st["world"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shape,ntiles,classes", [((32, 32, 32), 27, {8000: 1, 4000: 6, 2000: 12, 1000: 8}),
                                                  ((64, 64, 64), 216, {8000: 1, 4000: 15, 2000: 75, 1000: 125})])
def test_distributed_tiling_eight_ranks_with_the_reference_tile_lists(shape, ntiles, classes):
    """VERDICT r5 #5c: world size 8 (what SCALE runs) on the tile LISTS of the two headline volumes -- 256^3 -> 27 tiles
    (1 / 6 / 12 / 8 of the four shapes) and 512^3 -> 216 tiles (1 / 15 / 75 / 125), scripts/demo_test.py:79-119 with stride 80
    and window 160 -- scaled by 1/8 per axis (stride 10, window 20) so that eight CPU processes finish in seconds: same
    interval structure ((0,20),(20,30),(22,32) for 32 as (0,160),(160,240),(176,256) for 256), same number of tiles per shape
    class, so the same LPT assignment by shape group, the same padding rounds and the same order of accumulation on rank 0.
    Only rank 0 holds the volume.  The stitched keys must equal the single-process loop in reference tile order bit for bit."""
    import torch.multiprocessing as mp
    from brainfm_amd import test_utils as TU
    ranges = TU.tiling_ranges(shape, [10] * 3, [20] * 3)
    assert len(ranges) == ntiles
    got = {}
    for r in ranges:
        v = int(np.prod([b - a for a, b in r]))
        got[v] = got.get(v, 0) + 1
    assert got == classes
    big = tuple(8 * v for v in shape)
    assert [[(8 * a, 8 * b) for a, b in r] for r in ranges] == TU.tiling_ranges(big, [80] * 3, [160] * 3)
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real_lists, args=(r, world, port, q, shape, 10, 20)) for r in range(world)]
    for p in procs:
        p.start()
    res, n, rounds, w = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert n == ntiles and w == world and rounds >= 1
    torch.manual_seed(1)
    full = torch.rand(1, 1, *shape)
    full[:, :, :, : shape[1] // 5] = 0
    ops = _HostOps()
    cnt = TU.count_volume(shape, ranges, "cpu")
    acc = {k: torch.zeros(shape) for k in ops.keys}
    for rng in ranges:
        (x0, x1), (y0, y1), (z0, z1) = rng
        _, rows = ops.run_tile(full[:, :, x0:x1, y0:y1, z0:z1])
        for j, k in enumerate(ops.keys):
            ops.add(acc[k], rows[j], rng, shape)
    for k in ops.keys:
        acc[k] /= cnt
        assert np.array_equal(res[k], acc[k].numpy()), k


def test_prepare_image_host_plans_match_reference():
    """Host halves of the pre-processing chain against vectors from the reference's torch_resize /
    align_volume_to_ref: target size, aligned affine, axis permutation + flips reproduced with numpy on the golden
    resized volume, and the centre-crop origin."""
    from brainfm_amd import misc as MI
    from brainfm_amd import test_utils as TU
    d = load_npz("prep_image.npz")
    for name in ("A", "B", "C"):
        vol, aff = d[name + "/vol"], d[name + "/aff"]
        newsize, sigmas = MI.resize_plan(vol.shape, aff, 1.0)
        assert tuple(newsize) == d[name + "/resized"].shape
        if name == "C":
            assert (sigmas > 0).all()                        # 0.7/0.8 mm -> 1 mm is a down-sampling: pre-blur on
        perm, flip, aff_al = MI.align_plan(d[name + "/resized"].shape, d[name + "/aff_resized"], np.eye(4))
        assert np.allclose(aff_al, d[name + "/aff_aligned"], rtol=0, atol=1e-12)
        al = np.transpose(d[name + "/resized"], perm)
        for a in range(3):
            if flip[a]:
                al = np.flip(al, a)
        assert np.array_equal(al, d[name + "/aligned"])
    # anisotropic zoom tables: lengths and clamping
    f, c, wf, wc = MI.aniso_zoom_tables(22, 44)
    assert len(f) == 44 and f.min() == 0 and c.max() == 21 and np.allclose(wf + wc, 1)
    # centre crop bookkeeping (indexing only, CPU tensors are fine here)
    al = torch.from_numpy(d["A/aligned"].copy())
    # (prepare_image's returned crop_start / affine belong to its LAST crop, of the un-aligned `final`, and the
    # affine object is shifted in place by every crop: reference quirk, reproduced end to end in the GPU test)
    crop, start, shp, aff_out = TU.center_crop(al, [32, 32, 32], aff=d["A/aff_aligned"].copy())
    assert list(start) == [(al.shape[i] - 32) // 2 for i in range(3)] and tuple(shp) == tuple(al.shape)
    assert np.array_equal(crop[0, 0].numpy(), d["A/orig"][0, 0])


def test_volume_io_nifti_mgh_roundtrip_and_format_fields(tmp_path):
    """brainfm_amd.volio against the file-format definitions: NIfTI-1 / MGH round trips, header fields a
    third-party reader relies on, a hand-built big-endian qform-only NIfTI with intensity scaling, cropped reads."""
    import gzip
    import struct
    from brainfm_amd import volio as V
    rng = np.random.RandomState(3)
    aff = np.array([[0., -1.2, 0., 30.], [1.0, 0., 0., -20.], [0., 0., -2.0, 5.], [0., 0., 0., 1.]])
    for ext, dt in ((".nii", np.float32), (".nii.gz", np.int16), (".nii", np.uint8), (".mgz", np.float32), (".mgz", np.int16)):
        vol = (rng.rand(7, 5, 9) * 100).astype(dt)
        f = str(tmp_path / ("v_%s%s" % (np.dtype(dt).name, ext)))
        V.MRIwrite(vol, aff, f)
        got, a2 = V.MRIread(f)
        assert got.dtype == np.float64 and np.array_equal(got, vol.astype(np.float64))
        assert np.allclose(a2, aff, atol=1e-5)
        assert np.array_equal(V.MRIread(f, dtype="int", im_only=True), vol.astype(np.float64).astype("int"))
        img = V.load(f)
        assert np.array_equal(img.dataobj[1:6, 2:4, 3:8], got[1:6, 2:4, 3:8])
    # header fields of a written NIfTI-1
    f = str(tmp_path / "fields.nii")
    V.MRIwrite(np.zeros((4, 3, 2), np.float32), aff, f)
    raw = open(f, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\x00"
    assert struct.unpack("<8h", raw[40:56])[:4] == (3, 4, 3, 2)
    assert struct.unpack("<2h", raw[70:74]) == (16, 32) and struct.unpack("<f", raw[108:112])[0] == 352.0
    assert struct.unpack("<2h", raw[252:256]) == (0, 2)
    assert np.allclose(struct.unpack("<12f", raw[280:328]), aff[:3].reshape(-1))
    assert np.allclose(struct.unpack("<8f", raw[76:108])[1:4], [1.0, 1.2, 2.0]) and len(raw) == 352 + 4 * 24
    # hand-built big-endian, qform only: 90 degrees about z (b,c,d = 0,0,sin45), zooms (1,2,3), qfac -1, y = 2*raw - 1
    hdr = bytearray(348)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">2h", hdr, 70, 4, 16)
    struct.pack_into(">8f", hdr, 76, -1.0, 1.0, 2.0, 3.0, 1, 1, 1, 1)
    struct.pack_into(">f", hdr, 108, 352.0)
    struct.pack_into(">2f", hdr, 112, 2.0, -1.0)
    struct.pack_into(">2h", hdr, 252, 1, 0)
    struct.pack_into(">6f", hdr, 256, 0.0, 0.0, np.sqrt(0.5), 10.0, 20.0, 30.0)
    hdr[344:348] = b"n+1\x00"
    data = np.arange(24, dtype=">i2")
    f = str(tmp_path / "be.nii.gz")
    with gzip.open(f, "wb") as fh:
        fh.write(bytes(hdr) + b"\x00" * 4 + data.tobytes())
    got, a2 = V.MRIread(f)
    assert got.shape == (2, 3, 4) and np.array_equal(got, 2.0 * np.arange(24).reshape((2, 3, 4), order="F") - 1.0)
    expect = np.array([[0., -2., 0., 10.], [1., 0., 0., 20.], [0., 0., -3., 30.], [0., 0., 0., 1.]])
    assert np.allclose(a2, expect, atol=1e-6)
    # MGH without geometry: FreeSurfer's default LIA orientation, centre at the origin
    hdr = bytearray(284)
    struct.pack_into(">7i", hdr, 0, 1, 2, 2, 2, 1, 0, 0)
    f = str(tmp_path / "plain.mgz")
    with gzip.open(f, "wb") as fh:
        fh.write(bytes(hdr) + bytes(range(8)))
    got, a2 = V.MRIread(f)
    assert np.array_equal(got.reshape(-1, order="F"), np.arange(8.0))
    assert np.allclose(a2, [[-1, 0, 0, 1], [0, 0, 1, -1], [0, -1, 0, 1], [0, 0, 0, 1]])


def _allreduce_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from brainfm_amd import train as TR
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    grads = {"a": torch.randn(3, 5, generator=g), "b": torch.randn(7, generator=g), "c": torch.randn(2, 2, 2, generator=g)}
    TR.allreduce_mean_(grads)
    if rank == 0:
        q.put({k: v.numpy() for k, v in grads.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_is_the_mean():
    """N2 multi-GPU: the flat-bucket gradient all-reduce (DDP's averaging) over 2 gloo ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = {}
    for rank in range(2):
        g = torch.Generator().manual_seed(100 + rank)
        for k, shp in (("a", (3, 5)), ("b", (7,)), ("c", (2, 2, 2))):
            exp[k] = exp.get(k, 0) + torch.randn(*shp, generator=g) / 2
    for k in exp:
        assert np.allclose(res[k], exp[k].numpy(), atol=1e-7), k


def _gradstore_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from brainfm_amd import train as TR
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    named = [("head.weight_all", (5, 4)), ("head.bias_all", (5,)), ("d1.conv.weight", (4, 3, 3, 3, 3)), ("d1.groupnorm.weight", (3,)),
             ("d1.groupnorm.bias", (3,)), ("e0.conv.weight", (6, 2, 3, 3, 3)), ("e0.groupnorm.weight", (2,)), ("e0.groupnorm.bias", (2,))]
    st = TR.GradStore(named, "cpu", n_buckets=3)
    launched = []
    for it in range(2):                                          # the buffer is persistent: a second iteration reuses it
        st.begin()
        g = torch.Generator().manual_seed(100 * it + rank)
        part = None
        if it == 1:                                              # an earlier sample's gradients, added in front of `done`
            part = {"head.final_conv_a.weight": torch.full((2, 4), 1.0), "head.final_conv_b.weight": torch.full((3, 4), 2.0),
                    "head.final_conv_a.bias": torch.full((2,), 3.0), "head.final_conv_b.bias": torch.full((3,), 4.0)}
            part.update({n: torch.full(s, 0.5) for n, s in named[2:]})
        sink = TR._Sink(st, part, {"a": (0, 2), "b": (2, 3)})
        for n, shp in named:
            sink.out(n, shp).copy_(torch.randn(*shp, generator=g))
            sink.done(n)
            launched.append(sum(st.launched))
        st.finish()
        res = {n: v.clone().numpy() for n, v in st.views.items()}
        if rank == 0:
            q.put((res, list(st.first), list(launched), st.total, st.world))
        launched = []
    dist.barrier()
    dist.destroy_process_group()


def _gradstore_worker8(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from brainfm_amd import train as TR
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    # the slot list of a 3-level, 8-wide net as TrainStep.grad_store() lays it out: heads, decoders last to first, encoders
    named = [("head.weight_all", (12, 8)), ("head.bias_all", (12,))]
    for name, cin, cout in (("dec1.2", 8, 8), ("dec1.1", 24, 8), ("dec0.2", 16, 16), ("dec0.1", 48, 16), ("enc2.2", 16, 32), ("enc2.1", 16, 16),
                            ("enc1.2", 8, 16), ("enc1.1", 8, 8), ("enc0.2", 4, 8), ("enc0.1", 1, 4)):
        named += [(name + ".conv.weight", (cout, cin, 3, 3, 3)), (name + ".groupnorm.weight", (cin,)), (name + ".groupnorm.bias", (cin,))]
    st = TR.GradStore(named, "cpu", n_buckets=6)
    st.begin()
    g = torch.Generator().manual_seed(7 + rank)
    sink = TR._Sink(st, None, {})
    order = []
    for n, shp in named:
        sink.out(n, shp).copy_(torch.randn(*shp, generator=g))
        sink.done(n)
        order.append(sum(st.launched))
    st.finish()
    if rank == 0:
        q.put(({n: v.clone().numpy() for n, v in st.views.items()}, [n for n, _ in named], [list(s_) for _, s_ in named],
               len(st.range), order, st.world))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_store_eight_ranks_six_buckets():
    """The training side of SCALE's N = 8: eight gloo ranks, the slot list of a whole (small) backbone in
    backward order, six buckets going out one after the other while later slots are still being written; every slot ends up
    holding the sum of the eight ranks' gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gradstore_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res, names, shapes, nb, order, world = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert world == 8 and nb == 6 and order == sorted(order) and order[-1] == 6 and order[0] == 0
    exp = {}
    for rank in range(8):
        g = torch.Generator().manual_seed(7 + rank)
        for n, shp in zip(names, shapes):
            exp[n] = exp.get(n, 0) + torch.randn(*shp, generator=g)
    for n in names:
        assert np.allclose(res[n], exp[n].numpy(), atol=2e-6), n


def test_grad_store_buckets_sum_over_two_ranks_in_place():
    """VERDICT r5 #2 / #5b: train.GradStore -- one persistent flat buffer, slots in backward order, a few buckets, each
    all-reduced as soon as its last slot is complete (async, while later slots are still being written), no concatenation and
    no copy back.  Two gloo ranks: every slot ends up holding the SUM over the ranks (the mean's division goes into the
    optimiser's gradient scale), buckets start in order as their last tensor completes, slots are 16-byte aligned, and an
    earlier sample's partial sums are added before the bucket goes out."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gradstore_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120), q.get(timeout=120)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    named = [("head.weight_all", (5, 4)), ("head.bias_all", (5,)), ("d1.conv.weight", (4, 3, 3, 3, 3)), ("d1.groupnorm.weight", (3,)),
             ("d1.groupnorm.bias", (3,)), ("e0.conv.weight", (6, 2, 3, 3, 3)), ("e0.groupnorm.weight", (2,)), ("e0.groupnorm.bias", (2,))]
    for it, (res, first, launched, total, world) in enumerate(outs):
        assert world == 2 and total % 4 == 0 and first[0] == 0 and first[-1] == len(named) and len(first) == 4
        assert launched == sorted(launched) and launched[-1] == 3 and launched[0] == 0      # buckets go out one by one, in order
        exp = {}
        for rank in range(2):
            g = torch.Generator().manual_seed(100 * it + rank)
            for n, shp in named:
                v = torch.randn(*shp, generator=g)
                if it == 1:
                    if n == "head.weight_all":
                        v = v + torch.cat([torch.full((2, 4), 1.0), torch.full((3, 4), 2.0)])
                    elif n == "head.bias_all":
                        v = v + torch.cat([torch.full((2,), 3.0), torch.full((3,), 4.0)])
                    else:
                        v = v + 0.5
                exp[n] = exp.get(n, 0) + v
        for n, _ in named:
            assert np.allclose(res[n], exp[n].numpy(), atol=1e-6), (it, n)


def test_loss_scaler_and_cosine_schedule_host_logic():
    from brainfm_amd import train as TR
    s = TR.LossScaler(init_scale=8.0, growth_interval=2)
    s.update(False)
    assert s.scale == 8.0
    s.update(False)
    assert s.scale == 16.0
    s.update(True)
    assert s.scale == 8.0
    s.update(False)
    s.update(True)                                  # an inf resets the clean-step counter
    s.update(False)
    assert s.scale == 4.0
    off = TR.LossScaler(enabled=False)
    off.update(True)
    assert off.scale == 1.0
    # utils/misc.py:1265-1276 on a hand-computed case: 2 epochs x 4 iterations, 1 warm-up epoch
    sch = TR.cosine_scheduler(1.0, 0.0, 2, 4, warmup_epochs=1)
    assert len(sch) == 8
    assert np.allclose(sch[:4], [0.0, 1 / 3, 2 / 3, 1.0])
    assert np.allclose(sch[4:], 0.5 * (1 + np.cos(np.pi * np.arange(4) / 4)))
    # utils/misc.py:1251-1262 by hand: 4 epochs x 2 iterations, 1 warm-up epoch, drop at "epoch 1" of the post-warm-up array
    ms = TR.multistep_scheduler(1.0, [1], 4, 2, warmup_epochs=1, gamma=0.1)
    assert np.allclose(ms, [0.0, 1.0, 1.0, 1.0, 0.1, 0.1, 0.1, 0.1])
    import pytest
    from brainfm_amd import _lib as L
    with pytest.raises(L.BfmError):
        TR.TrainStep(None, None, ["contrastive"], {}, [1.0], 1)


def test_volio_reads_the_reference_atlas_file():
    """SURVEY N3 pinned to the one real volume file on the reference's hot path, files/gca.mgz (utils/test_utils.py:38-43
    reads it at import: MNI, aff2 = MRIread(atlas_path); A = inv(aff2)).  Expected values: tests/golden/volio_gca.npz, an
    independent parse of the MGH format definition (make_golden_volio.py; nibabel is absent, so volio stays 'parity
    unpinned against nibabel').  The file itself never leaves the reference tree: skipped where that is absent."""
    import gzip
    import hashlib
    from brainfm_amd import volio
    from brainfm_amd import test_utils as TU
    path = os.path.join(os.environ.get("BRAINFM_REFERENCE", "/root/reference"), "files", "gca.mgz")
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    d = load_npz("volio_gca.npz")
    v = volio.load(path)
    assert tuple(v.shape) == tuple(d["dims"][:3]) == (256, 256, 256)
    assert np.array_equal(np.asarray(v.affine), d["affine"])
    # the conformed-space vox2ras every FreeSurfer atlas carries
    assert np.array_equal(d["affine"], np.array([[-1., 0, 0, 128], [0, 0, 1, -128], [0, -1, 0, 128], [0, 0, 0, 1]]))
    im, aff = volio.MRIread(path)                                # the reference's reader signature (utils/misc.py:194-208)
    assert im.shape == (256, 256, 256) and np.array_equal(aff, d["affine"])
    assert np.array_equal(im[::8, ::8, ::8].astype(np.float32), d["sub8"])
    assert float(im.astype(np.float64).sum()) == float(d["sum"]) and float(im.max()) == float(d["max"])
    # byte level: the payload volio decoded, re-encoded big endian x-fastest, is the file's payload
    be = np.asfortranarray(im.astype(">f4")).tobytes(order="F")
    assert np.array_equal(np.frombuffer(hashlib.sha256(be).digest(), dtype=np.uint8), d["sha256_be_payload"])
    assert np.array_equal(np.frombuffer(gzip.open(path, "rb").read(284), dtype=np.uint8), d["header"])
    # a round trip through volio's own MGZ writer reproduces header fields and voxels
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "copy.mgz")
        volio.MRIwrite(im.astype(np.float32), aff, out)
        raw = gzip.open(out, "rb").read()
        assert raw[:28] == d["header"][:28].tobytes() or raw[:20] == d["header"][:20].tobytes()
        assert raw[28:90] == d["header"][28:90].tobytes()        # goodRAS, spacing, direction cosines, c_ras
        im2, aff2 = volio.MRIread(out)
        assert np.array_equal(im2, im) and np.array_equal(aff2, aff)
    # what get_deformed_atlas needs from it
    MNI, A = TU.load_atlas(path)
    assert MNI.dtype == np.float32 and MNI.shape == (256, 256, 256)
    assert np.array_equal(A, np.linalg.inv(d["affine"]).astype(np.float32))
    TU.MNI, TU.A, TU.atlas_path = None, None, None


def test_bench_gpus_flag_launches_ranks_or_refuses():
    """`python bench.py --gpus N` without a launcher must start N ranks itself -- or fail loudly when the box has fewer
    than N devices -- instead of silently running one rank and printing n_gpus: 1 (round-1 defect).  No GPU here: the
    refusal path is what can run; the spawn path is exercised on the GPU box (BFM_BENCH_SHARE_GPU=1 dry runs)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BFM_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the spawn path would run the benchmark")
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "--gpus 2 but only" in r.stderr and '"metric"' not in r.stdout
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "launch_ranks(args.gpus" in src and "os.exec" not in src          # children are spawned, nothing is re-executed


def test_python_boundary_signatures_match_the_reference():
    """the mirrors keep the reference's Python call surface.  tests/golden/api_signatures.json holds
    inspect.signature of every boundary callable of the REAL reference (make_golden_signatures.py, build container);
    each mirror must have the same parameter names in the same order with the same defaults.  A mirror may only add
    parameters that have defaults, and only those the fixture lists under allowed_extra (recorded deviations)."""
    import inspect
    import json
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_signatures.json")))
    assert not d["unresolved_in_reference"], d["unresolved_in_reference"]
    assert len(d["signatures"]) >= 55

    def resolve(spec):
        mod, qual = spec.split(":")
        obj = __import__(mod, fromlist=["_"])
        for part in qual.split("."):
            obj = getattr(obj, part)
        return obj

    problems = []
    for ref, e in sorted(d["signatures"].items()):
        fn = resolve(e["mirror"])
        mine = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else
                 ("<function %s>" % p.default.__name__ if inspect.isfunction(p.default) else repr(p.default))]
                for p in inspect.signature(fn).parameters.values()]
        want = e["params"]
        extra = d["allowed_extra"].get(e["mirror"], {})
        if mine[:len(want)] != want:
            problems.append((e["mirror"], "reference %s" % want, "mirror %s" % mine))
            continue
        for name, kind, default in mine[len(want):]:
            if name not in extra or default is None:
                problems.append((e["mirror"], "extra parameter %s (default %s) is not a recorded deviation" % (name, default)))
    assert not problems, "\n".join(str(p) for p in problems)


def test_reference_format_checkpoints_load_without_running_their_pickle(tmp_path, monkeypatch):
    """scripts/train.py:206-214 pickles utils.config.Config objects (dict subclasses of the reference's own module)
    beside 'model' and 'optimizer'.  read_checkpoint_file must return the tensors of such a file (a) without the
    reference's modules being importable, (b) without BFM_TRUST_CHECKPOINT, and (c) without executing anything the
    file asks for: a pickled object whose __reduce__ would run code comes back as an inert dict."""
    import subprocess
    import sys
    import torch
    from brainfm_amd import models as M
    path = str(tmp_path / "ckp.pth")
    marker = str(tmp_path / "executed")
    # written by a child process in which the classes live in a module `utils.config` that this process never has
    writer = '''
import sys, types, torch, os
m = types.ModuleType("utils"); c = types.ModuleType("utils.config"); m.config = c
sys.modules["utils"] = m; sys.modules["utils.config"] = c
class AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); self.__dict__ = self
class Config(AttrDict):
    pass
class Evil:
    def __reduce__(self):
        return (os.system, ("touch %s",))
for k in (AttrDict, Config, Evil):
    k.__module__ = "utils.config"; setattr(c, k.__name__, k)
w = torch.arange(12, dtype=torch.float32).reshape(3, 4)
torch.save({"model": {"module.backbone.w": w, "head.b": torch.ones(2)}, "optimizer": {"state": {0: {"step": torch.tensor(3.)}}},
            "epoch": 7, "gen_args": Config(a=1, nested=AttrDict(b=[1, 2])), "train_args": Config(lr=0.1), "evil": Evil()}, %r)
''' % (marker, path)
    subprocess.check_call([sys.executable, "-c", writer])
    assert "utils.config" not in sys.modules
    monkeypatch.delenv("BFM_TRUST_CHECKPOINT", raising=False)
    ckp = M.read_checkpoint_file(path)
    assert not os.path.exists(marker), "the checkpoint's pickle was executed"
    assert torch.equal(ckp["model"]["module.backbone.w"], torch.arange(12, dtype=torch.float32).reshape(3, 4))
    assert torch.equal(ckp["model"]["head.b"], torch.ones(2)) and ckp["epoch"] == 7
    assert float(ckp["optimizer"]["state"][0]["step"]) == 3.0
    assert isinstance(ckp["gen_args"], M.InertObject) and ckp["gen_args"]["a"] == 1 and ckp["gen_args"].nested["b"] == [1, 2]
    assert isinstance(ckp["evil"], M.InertObject)
    # a plain tensor checkpoint still takes the weights_only path
    plain = str(tmp_path / "plain.pth")
    torch.save({"model": {"w": torch.zeros(2)}}, plain)
    assert torch.equal(M.read_checkpoint_file(plain)["model"]["w"], torch.zeros(2))
    # a weight pickled through a class the restricted reader does not rebuild must be REPORTED, not handed on as
    # an empty dict that fails later as a missing key; and the restricted path says that it was taken
    odd = str(tmp_path / "odd.pth")
    writer2 = '''
import sys, types, torch
m = types.ModuleType("theirs"); sys.modules["theirs"] = m
class Wrapped:
    def __init__(self, t): self.t = t
Wrapped.__module__ = "theirs"; m.Wrapped = Wrapped
torch.save({"model": {"w": Wrapped(torch.ones(3))}, "epoch": 1}, %r)
''' % odd
    subprocess.check_call([sys.executable, "-c", writer2])
    import pytest
    with pytest.warns(UserWarning, match="restricted unpickler"):
        with pytest.raises(M.BfmCheckpointError, match="model\\['w'\\]"):
            M.read_checkpoint_file(odd)


def _device_code_objects(lib_path):
    """The gfx950 code objects inside a HIP shared library: the .hip_fatbin section is a sequence of clang offload bundles
    (magic, entry count, then (offset, size, triple) records), one per translation unit."""
    import struct
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib_path])
        d = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        for m in re.finditer(magic, d):
            p = m.start()
            n = struct.unpack_from("<Q", d, p + 24)[0]
            off = p + 32
            for _ in range(n):
                o, sz, tl = struct.unpack_from("<QQQ", d, off)
                off += 24
                triple = d[off:off + tl].decode()
                off += tl
                if "gfx950" in triple:
                    co = os.path.join(tmp, "co.o")
                    open(co, "wb").write(d[p + o:p + o + sz])
                    yield subprocess.run([llvm + "/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def test_no_packed_fp32_instruction_in_the_code_objects():
    """profiles/r06_hazard_root_cause.txt: the low half of a packed-FP32 multiply (v_pk_mul_f32 with bank-conflicting sources)
    is lost in lanes 48..63 when conv_wino4 / conv_wino4d's MFMA tap loop shares the compute unit -- the cause of every
    "gather beside a convolution" wrong result of rounds 2-5.  The library is therefore built without that instruction
    class (brainfm_amd/build.py: NO_PACKED_FP32).  This checks what was BUILT, not the flags: the disassembly of every gfx950
    code object in libbrainfm_hip.so holds no v_pk_*_f32 (v_pk_mul / add / fma / mov _f32)."""
    from brainfm_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from brainfm_amd import build
        build.build(verbose=False)
    nobj = nkern = 0
    for dis in _device_code_objects(L.LIB_PATH):
        nobj += 1
        nkern += len(re.findall(r"^[0-9a-f]+ <\S+>:", dis, re.M))
        found = re.findall(r"\bv_pk_\w+_f32\b.*", dis)
        assert not found, found[:3]
        assert "v_mfma_" in dis or "s_endpgm" in dis              # it IS a disassembly
    assert nobj >= 15 and nkern >= 200, (nobj, nkern)


def test_m0_is_written_only_for_the_lds_dma_of_conv_wino4d():
    """conv_wino4d's inline assembly writes M0 for global_load_lds_dwordx4 without being able to declare it (M0 is
    reserved; naming it as a clobber is a warning, not a constraint).  That is safe only while nothing else in those kernels
    holds a value in M0.  Checked on the built code: in every conv_wino4d* kernel each instruction that mentions m0 is an
    `s_mov_b32 m0, sN` whose next instructions are `s_nop 0` and the DMA, every DMA has that pair in front of it, and no
    other reader of M0 (movrel, readlane by m0, sendmsg, ds_gws, interp) occurs."""
    from brainfm_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        from brainfm_amd import build
        build.build(verbose=False)
    nk = ndma = 0
    for dis in _device_code_objects(L.LIB_PATH):
        for m in re.finditer(r"^[0-9a-f]+ <(\S*conv_wino4d\S*)>:\n(.*?)(?=^\n|\Z)", dis, re.M | re.S):
            body = [l.split("//")[0].strip() for l in m.group(2).splitlines() if l.strip()]
            if not any("global_load_lds" in l for l in body):
                continue
            nk += 1
            for i, l in enumerate(body):
                if "global_load_lds" in l:
                    ndma += 1
                    assert body[i - 1].startswith("s_nop 0") and re.match(r"s_mov_b32 m0, s\d+", body[i - 2]), (m.group(1), body[i - 3:i + 1])
                elif re.search(r"\bm0\b", l):
                    assert re.match(r"s_mov_b32 m0, s\d+", l) and "global_load_lds" in body[i + 2], (m.group(1), l)
                assert not re.match(r"(s_movrel|v_movrel|s_sendmsg|ds_gws|v_interp)", l), (m.group(1), l)
    assert nk >= 4 and ndma >= 4 * 13, (nk, ndma)
