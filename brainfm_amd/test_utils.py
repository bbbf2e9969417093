"""Host-side mirror of the reference's inference harness
(``utils/test_utils.py`` + the tile loop of ``scripts/demo_test.py``).

  tiling(img, stride, win_size, zero_crop_first)      utils/test_utils.py:93-137
  zero_crop / center_crop                             :60-90, :141-188
  evaluate_image(inputs, ckp_path, ...)               :289-312  (model cached: fixes quirk Q1 behind the same signature)
  test_tile -> tiled_inference(...)                   scripts/demo_test.py:66-119, in HBM, no NIfTI round trip
  tiled_inference_distributed(...)                    tiles sharded over ranks, RCCL gather to rank 0

All arithmetic runs in libbrainfm_hip.so (fused tail kernel + stitch kernels).
"""
import ctypes as C
import os
from argparse import Namespace
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import cfg as _cfg
from . import generator_utils as GU
from . import misc as MI
from . import models as M
from .engine import UNetEngine

# default cfg locations, overridable like the reference's module globals (test_utils.py:28-36; quirk Q3:
# the shipped defaults point at files that do not exist, so they are None here and must be given)
default_gen_cfg_file = None
default_train_cfg_file = None
default_val_file = None
gen_cfg_dir = ""
train_cfg_dir = ""

STITCH_KEYS = ["T1", "T2", "FLAIR", "CT", "high_res_residual", "high_res", "bias_field", "lp", "lw", "rp", "rw",
               "fake_cortical", "regx", "regy", "regz", "label", "deformed_atlas"]
# the 17 keys scripts/demo_test.py:107-119 stitches: every output without 'feat' / 'segmentation' in its name, then
# 'deformed_atlas' (outs['deformed_atlas'] = None is appended before the stitch loop, :108)

# atlas the tile loop deforms (utils/test_utils.py:38-43 reads it into module globals at import).  A path here (or
# InferenceSession.set_atlas) switches the 17th key on; the reference's value is 'files/gca.mgz' relative to its tree.
atlas_path = None


# ----------------------------------------------------------------------------- configs without YAML files
def default_inference_args(f_maps=64, num_levels=6, left_hemis_only=False, size=(160, 160, 160), tasks=None,
                           max_surf_distance=3.0, num_groups=8, unit_feat=True, in_channels=1):
    """(gen_args, train_args) carrying the keys build_model reads, with the values of
    cfgs/generator/test/demo_test.yaml + cfgs/trainer/{default_train,test/demo_test}.yaml."""
    if tasks is None:
        tasks = dict(T1=True, T2=True, FLAIR=True, CT=True, segmentation=True, distance=True, bias_field=True,
                     registration=True, super_resolution=True, surface=False, pathology=False, contrastive=False)
    gen_args = Namespace(task=Namespace(**tasks), max_surf_distance=max_surf_distance,
                         generator=Namespace(size=list(size), left_hemis_only=left_hemis_only))
    train_args = Namespace(backbone="unet3d", in_channels=in_channels, f_maps=f_maps, layer_order="gcl",
                           num_groups=num_groups, num_levels=num_levels, unit_feat=unit_feat,
                           task_f_maps=[f_maps if isinstance(f_maps, int) else f_maps[0]],
                           losses=Namespace(uncertainty=None, implicit_pathol=False))
    return gen_args, train_args


# ----------------------------------------------------------------------------- tiling (host logic)
def zero_crop(orig, tol=0, crop_range_lst=None, save_path=None):
    """utils/test_utils.py:60-90.  On the device the bounding box of (orig > tol) comes from one reduction kernel
    instead of torch.argwhere's coordinate list.  `save_path` is accepted and unused, as in the reference."""
    if crop_range_lst is None and orig.is_cuda and orig.dim() == 3:
        src = orig.to(torch.float32).contiguous()
        box = torch.empty(6, dtype=torch.int32, device=orig.device)
        L.check(L.load().bfm_bbox_nonzero(L.ptr(src), src.shape[0], src.shape[1], src.shape[2], float(tol), L.ptr(box),
                                          L.stream_ptr()), "bbox_nonzero")
        x0, y0, z0, x1, y1, z1 = [int(v) for v in box.tolist()]
        if x0 > x1:
            raise RuntimeError("zero_crop: no voxel above tol")      # the reference's argwhere().min() raises here too
    elif crop_range_lst is None:
        coords = torch.argwhere(orig > tol)
        x0, y0, z0 = coords.min(dim=0)[0]
        x1, y1, z1 = coords.max(dim=0)[0] + 1
    else:
        [[x0, y0, z0], [x1, y1, z1]] = crop_range_lst
    return orig[x0:x1, y0:y1, z0:z1]


def axis_intervals(n, win, stride):
    """One axis of `tiling` (test_utils.py:105-124): the first window is `win` wide, each later
    one `stride` wide, the last pulled back to end at n (quirk Q4)."""
    start, end = 0, min(win, n)
    out = [(start, end)]
    while end < n:
        start = min(end, n - stride)
        end = min(start + stride, n)
        out.append((start, end))
    return out


def tiling_ranges(shape, stride, win_size):
    xs, ys, zs = (axis_intervals(shape[a], win_size[a], stride[a]) for a in range(3))
    return [[x, y, z] for x in xs for y in ys for z in zs]


def count_volume(shape, ranges, device):
    """cnt[range] += 1 for every tile (test_utils.py:126-135)."""
    dev = torch.device(device)
    if dev.type == "cuda":
        lib = L.load()
        cnt = torch.zeros(tuple(shape), dtype=torch.float32, device=dev)
        for (x0, x1), (y0, y1), (z0, z1) in ranges:
            L.check(lib.bfm_tile_count_add(L.ptr(cnt), shape[0], shape[1], shape[2], x0, x1, y0, y1, z0, z1,
                                           L.stream_ptr()), "tile_count_add")
        return cnt
    cnt = np.zeros(tuple(shape), dtype=np.float32)          # interval bookkeeping for host-side callers
    for (x0, x1), (y0, y1), (z0, z1) in ranges:
        cnt[x0:x1, y0:y1, z0:z1] += 1
    return torch.from_numpy(cnt)


def tiling(img, stride=[40, 40, 40], win_size=[160, 160, 160], zero_crop_first=False):
    """utils/test_utils.py:93-137: (list of (tile view, [(x0,x1),(y0,y1),(z0,z1)]), cnt)."""
    if zero_crop_first:
        img = zero_crop(img[0, 0])[None, None]
    shape = tuple(img.shape[2:])
    ranges = tiling_ranges(shape, stride, win_size)
    img_list = [(img[:, :, x0:x1, y0:y1, z0:z1], [(x0, x1), (y0, y1), (z0, z1)])
                for (x0, x1), (y0, y1), (z0, z1) in ranges]
    return img_list, count_volume(shape, ranges, img.device)


def center_crop(img, win_size=[220, 220, 220], zero_crop_first=False, aff=np.eye(4)):
    """utils/test_utils.py:141-188 (indexing only)."""
    if img.dim() == 4:
        img = torch.permute(img, (3, 0, 1, 2))[None]
        permuted = True
    else:
        assert img.dim() == 3
        img = img[None, None]
        permuted = False
    orig_shp = img.shape[2:]
    if zero_crop_first:
        img = zero_crop(img[0, 0])[None, None]
        orig_shp = img.shape[2:]
    if win_size is None:
        if permuted:
            return torch.permute(img, (0, 2, 3, 4, 1)), [0, 0, 0], orig_shp
        return img, [0, 0, 0], orig_shp, aff
    if any(orig_shp[i] > win_size[i] for i in range(3)):
        crop_start = [max((orig_shp[i] - win_size[i]), 0) // 2 for i in range(3)]
        aff[:-1, -1] = aff[:-1, -1] + aff[:-1, :-1] @ np.array(crop_start)
        crop = img[:, :, crop_start[0]:crop_start[0] + win_size[0], crop_start[1]:crop_start[1] + win_size[1],
                   crop_start[2]:crop_start[2] + win_size[2]]
        if permuted:
            return torch.permute(crop, (0, 2, 3, 4, 1)), [0, 0, 0], orig_shp, aff
        return crop, crop_start, orig_shp, aff
    if permuted:
        return torch.permute(img, (0, 2, 3, 4, 1)), [0, 0, 0], orig_shp, aff
    return img, [0, 0, 0], orig_shp, aff


# ----------------------------------------------------------------------------- prepare_image
from . import volio as _volio                       # NIfTI-1 / MGH reader-writer, no nibabel needed

_VOLUME_READER = _volio.MRIread


def set_volume_reader(fn):
    """Replace the reader ``fn(path, im_only=False, dtype='float') -> (ndarray, affine)`` used for path arguments
    (default: brainfm_amd.volio.MRIread; the reference's nibabel-based utils.misc.MRIread plugs in here too)."""
    global _VOLUME_READER
    _VOLUME_READER = fn


def _read_volume(img, is_label):
    if isinstance(img, (tuple, list)) and len(img) == 2:
        return np.asarray(img[0]), np.asarray(img[1], dtype=np.float64)
    return _VOLUME_READER(img, im_only=False, dtype="int" if is_label else "float")


def add_bias_field(I, bf_scale_min=0.02, bf_scale_max=0.04, bf_std_min=0.1, bf_std_max=0.6, device="cpu"):
    """utils/test_utils.py:191-199 (test-time variant: draws its own field)."""
    bf_scale = bf_scale_min + np.random.rand(1) * (bf_scale_max - bf_scale_min)
    size_BF_small = np.round(bf_scale * np.array(I.shape)).astype(int).tolist()
    amp = float(np.float32(bf_std_min + (bf_std_max - bf_std_min) * np.random.rand(1))[0])
    noise = torch.randn(size_BF_small, dtype=torch.float).to(I.device)    # host generator: same stream as the reference on CPU
    BFsmall = GU.ew_unary(L.EW_AFFINE, noise, amp, 0.0)
    BFlog = GU.myzoom_torch(BFsmall, np.array(I.shape) / size_BF_small)
    BF = GU.ew_unary(L.EW_EXP, BFlog)
    I_bf = GU.ew_binary(L.EW_MUL, I.to(torch.float32), BF)
    return I_bf, BF


def resample(I, orig_res=[1., 1., 1.], new_res=[1., 1., 1.]):
    """utils/test_utils.py:201-224: sample on the coarse grid, zoom back to the original size."""
    if not isinstance(orig_res, list):
        orig_res = [orig_res, orig_res, orig_res]
    if not isinstance(new_res, list):
        new_res = [new_res, new_res, new_res]
    resolution = np.array(new_res)
    new_size = (np.array(I.shape) * orig_res / resolution).astype(int)
    factors = np.array(new_size) / np.array(I.shape)
    delta = (1.0 - factors) / (2.0 * factors)
    v = [np.arange(delta[a], delta[a] + new_size[a] / factors[a], 1 / factors[a])[:new_size[a]] for a in range(3)]
    II, JJ, KK = np.meshgrid(v[0], v[1], v[2], sparse=False, indexing="ij")
    II = torch.tensor(II, dtype=torch.float, device=I.device)
    JJ = torch.tensor(JJ, dtype=torch.float, device=I.device)
    KK = torch.tensor(KK, dtype=torch.float, device=I.device)
    I_resize = GU.fast_3D_interp_torch(I, II, JJ, KK, "linear")
    return GU.myzoom_torch(I_resize, 1 / factors)


@L.on_device(lambda *a, **k: _resolve_device(k.get("device", a[10] if len(a) > 10 else "cpu")))
def prepare_image(img_path, win_size=None, zero_crop_first=False, spacing=None, add_bf=False, is_CT=False,
                  is_label=False, rescale=True, hemis_mask=None, im_only=False, device="cpu"):
    """utils/test_utils.py:235-284 with every array operation on the device.  ``img_path`` is a path (needs
    set_volume_reader) or an in-memory ``(array, affine)`` pair.  Returns what the reference returns:
    final, orig, high_res, bf, aff, crop_start, orig_shp."""
    device = _resolve_device(device)
    if torch.device(device).type != "cuda":
        raise L.BfmError("prepare_image runs on a HIP device only; there is no CPU fallback in the product path")
    im, aff = _read_volume(img_path, is_label)
    aff = np.array(aff, dtype=np.float64)
    im = torch.as_tensor(np.squeeze(im), dtype=torch.int if is_label else torch.float32).to(device)
    lib = L.load()
    st = L.stream_ptr
    if not is_label:
        im = GU.ew_unary(L.EW_NAN_TO_NUM, im)
    if im.dim() > 3:                                         # averaging the RGB / frames
        flat = im.to(torch.float32).contiguous()
        n = flat.numel() // flat.shape[-1]
        out = torch.empty(flat.shape[:-1], dtype=torch.float32, device=flat.device)
        L.check(lib.bfm_mean_lastdim(L.ptr(flat), n, flat.shape[-1], L.ptr(out), st()), "mean_lastdim")
        im = out
    if is_CT and rescale:
        im = GU.ew_unary(L.EW_CLAMP, im, 0.0, 80.0)
    if not is_label and rescale:
        im = GU.ew_unary(L.EW_AFFINE, im, 1.0, -float(GU.tensor_min(im)))
        im = GU.ew_unary(L.EW_DIV, im, float(GU.tensor_max(im)))
    im, aff = MI.torch_resize(im, aff, 1.)
    orig, aff_before_crop = MI.align_volume_to_ref(im, aff, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    orig, crop_start, orig_shp, aff = center_crop(orig, win_size, zero_crop_first=zero_crop_first, aff=aff_before_crop)
    if add_bf and not is_CT:
        high_res, bf = add_bias_field(im, device=device)
        bf, _ = MI.align_volume_to_ref(bf, aff_before_crop, aff_ref=np.eye(4), return_aff=True, n_dims=3)
        bf, crop_start, orig_shp, _ = center_crop(bf, win_size, zero_crop_first=zero_crop_first, aff=aff_before_crop)
    else:
        high_res, bf = im, None
    final = resample(high_res, new_res=spacing) if spacing is not None else high_res
    high_res, _ = MI.align_volume_to_ref(high_res, aff_before_crop, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    high_res, crop_start, orig_shp, _ = center_crop(high_res, win_size, zero_crop_first=zero_crop_first,
                                                    aff=aff_before_crop)
    final, _ = MI.align_volume_to_ref(final, aff_before_crop, aff_ref=np.eye(4), return_aff=True, n_dims=3)
    final, crop_start, orig_shp, _ = center_crop(final, win_size, zero_crop_first=zero_crop_first, aff=aff_before_crop)
    if hemis_mask is not None:
        final[hemis_mask == 0] = 0
    if im_only:
        return final
    return final, orig, high_res, bf, aff, crop_start, orig_shp


# ----------------------------------------------------------------------------- deformed atlas
MNI = None          # utils/test_utils.py:38-43: the atlas volume and A = inv(its affine), read once by load_atlas()
A = None


def load_atlas(path=None):
    """MNI, aff2 = MRIread(atlas_path); A = inv(aff2) (utils/test_utils.py:38-43, done at import there).  Sets the
    module globals get_deformed_atlas falls back to and returns (MNI, A) as host arrays."""
    global MNI, A, atlas_path
    if path is not None:
        atlas_path = path
    if atlas_path is None:
        raise ValueError("set brainfm_amd.test_utils.atlas_path (the reference ships files/gca.mgz)")
    vol, aff2 = _read_volume(atlas_path, False)
    MNI = np.asarray(vol, dtype=np.float32)
    A = np.asarray(torch.tensor(np.linalg.inv(aff2), dtype=torch.float32))
    return MNI, A


@L.on_device(lambda brain_labels, regx, *a, **k: regx)
def get_deformed_atlas(brain_labels, regx, regy, regz, MNI=None, A=None):
    """utils/test_utils.py:45-57.  The atlas volume and its inverse affine are the module globals MNI / A of the
    reference (read from files/gca.mgz at import); here they can also be passed in, and default to what load_atlas()
    read.  DEF[M] = trilinear(MNI, A @ (100*reg)) for M = labels>0, one fused kernel."""
    if regx.device.type != "cuda":
        raise L.BfmError("get_deformed_atlas runs on a HIP device only; there is no CPU fallback in the product path")
    if MNI is None or A is None:
        g = globals()
        if g["MNI"] is None:
            load_atlas()
        MNI, A = g["MNI"], g["A"]
    Ah = np.asarray(torch.as_tensor(A).detach().cpu(), dtype=np.float32)[:3, :4].reshape(-1)
    Ac = (C.c_float * 12)(*[float(v) for v in Ah])
    f = lambda t: t.to(device=regx.device, dtype=torch.float32).contiguous()
    mask, rx, ry, rz, atlas = f(brain_labels), f(regx), f(regy), f(regz), f(MNI)
    out = torch.empty_like(rx)
    L.check(L.load().bfm_deformed_atlas(L.ptr(mask), L.ptr(rx), L.ptr(ry), L.ptr(rz), L.ptr(atlas), atlas.shape[0],
                                        atlas.shape[1], atlas.shape[2], Ac, rx.numel(), L.ptr(out), L.stream_ptr()),
            "deformed_atlas")
    return out


# ----------------------------------------------------------------------------- sessions
class InferenceSession:
    """Model + packed weights kept resident between calls (the reference rebuilds the 264 M-parameter
    model and reloads the checkpoint on every evaluate_image call, i.e. once per tile: quirk Q1)."""

    def __init__(self, gen_args, train_args, device, state_dict=None, ckp_path=None, passes=3):
        train_args.mfma_passes = passes
        (self.gen_args, self.train_args, self.model, self.processors, _, self.postprocessor) = \
            M.build_model(gen_args, train_args, device)
        if ckp_path is not None:
            M.load_checkpoint(ckp_path, [self.model], model_keys=["model"])
        if state_dict is not None:
            M.load_state_dict_by_suffix(self.model, state_dict)
        self.device = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
        self.tasks = self.gen_args.tasks
        # hipGraph replay of the per-tile kernel sequence (about 80 launches, many of them a few microseconds long on
        # the deep levels, where python submission is slower than the GPU): "auto" runs the first tile of a shape
        # eagerly (this also tunes the conv variants), captures on the second, replays afterwards.
        self.use_graphs = False
        self._graph_seen = set()
        self._graphs = {}
        self._graph_pool = {}
        self._graphs_epoch = 0
        # Tiles of one volume are independent until they are stitched: with lanes > 1 consecutive tiles replay their
        # graphs on separate streams (own static buffers, own scratch), so that one tile's small kernels (GroupNorm
        # finalize, split-K reduce, pooling: ~10 % of a step at 8 blocks each) and kernel tails run under the other
        # tile's convolutions.  Stitching stays on the caller's stream in the reference's tile order.
        self.lanes = max(1, int(os.environ.get("BFM_LANES", "2")))
        self._lane_streams = []
        self.atlas = None                                  # (MNI (X,Y,Z) fp32 on the device, 12 floats of inv(affine))
        if atlas_path is not None:
            self.set_atlas(*_read_volume(atlas_path, False))

    def set_atlas(self, MNI, aff):
        """The atlas volume and its vox2ras affine (utils/test_utils.py:38-43: MNI, aff2 = MRIread(atlas_path);
        A = inv(aff2), both float32).  From then on every tile of tiled_inference also yields 'deformed_atlas'
        (scripts/demo_test.py:102-104).  None switches it off."""
        if MNI is None:
            self.atlas = None
        else:
            A = np.linalg.inv(np.asarray(aff, dtype=np.float64))
            A32 = np.asarray(torch.tensor(A, dtype=torch.float32))[:3, :4].reshape(-1)
            vol = torch.as_tensor(np.asarray(MNI)).to(device=self.device, dtype=torch.float32).contiguous()
            if vol.dim() != 3:
                raise L.BfmError("atlas must be a 3-D volume, got %s" % (tuple(vol.shape),))
            self.atlas = (vol, (C.c_float * 12)(*[float(v) for v in A32]))
        self._graphs.clear()                               # captured tile graphs bake the atlas pointer and matrix in
        self._graph_seen.clear()
        self._graph_pool = {}

    def lane_streams(self, n):
        while len(self._lane_streams) < n:
            self._lane_streams.append(torch.cuda.Stream(device=self.device))
        return self._lane_streams[:n]

    def stitch_keys(self):
        """The keys tiled inference stitches for this head set, in STITCH_KEYS order."""
        tail = self.model.head.tail(self.engine)
        names = set(tail.map_names)
        if self.atlas is not None and {"regx", "regy", "regz"} <= names:
            names.add("deformed_atlas")
        return [k for k in STITCH_KEYS if k in names or (k == "label" and tail.desc.n_seg > 0)]

    @L.on_device(lambda self, *a, **k: self.device)
    def graph_group(self, ims, lane=0):
        """Run S same-shape tiles (one batch: the deep levels go through one launch per layer for all of them) through
        backbone + tail via a captured hipGraph for (shape, S) -- one graph, one set of static buffers and one scratch
        area per lane.  Returns what _run_group returns, one (maps_buf, names, label, x_cl) per tile; the buffers are
        static per (shape, S, lane) and are overwritten by the next replay on that lane, so consume them first.  Runs on
        the current stream."""
        dims = tuple(ims[0].shape[2:])
        S = len(ims)
        key = (dims, S, lane)
        eng = self.engine
        if eng.weights_epoch != self._graphs_epoch:            # trained in between: graphs hold stale packing exponents
            self._graphs.clear()
            self._graph_seen.clear()
            self._graph_pool = {}
            self._graphs_epoch = eng.weights_epoch
        if key not in self._graph_seen:
            self._graph_seen.add(key)
            eng.lane = lane
            try:
                return _run_group(self, ims)
            finally:
                eng.lane = 0
        if key not in self._graphs:
            static_in = torch.empty((S,) + tuple(ims[0].shape[1:]), dtype=torch.float32, device=self.device)
            for s_, im in enumerate(ims):
                self._tile_in(static_in[s_:s_ + 1], im)
            if lane not in self._graph_pool:
                self._graph_pool[lane] = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            eng.lane = lane
            # no cyclic garbage collection inside the capture: an earlier session (its graphs, their private pools) that
            # is only reachable through a reference cycle would be torn down by whichever allocation trips the collector,
            # and destroying a graph while this thread captures aborts the process (seen with two sessions in one test)
            import gc
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                # thread_local: a collective still in flight on a communication thread (gloo stages through the host,
                # RCCL's watchdog polls events) must not invalidate this thread's capture
                with torch.cuda.graph(g, pool=self._graph_pool[lane], capture_error_mode="thread_local"):
                    outs = _run_group(self, [static_in[s_:s_ + 1] for s_ in range(S)])
            finally:
                eng.lane = 0
                if gc_was_on:
                    gc.enable()
            self._graphs[key] = (g, static_in, outs)
        g, static_in, outs = self._graphs[key]
        for s_, im in enumerate(ims):
            self._tile_in(static_in[s_:s_ + 1], im)
        g.replay()
        return outs

    def _tile_in(self, dst, im):
        """A tile's window of the volume into its graph's contiguous input: bfm_crop3d when `im` is a window of a
        contiguous fp32 volume (what the tile loops hand over), a tensor copy for anything else."""
        st = im.stride()
        d, h, w = im.shape[-3:]
        if (im.is_cuda and im.dtype == torch.float32 and im.numel() == d * h * w and st[-1] == 1 and st[-2] >= w
                and st[-3] >= h * st[-2] and st[-3] % st[-2] == 0):
            L.check(self.engine.lib.bfm_crop3d(L.ptr(im), d, st[-3] // st[-2], st[-2], 0, 0, 0, d, h, w, L.ptr(dst),
                                               L.stream_ptr()), "crop3d")
        else:
            dst.copy_(im)

    def graph_tile(self, im, lane=0):
        """graph_group for a single tile."""
        return self.graph_group([im], lane=lane)[0]

    def has_graph(self, dims, lane=0, S=1):
        return (tuple(dims), int(S), lane) in self._graphs

    @property
    def engine(self):
        return self.model.backbone.engine(self.model.head)

    @torch.no_grad()
    @L.on_device(lambda self, *a, **k: self.device)
    def forward_fused(self, x, want_feat=True, want_seg=True):
        """One sample (1,C,D,H,W) through backbone + fused tail.  Returns the reference's output dict."""
        eng = self.engine
        dims = tuple(x.shape[2:])
        x_cl = eng.to_cl(x)
        feats = eng.backbone_cl(x_cl, dims)
        tail = self.model.head.tail(eng)
        inp = x_cl if x_cl.shape[-1] == 1 else x_cl[..., 0].contiguous()
        maps, fnorm, seg, label = tail.run(feats[-1][0], dims, input_cl=inp, want_feat=want_feat, want_seg=want_seg)
        out = OrderedDict()
        if want_feat:
            bufs = [f for f, _ in feats]
            if fnorm is not None:
                bufs[-1] = fnorm
            out["feat"] = [UNetEngine.as_ncdhw(f) for f in bufs]
        order = ["T1", "T2", "FLAIR", "CT", "segmentation", "high_res_residual", "high_res", "bias_field", "lp", "lw",
                 "rp", "rw", "fake_cortical", "regx", "regy", "regz"]
        def chans(k):
            # a head with c channels (`losses.uncertainty`: value + sigma) is c adjacent rows of the tail's buffer:
            # the reference keeps them as one (1,c,D,H,W) tensor (Trainer/models/__init__.py:57-111, joiner.py:50-55)
            r0, c = tail.channels.get(k, (None, 1))
            return tail.last_buf[r0:r0 + c][None] if c > 1 else maps[k][None, None]

        for k in order:
            if k == "segmentation":
                if seg is not None:
                    out[k] = seg.permute(3, 0, 1, 2).unsqueeze(0)
            elif k in maps:
                out[k] = chans(k)
        for k, v in maps.items():
            if k not in out and "#" not in k:
                out[k] = chans(k)
        if label is not None:
            out["label"] = label[None, None]
        return out, x_cl

    @torch.no_grad()
    def evaluate(self, inputs, feature_only=True):
        if inputs.shape[0] != 1:
            res = [self.evaluate(inputs[b:b + 1], feature_only) for b in range(inputs.shape[0])]
            if feature_only:
                return torch.cat(res, 0)
            return OrderedDict((k, ([torch.cat([r[k][j] for r in res], 0) for j in range(len(res[0][k]))]
                                    if isinstance(res[0][k], list) else torch.cat([r[k] for r in res], 0)))
                               for k in res[0])
        out, _ = self.forward_fused(inputs, want_feat=True, want_seg=not feature_only)
        return out["feat"][-1] if feature_only else out


_SESSIONS = {}


def _resolve_device(device):
    if isinstance(device, int):
        return "cuda:%d" % device
    return device


@torch.no_grad()
def evaluate_image(inputs, ckp_path, feature_only=True, device="cpu", gen_cfg=None, model_cfg=None):
    """utils/test_utils.py:289-312.  inputs: (batch, 1, s, r, c)."""
    device = _resolve_device(device)
    if torch.device(device).type != "cuda":
        raise L.BfmError("evaluate_image runs on a HIP device only; there is no CPU fallback in the product path")
    mtime = os.path.getmtime(ckp_path) if ckp_path and os.path.exists(ckp_path) else None
    key = (ckp_path, mtime, gen_cfg, model_cfg, str(device))
    if key not in _SESSIONS:
        if default_gen_cfg_file is None or default_train_cfg_file is None:
            raise ValueError("set brainfm_amd.test_utils.default_gen_cfg_file / default_train_cfg_file "
                             "(absolute paths of cfgs/generator/default.yaml, cfgs/trainer/default_train.yaml)")
        gen_args = _cfg.preprocess_cfg([default_gen_cfg_file, gen_cfg], cfg_dir=gen_cfg_dir)
        train_args = _cfg.preprocess_cfg([default_train_cfg_file, default_val_file, model_cfg], cfg_dir=train_cfg_dir)
        _SESSIONS[key] = InferenceSession(gen_args, train_args, device, ckp_path=ckp_path)
    return _SESSIONS[key].evaluate(inputs, feature_only)


# ----------------------------------------------------------------------------- tiled whole-volume inference
def tile_cost(rng):
    (x0, x1), (y0, y1), (z0, z1) = rng
    return (x1 - x0) * (y1 - y0) * (z1 - z0)


TILE_OVERHEAD_VOXELS = 160000      # fixed cost of a tile in voxel units: 80^3 takes 3.0 ms, 160^3 19 ms (t = 4.46 ms/Mvox + 0.71 ms)
PEER_SEND_VOXELS = 160000          # a peer's last (smallest) tile still has to travel to rank 0: ~33 MB over one xGMI link


def tile_time(rng):
    """Modelled cost of one tile (voxel units) used to balance the ranks."""
    return tile_cost(rng) + TILE_OVERHEAD_VOXELS


def assign_tiles(ranges, world_size):
    """Tiles -> ranks.  Tiles of one shape are kept together (they run the deep levels as one batch and read the 1 GB of
    weights once): every shape group is cut into as few chunks as keep a chunk within one rank's share of the modelled
    work, and the chunks go to the ranks longest-processing-time-first.  Rank 0 keeps its own tile outputs where the
    accumulation runs (nothing of it travels), so ties go to rank 0 and every peer starts with a handicap for the
    exposed transfer of its last tiles.  For the reference tiling of 256^3 the 27 tiles form 8 groups of equal work
    (1 x 160^3, 3 x 2, 3 x 4, 1 x 8 tiles): one group per rank at 8 ranks.  Deterministic (every rank computes the same
    map).  Returns the owner of every tile."""
    n = len(ranges)
    if world_size <= 1:
        return [0] * n
    total = sum(tile_cost(r) for r in ranges)                 # chunk sizes by voxels: what a batch costs (the per-tile
    target = total / float(world_size)                        # overhead of tile_time mostly disappears inside a batch)
    by = OrderedDict()
    for i, r in enumerate(ranges):
        by.setdefault(tuple(b - a for a, b in r), []).append(i)
    chunks = []
    for lst in by.values():
        work = sum(tile_cost(ranges[i]) for i in lst)
        nc = min(len(lst), max(1, int(-(-work // (target * 1.0001)))))
        base, extra = divmod(len(lst), nc)
        pos = 0
        for c in range(nc):
            sz = base + (1 if c < extra else 0)
            chunks.append(lst[pos:pos + sz])
            pos += sz
    cost = lambda ch: sum(tile_time(ranges[i]) for i in ch)
    chunks.sort(key=lambda ch: (-cost(ch), ch[0]))
    load = [0] * world_size
    bins = [[] for _ in range(world_size)]
    for ch in chunks:
        r = min(range(world_size), key=lambda k: (load[k], k))
        bins[r].extend(ch)
        load[r] += cost(ch)
    # rank 0 also indexes, receives and stitches (~1 ms at 8 ranks of 256^3): it takes the lightest share that keeps two
    # lanes busy -- a share of one tile runs on one lane with its small kernels exposed (measured at 8 ranks: the single
    # 160^3 tile 12.8 ms, two 160x160x80 tiles on two lanes 11.9) --; the others keep the LPT order
    order = sorted(range(world_size), key=lambda k: (load[k] == 0, len(bins[k]) < 2, load[k], k))
    owner = [0] * n
    for rank, k in enumerate(order):
        for i in bins[k]:
            owner[i] = rank
    return owner


def _stitch_tile(lib, acc, keys, maps, label, x_cl, rng, shape):
    (x0, x1), (y0, y1), (z0, z1) = rng
    td, th, tw = x1 - x0, y1 - y0, z1 - z0
    st = L.stream_ptr()
    for k in keys:
        if k == "label":
            L.check(lib.bfm_stitch_accumulate(None, L.ptr(label), L.ptr(x_cl), td, th, tw, L.ptr(acc[k]), shape[0],
                                              shape[1], shape[2], x0, y0, z0, st), "stitch label")
        else:
            L.check(lib.bfm_stitch_accumulate(L.ptr(maps[k]), None, L.ptr(x_cl), td, th, tw, L.ptr(acc[k]), shape[0],
                                              shape[1], shape[2], x0, y0, z0, st), "stitch " + k)


@torch.no_grad()
@L.on_device(lambda full_im, session, *a, **k: session.device)
def tiled_inference(full_im, session, stride=[80, 80, 80], win_size=[160, 160, 160], graphs=None, batched=None):
    """scripts/demo_test.py:66-119 on the device: per tile infer -> mask -> accumulate; then /cnt.
    full_im: (1,1,D,H,W) on the session's device.  Returns ({key: (D,H,W) fp32}, ranges, cnt).
    graphs: replay a captured hipGraph per batch of same-shape tiles (default: session.use_graphs); same kernels, same
    results.  batched (default: = graphs): tiles of one shape go through the deep levels together and the stitch is one
    launch; False = the reference's loop, tile by tile, `full[range] +=` in its order.  Same bits either way."""
    if graphs is None:
        graphs = session.use_graphs
    if batched is None:
        batched = graphs
    lib = L.load()
    eng = session.engine
    full_im = full_im.to(device=eng.device, dtype=torch.float32)
    shape = tuple(full_im.shape[2:])
    ranges = tiling_ranges(shape, stride, win_size)
    nl = session.lanes if graphs else 1
    if batched and GATHER_STITCH:                              # batches of same-shape tiles on the lanes' streams
        prev = session.use_graphs                              # graphs=True asked for replay whatever the session's default
        session.use_graphs = bool(graphs)
        try:
            return _tiled_inference_lanes(full_im, session, ranges, shape, stride, win_size)
        finally:
            session.use_graphs = prev
    cnt = count_volume(shape, ranges, eng.device)
    acc_buf, keys, sel = None, None, None
    main = torch.cuda.current_stream(eng.device)
    streams = session.lane_streams(nl) if nl > 1 else []
    consumed = [None] * nl
    start = torch.cuda.Event()
    start.record(main)
    for idx, rng in enumerate(ranges):
        (x0, x1), (y0, y1), (z0, z1) = rng
        im = full_im[:, :, x0:x1, y0:y1, z0:z1]
        k = idx % nl
        dims = (x1 - x0, y1 - y0, z1 - z0)
        if nl > 1 and session.has_graph(dims, k):
            st = streams[k]
            # the lane waits for the stitch that consumed its previous outputs (first use: for the caller's stream as it
            # stood at the start), NOT for the other lane's tile: that is the overlap
            st.wait_event(consumed[k] if consumed[k] is not None else start)
            with torch.cuda.stream(st):
                maps_buf, names, label, x_cl = session.graph_tile(im, lane=k)
                done = torch.cuda.Event()
                done.record(st)
            main.wait_event(done)
        elif graphs:
            if nl > 1:
                torch.cuda.synchronize(eng.device)             # eager / capture passes run alone (warm-up only)
            maps_buf, names, label, x_cl = session.graph_tile(im, lane=k)
        else:
            maps_buf, names, label, x_cl = _run_tile(session, im, raw=True)
        if acc_buf is None:
            keys = [k_ for k_ in STITCH_KEYS if k_ in names or (k_ == "label" and label is not None)]
            sel = torch.tensor([names.index(k_) if k_ != "label" else -1 for k_ in keys], dtype=torch.int32,
                               device=eng.device)
            acc_buf = torch.zeros((len(keys),) + shape, dtype=torch.float32, device=eng.device)
        tv = (x1 - x0) * (y1 - y0) * (z1 - z0)
        L.check(lib.bfm_stitch_accumulate_multi(L.ptr(maps_buf), tv, L.ptr(sel), len(keys), L.ptr(label), L.ptr(x_cl),
                                                x1 - x0, y1 - y0, z1 - z0, L.ptr(acc_buf), shape[0], shape[1],
                                                shape[2], x0, y0, z0, L.stream_ptr()), "stitch_multi")
        if nl > 1:
            consumed[k] = torch.cuda.Event()
            consumed[k].record(main)
    n = shape[0] * shape[1] * shape[2]
    L.check(lib.bfm_divide_by_count_multi(L.ptr(acc_buf), L.ptr(cnt), n, len(keys), L.stream_ptr()), "divide_by_count")
    acc = OrderedDict((k, acc_buf[j]) for j, k in enumerate(keys))
    return acc, ranges, cnt


def _cached_count_volume(session, shape, ranges, stride, win_size, dev):
    """The count volume depends on the tiling alone: kept on the session (callers read it, nobody writes it)."""
    ckey = ("cnt", tuple(shape), tuple(stride), tuple(win_size), str(dev))
    cache = session.__dict__.setdefault("_dist_cnt", {})
    if ckey not in cache:
        cache[ckey] = count_volume(shape, ranges, dev)
    return cache[ckey]


def _session_stitch_ops(session):
    ops = getattr(session, "_stitch_ops", None)               # kept on the session: its device tables are reused
    if ops is None:
        ops = session._stitch_ops = HipStitchOps(session)
    return ops


@torch.no_grad()
def _tiled_inference_lanes(full_im, session, ranges, shape, stride, win_size):
    """tiled_inference with several tiles in flight: the lanes run independently of each other.  Every tile's masked
    maps are packed ([K][n], bfm_pack_tile_multi) on its lane's stream into its slot of one persistent buffer -- no
    lane ever waits for another lane's tile, which the per-tile `full[range] +=` on the caller's stream forced through
    the reference's tile order -- tiles go to the lanes largest first onto the less loaded lane, and one launch
    (bfm_stitch_gather_multi) then sums every voxel's tiles in the reference's order and divides by their number: the
    same bits as the sequential form."""
    dev = full_im.device
    ops = _session_stitch_ops(session)
    nl = session.lanes
    nkeys = len(session.stitch_keys())
    offs, total = [], 0
    for r in ranges:
        offs.append(total)
        total += tile_cost(r) * nkeys
    buf = _exchange_buffer(session, "rows", total, dev)
    load = [0] * nl
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event()
    start.record(main)                                         # the input is in place, last volume's rows are consumed
    # compact rows (only what the tile mask keeps is packed and read back): every slot keeps its full-size stride here,
    # so nothing about the volume has to reach the host
    # BFM_SKIP_EMPTY_TILES=all: also bring the tiles' survivor counts to the host (one small copy and a wait per volume)
    # and do not run a tile without any input at all, as the multi-GPU path does for free
    skip = COMPACT and os.environ.get("BFM_SKIP_EMPTY_TILES", "1") == "all"
    index = ops.index_volume(full_im, ranges, counts=skip) if COMPACT else None
    live = None
    if skip and index is not None and index.nnz is not None:
        live = [i for i in range(len(ranges)) if index.nnz[i] > 0]
    last, keys = {}, None
    for batch in tile_batches(ranges, live):                   # same-shape tiles together: the deep levels run batched
        k = min(range(nl), key=lambda j: (load[j], j))
        load[k] += sum(tile_time(ranges[i]) for i in batch)
        ims = [full_im[:, :, ranges[i][0][0]:ranges[i][0][1], ranges[i][1][0]:ranges[i][1][1],
                       ranges[i][2][0]:ranges[i][2][1]] for i in batch]
        outs = [buf[offs[i]:offs[i] + tile_cost(ranges[i]) * nkeys] for i in batch]
        keys, _, done = ops.run_group(ims, outs, lane=k, after=start, index=index, tiles_idx=batch,
                                      strides=[tile_cost(ranges[i]) for i in batch])
        if done is not None:
            last[k] = done
    for ev in last.values():
        main.wait_event(ev)
    srcs = [buf[offs[i]:offs[i] + tile_cost(r) * nkeys].view(nkeys, tile_cost(r)) for i, r in enumerate(ranges)]
    acc_buf = torch.empty((nkeys,) + tuple(shape), dtype=torch.float32, device=dev)
    ops.gather_all(acc_buf, srcs, ranges, shape, index=index)
    cnt = _cached_count_volume(session, shape, ranges, stride, win_size, dev)
    if keys is None:                                           # every tile was empty
        keys = session.stitch_keys()
    return OrderedDict((k_, acc_buf[j]) for j, k_ in enumerate(keys)), ranges, cnt


@torch.no_grad()
@L.on_device(lambda full_im, session, *a, **k: session.device)
def prepare_tile_graphs(full_im, session, stride=[80, 80, 80], win_size=[160, 160, 160], world=1, rank=0, group=None):
    """Tune + capture the hipGraph of every tile shape this rank will see (first tile of each shape, run twice:
    eager, then capture + replay).  Optional: tiled_inference does the same lazily on the first volumes.  With more
    than one rank (collective call then) the ranks first agree on the conv variants, so that the graphs captured here
    are the ones tiled_inference_distributed replays."""
    shape = tuple(full_im.shape[2:])
    ranges = tiling_ranges(shape, stride, win_size)
    owner = assign_tiles(ranges, world)
    if world > 1:
        agree_on_conv_variants(session, full_im, ranges, group)
    done = set()
    mine = [i for i in range(len(ranges)) if owner[i] == rank]
    for batch in tile_batches(ranges, mine, min_batches=session.lanes if world > 1 else 1):
        dims = tuple(b - a for a, b in ranges[batch[0]])
        if (dims, len(batch)) in done:
            continue
        done.add((dims, len(batch)))
        ims = [full_im[:, :, ranges[i][0][0]:ranges[i][0][1], ranges[i][1][0]:ranges[i][1][1],
                       ranges[i][2][0]:ranges[i][2][1]].to(device=session.device, dtype=torch.float32) for i in batch]
        for lane in range(session.lanes):
            while not session.has_graph(dims, lane, len(batch)):
                session.graph_group(ims, lane=lane)
    torch.cuda.synchronize(session.device)
    return sorted(done)                                       # [(tile shape, tiles per batch)]


def _tile_tail(session, feats, x_cl, dims, raw):
    """Fused tail (+ the deformed atlas) of one tile whose backbone features are `feats`."""
    eng = session.engine
    tail = session.model.head.tail(eng)
    atlas = getattr(session, "atlas", None)
    if atlas is not None and not {"regx", "regy", "regz"} <= set(tail.map_names):
        atlas = None                                           # the reference's loop needs the registration head too
    # every consumer of a tile's maps (stitch / pack kernels, the atlas gather) keeps them only where the tile's input is
    # non-zero: the heads of the other voxels are not evaluated (engine.mask_skip; rows hold unwritten memory there)
    maps, _, _, label = tail.run(feats[-1][0], dims, input_cl=x_cl, want_feat=False, want_seg=False,
                                 extra_rows=1 if atlas is not None else 0, skip_zero_input=eng.mask_skip)
    if atlas is not None:
        # scripts/demo_test.py:102-104: get_deformed_atlas(mask, regx, regy, regz) with mask = (im != 0), from the
        # unmasked registration maps, into the spare row of the tail's map buffer; the stitcher multiplies by the mask
        # again (DEF is 0 outside it already)
        vol, A = atlas
        row = tail.last_buf[len(tail.map_names)]
        L.check(L.load().bfm_deformed_atlas_tile(L.ptr(x_cl), L.ptr(maps["regx"]), L.ptr(maps["regy"]),
                                                 L.ptr(maps["regz"]), L.ptr(vol), vol.shape[0], vol.shape[1],
                                                 vol.shape[2], A, row.numel(), L.ptr(row), L.stream_ptr()),
                "deformed_atlas_tile")
        maps["deformed_atlas"] = row
    if raw:
        return tail.last_buf, list(maps.keys()), label, x_cl
    return maps, label, x_cl


def _run_tile(session, im, raw=False):
    eng = session.engine
    dims = tuple(im.shape[2:])
    x_cl = eng.to_cl(im)
    return _tile_tail(session, eng.backbone_cl(x_cl, dims, mask_last=True), x_cl, dims, raw)


def _run_group(session, ims):
    """S same-shape tiles: the deep levels of the backbone run once over the batch (engine.backbone_batch; a tile's
    result is bit-identical to its single-tile pass), the rest per tile.  Returns [_run_tile(raw=True) tuples].
    (Issuing the per-tile parts on S forked streams -- parallel branches of the captured graph -- was measured and
    dropped: 132.5 ms per 256^3 volume against 128.1 with the tiles in sequence; two lanes already fill the chip.)"""
    eng = session.engine
    if len(ims) == 1 or not eng.has_deep_region():
        return [_run_tile(session, im, raw=True) for im in ims]
    dims = tuple(ims[0].shape[2:])
    x_cls = [eng.to_cl(im) for im in ims]
    feats = eng.backbone_batch(x_cls, dims, mask_last=True)
    return [_tile_tail(session, f, x_cl, dims, True) for f, x_cl in zip(feats, x_cls)]


GROUP_MAX = max(1, int(os.environ.get("BFM_GROUP_MAX", "8")))       # most tiles batched through the deep levels at once


def tile_batches(ranges, idxs=None, group_max=None, min_batches=1):
    """The tiles `idxs` (default: all) as batches of same-shape tiles, at most `group_max` per batch, of balanced
    sizes, reference order kept inside a batch; batches sorted by modelled work, largest first.  For the reference
    tiling of 256^3 this is 8 batches of equal work (1 x 160^3, 3 x 2, 3 x 4, 1 x 8 tiles).  min_batches: when there
    would be fewer batches than that (a rank of an 8-GPU run owns ONE shape group but has two lanes), the batches with
    the most tiles are halved until there are enough, so that every lane has work and one half's small kernels run
    under the other half's convolutions."""
    gm = GROUP_MAX if group_max is None else max(1, int(group_max))
    by = OrderedDict()
    for i in (range(len(ranges)) if idxs is None else idxs):
        by.setdefault(tuple(b - a for a, b in ranges[i]), []).append(i)
    out = []
    for lst in by.values():
        nb = -(-len(lst) // gm)
        base, extra = divmod(len(lst), nb)
        pos = 0
        for b in range(nb):
            sz = base + (1 if b < extra else 0)
            out.append(lst[pos:pos + sz])
            pos += sz
    while 0 < len(out) < min_batches:
        j = max(range(len(out)), key=lambda k: (len(out[k]), -k))
        if len(out[j]) < 2:
            break
        h = (len(out[j]) + 1) // 2
        out[j:j + 1] = [out[j][:h], out[j][h:]]
    out.sort(key=lambda b: (-sum(tile_time(ranges[i]) for i in b), b[0]))
    return out


COMPACT = os.environ.get("BFM_COMPACT", "1") != "0"    # tiles' rows hold the voxels the tile mask keeps only (0: all of them)
SKIP_EMPTY = os.environ.get("BFM_SKIP_EMPTY_TILES", "1") != "0"   # multi-GPU path: tiles without any input are not run


class TileIndex:
    """HipStitchOps.index_volume's result: volume (D,H,W) fp32, pos (int32, all tiles' entries back to back), base[i] /
    sizes[i] = tile i's slice of pos, nnz_dev / nnz = surviving voxels per tile (device / host or None), ready (event)."""


class HipStitchOps:
    """Device-side pack / accumulate used by the multi-GPU path (HIP kernels, one launch per tile each)."""

    def __init__(self, session):
        import weakref
        self._session = weakref.ref(session) if session is not None else None   # the session owns its ops: no cycle
        self.lib = L.load()
        self._sel = {}
        self._lane_after = {}

    @property
    def session(self):
        return self._session() if self._session is not None else None

    def _identity(self, k, dev):
        if (k, dev) not in self._sel:
            self._sel[(k, dev)] = torch.arange(k, dtype=torch.int32, device=dev)
        return self._sel[(k, dev)]

    def index_volume(self, full_im, ranges, counts=False):
        """The compact shipping form's index of one volume (bfm_tile_mask_index): for every tile of `ranges`, the column
        of each of its voxels among the tile's surviving (input != 0) voxels, computed from the volume before any tile
        runs (three launches on the current stream).  counts=True also brings the tiles' survivor counts to the host
        (one small copy + a wait for those three kernels): the multi-GPU exchange sizes its buffers with them.
        Returns None when the volume has more than one channel (the mask is then not a function of one image)."""
        if full_im.dim() != 5 or full_im.shape[0] != 1 or full_im.shape[1] != 1:
            return None
        dev = full_im.device
        vol = full_im[0, 0]
        if vol.dtype != torch.float32 or not vol.is_contiguous():
            vol = vol.to(torch.float32).contiguous()
        shape = tuple(vol.shape)
        key = ("index", shape, tuple(tuple(r_) for r in ranges for r_ in r), dev)
        ent = self._sel.get(key)
        if ent is None:
            rows, base, blk = [], 0, 0
            for r in ranges:
                dims = [b - a for a, b in r]
                n = dims[0] * dims[1] * dims[2]
                rows.append([base, r[0][0], r[1][0], r[2][0], dims[0], dims[1], dims[2], blk])
                base += n
                blk += self.lib.bfm_tile_mask_blocks(n)
            ent = self._sel[key] = dict(
                tab=torch.tensor(rows, dtype=torch.int64).to(dev), base=[r_[0] for r_ in rows], blocks=blk,
                pos=torch.empty(base, dtype=torch.int32, device=dev),
                nnz=torch.empty(len(ranges), dtype=torch.int32, device=dev),
                ws=torch.empty(blk, dtype=torch.int32, device=dev))
        L.check(self.lib.bfm_tile_mask_index(L.ptr(vol), shape[0], shape[1], shape[2], L.ptr(ent["tab"]), len(ranges),
                                             ent["blocks"], L.ptr(ent["pos"]), L.ptr(ent["nnz"]), L.ptr(ent["ws"]),
                                             L.stream_ptr()), "tile_mask_index")
        idx = TileIndex()
        idx.volume, idx.pos, idx.base, idx.nnz_dev = vol, ent["pos"], ent["base"], ent["nnz"]
        idx.sizes = [tile_cost(r) for r in ranges]
        idx.ready = torch.cuda.Event()
        idx.ready.record(torch.cuda.current_stream(dev))
        idx.nnz = [int(v) for v in ent["nnz"].cpu().tolist()] if counts else None
        return idx

    def run_group(self, ims, outs=None, lane=None, after=None, index=None, tiles_idx=None, strides=None):
        """Masked, float typed [K][n] rows of every tile of a same-shape batch (written into ``outs[i]`` when given: the
        send buffer / stitch slots).  lane / after: replay the batch's graph and pack on that lane's stream once event
        ``after`` (recorded on the caller's stream) has passed; the third return value is then the event to wait for
        before reading the rows.  index / tiles_idx / strides: pack the compact form instead (index_volume; the batch's
        tile numbers; the row stride of each: [K][stride] with the surviving voxels in the first columns).
        Returns (keys, [rows per tile], event | None)."""
        sess = self.session
        dims = tuple(ims[0].shape[2:])
        S = len(ims)
        outs = outs if outs is not None else [None] * S
        cp = [None] * S
        if index is not None:
            cp = [(index.pos[index.base[i]:index.base[i] + index.sizes[i]], int(rs)) for i, rs in zip(tiles_idx, strides)]
        on_lane = (lane is not None and sess.use_graphs and sess.has_graph(dims, lane, S))
        if on_lane:
            st = sess.lane_streams(sess.lanes)[lane]
            if after is not None:
                st.wait_event(after)
            if index is not None:
                st.wait_event(index.ready)
            if lane in self._lane_after:                       # an eager / capture pass of this lane ran on the caller's
                st.wait_event(self._lane_after.pop(lane))      # stream: its buffers share the lane's graph pool
            with torch.cuda.stream(st):
                res = [self._tile_rows(o, out, c) for o, out, c in zip(sess.graph_group(ims, lane=lane), outs, cp)]
                done = torch.cuda.Event()
                done.record(st)
            return res[0][0], [r[1] for r in res], done
        if sess.use_graphs:
            if lane is not None:
                torch.cuda.synchronize(sess.device)            # eager / capture passes run alone (warm-up only)
            tiles = sess.graph_group(ims, lane=lane or 0)
        else:
            tiles = _run_group(sess, ims)
        if index is not None:
            torch.cuda.current_stream(sess.device).wait_event(index.ready)
        res = [self._tile_rows(o, out, c) for o, out, c in zip(tiles, outs, cp)]
        if lane is not None and sess.use_graphs:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(sess.device))
            self._lane_after[lane] = ev
        return res[0][0], [r[1] for r in res], None

    def run_tile(self, im, out=None, lane=None, after=None):
        """run_group for one tile: (keys, rows[, event])."""
        keys, rows, done = self.run_group([im], [out], lane=lane, after=after)
        return (keys, rows[0], done) if lane is not None else (keys, rows[0])

    def _tile_rows(self, outs, out, compact=None):
        maps_buf, names, label, x_cl = outs
        keys = [k for k in STITCH_KEYS if k in names or (k == "label" and label is not None)]
        skey = (tuple(names), label is not None)
        if skey not in self._sel:
            self._sel[skey] = torch.tensor([names.index(k) if k != "label" else -1 for k in keys], dtype=torch.int32,
                                           device=x_cl.device)
        n = x_cl.numel()
        if compact is not None:
            pos, rs = compact
            if pos.numel() != n:
                raise L.BfmError("tile index has %d entries, the tile %d voxels" % (pos.numel(), n))
            rows = out if out is not None else torch.empty(len(keys) * rs, dtype=torch.float32, device=x_cl.device)
            if rows.numel() < len(keys) * rs:
                raise L.BfmError("compact rows need %d floats, the slot has %d" % (len(keys) * rs, rows.numel()))
            if rs > 0:                                         # rs == 0: an all-zero tile keeps nothing
                L.check(self.lib.bfm_pack_tile_compact(L.ptr(maps_buf), n, L.ptr(self._sel[skey]), len(keys),
                                                       L.ptr(label), L.ptr(x_cl), n, L.ptr(pos), rs, L.ptr(rows),
                                                       L.stream_ptr()), "pack_tile_compact")
            return keys, rows[:len(keys) * rs].view(len(keys), rs)
        rows = out if out is not None else torch.empty(len(keys) * n, dtype=torch.float32, device=x_cl.device)
        L.check(self.lib.bfm_pack_tile_multi(L.ptr(maps_buf), n, L.ptr(self._sel[skey]), len(keys), L.ptr(label),
                                             L.ptr(x_cl), n, L.ptr(rows), L.stream_ptr()), "pack_tile_multi")
        return keys, rows.view(len(keys), n)

    def add_all(self, acc_buf, rows, rng, shape):
        """acc_buf [K][D,H,W] += rows [K][n] of one tile (already masked)."""
        (x0, x1), (y0, y1), (z0, z1) = rng
        k, n = rows.shape
        L.check(self.lib.bfm_stitch_accumulate_multi(L.ptr(rows), n, L.ptr(self._identity(k, rows.device)), k, None,
                                                     None, x1 - x0, y1 - y0, z1 - z0, L.ptr(acc_buf), shape[0],
                                                     shape[1], shape[2], x0, y0, z0, L.stream_ptr()), "stitch_multi")

    def gather_all(self, acc_buf, srcs, ranges, shape, index=None):
        """acc_buf [K][D,H,W] = the whole stitch in one launch: srcs[i] = packed rows [K][n_i] of tile i (reference
        order), every voxel summed over its tiles in that order, divided by their number, written once
        (bfm_stitch_gather_multi; same bits as add_all per tile on a zeroed volume + finalize_all).
        index: the rows are compact ([K][stride_i], run_group(index=...)) -> bfm_stitch_gather_compact."""
        if index is not None:
            key = tuple((s_.data_ptr(), index.pos.data_ptr() + 4 * index.base[i]) + tuple(a for a, _ in r) +
                        tuple(b - a for a, b in r) + (s_.shape[1] if s_.dim() == 2 else 0,)
                        for i, (s_, r) in enumerate(zip(srcs, ranges)))
            tab = self._sel.get(("gatherc", acc_buf.device))
            if tab is None or tab[0] != key:
                host = torch.tensor([list(k_) + [0] for k_ in key], dtype=torch.int64)
                tab = self._sel[("gatherc", acc_buf.device)] = (key, host.to(acc_buf.device))
            torch.cuda.current_stream(acc_buf.device).wait_event(index.ready)
            L.check(self.lib.bfm_stitch_gather_compact(L.ptr(tab[1]), len(ranges), acc_buf.shape[0], L.ptr(index.volume),
                                                       L.ptr(acc_buf), shape[0], shape[1], shape[2], L.stream_ptr()),
                    "stitch_gather_compact")
            return
        key = tuple((s_.data_ptr(),) + tuple(a for a, _ in r) + tuple(b - a for a, b in r) for s_, r in zip(srcs, ranges))
        tab = self._sel.get(("gather", acc_buf.device))
        if tab is None or tab[0] != key:
            host = torch.tensor([list(k_) + [0] for k_ in key], dtype=torch.int64)
            tab = self._sel[("gather", acc_buf.device)] = (key, host.to(acc_buf.device))
        L.check(self.lib.bfm_stitch_gather_multi(L.ptr(tab[1]), len(ranges), acc_buf.shape[0], L.ptr(acc_buf), shape[0],
                                                 shape[1], shape[2], L.stream_ptr()), "stitch_gather_multi")

    def finalize_all(self, acc_buf, cnt):
        L.check(self.lib.bfm_divide_by_count_multi(L.ptr(acc_buf), L.ptr(cnt), cnt.numel(), acc_buf.shape[0],
                                                   L.stream_ptr()), "divide_multi")


def _exchange_buffer(session, name, numel, dev):
    """Send / receive buffers of the multi-GPU exchange, kept on the session between volumes."""
    bufs = getattr(session, "_dist_bufs", None)
    if bufs is None:
        bufs = session._dist_bufs = {}
    t = bufs.get(name)
    if t is None or t.numel() != numel or t.device != torch.device(dev):
        t = bufs[name] = torch.empty(numel, dtype=torch.float32, device=dev)
    return t


DIST_ROUNDS = os.environ.get("BFM_DIST_ROUNDS", "1") != "0"
GATHER_STITCH = os.environ.get("BFM_GATHER_STITCH", "1") != "0"     # 0: per-tile accumulate + divide on rank 0


@torch.no_grad()
def agree_on_conv_variants(session, full_im, ranges, group=None):
    """Every rank launches the same conv variant for the same layer shape.  The variants are timed per process
    (UNetEngine._autotune) and agree to ~1e-6, not bit for bit, so without this a volume's result would depend on which
    rank happened to compute which tile.  Rank 0 times the shapes of every distinct tile size of this volume (one eager
    tile each, once per session) and broadcasts its table; the others adopt it.  Skipped when the variants are pinned
    (BFM_CONV_VER, BFM_CONV_TUNE=0)."""
    import os
    import torch.distributed as dist
    if os.environ.get("BFM_CONV_VER") or os.environ.get("BFM_CONV_TUNE", "1") == "0" or \
            os.environ.get("BFM_DIST_AGREE", "1") == "0":
        return
    sizes = sorted({tuple(b - a for a, b in r) for r in ranges})
    agreed = session.__dict__.setdefault("_agreed_sizes", set())
    todo = [s_ for s_ in sizes if s_ not in agreed]
    if not todo:
        return
    eng = session.engine
    src = 0 if group is None else dist.get_global_rank(group, 0)
    box = [None]
    if dist.get_rank(group) == 0:
        for s_ in todo:
            im = full_im[:, :, :s_[0], :s_[1], :s_[2]]
            if tuple(im.shape[2:]) == s_:
                _run_tile(session, im, raw=True)                 # times whatever shapes are new
        torch.cuda.synchronize(full_im.device)
        box = [eng.conv_choices()]
    with torch.cuda.device(full_im.device):                   # RCCL stages the pickled table through the current device
        dist.broadcast_object_list(box, src=src, group=group)
    if eng.adopt_conv_choices(box[0]) and session._graphs:
        session._graphs.clear()
        session._graph_seen.clear()
        session._graph_pool = {}
    agreed.update(todo)


def broadcast_volume(full_im, device, group=None, shape=None):
    """The volume from rank 0 to every rank (the reference loads one file in one process,
    scripts/demo_test.py:71; the tiles of the other ranks are windows of it).  Rank 0 passes the (1,1,D,H,W) fp32
    volume; the others pass None and get a fresh tensor.  shape: the spatial shape when every rank knows it already (a
    5-number header goes first otherwise).  67 MB for 256^3, 537 MB for 512^3: one RCCL broadcast."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if dist.get_world_size(group) == 1:
        return full_im
    if shape is None:
        hdr = torch.zeros(5, dtype=torch.int64, device=device)
        if rank == 0:
            hdr = torch.tensor(list(full_im.shape), dtype=torch.int64, device=device)
        dist.broadcast(hdr, src=0, group=group)
        full_shape = tuple(int(v) for v in hdr.tolist())
    else:
        full_shape = (1, 1) + tuple(shape)
    if rank == 0:
        if tuple(full_im.shape) != full_shape or full_im.dtype != torch.float32:
            raise L.BfmError("rank 0's volume is %s %s, expected %s float32" % (tuple(full_im.shape), full_im.dtype, full_shape))
        buf = full_im.contiguous()
    else:
        buf = torch.empty(full_shape, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=0, group=group)
    return buf


@L.on_device(lambda full_im, session, *a, **k: session.device if session is not None else None)
def tiled_inference_distributed(full_im, session, stride=[80, 80, 80], win_size=[160, 160, 160], group=None,
                                ops=None, rounds=None, shape=None, stats=None, broadcast=False):
    """Tiles are independent (GroupNorm statistics are per tile), so they shard over ranks with no
    data-path collective; gathers to rank 0 carry every rank's masked tile outputs, and rank 0 accumulates them
    in the reference's tile order so the result is bit-identical to the single-GPU path (fp32 += is order
    dependent where cnt reaches 8).
    rounds (default, BFM_DIST_ROUNDS=0 turns it off): every rank runs its tiles largest first and round k -- the k-th
    tile of every rank -- leaves in its own asynchronous gather as soon as it is computed, so the transfers ride under
    the next round's kernels and only the last (smallest) round's is exposed; otherwise ONE gather at the end.  With
    27 tiles on 8 GPUs a rank computes ~20 ms and ships ~280 MB: the single gather would add ~1/3 to the step.
    broadcast=True (the same on every rank): only rank 0 holds the volume (the reference reads one file in one process)
    and sends it to the peers inside this call (broadcast_volume; the peers pass full_im=None and, if they know it, the
    spatial `shape` -- a 5-number header travels first otherwise).  broadcast=False: every rank passes its own copy.
    stats: a dict that receives what the exchange moved (bytes per peer and round, and on rank 0 two events that bracket
    what the gathers left exposed after its own tiles).
    Returns (acc, ranges, cnt) on rank 0, (None, ranges, None) elsewhere.
    ``ops`` (run_tile/add/finalize) defaults to the HIP kernels; the gloo unit tests inject host ops
    to exercise the sharding / ordering logic without a GPU."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    if rounds is None:
        rounds = DIST_ROUNDS
    if ops is None:
        ops = _session_stitch_ops(session) if session is not None else HipStitchOps(session)
    if broadcast:
        if rank != 0 and full_im is not None:
            raise L.BfmError("broadcast=True: only rank 0 passes the volume")
        bdev = full_im.device if full_im is not None else (session.device if session is not None else torch.device("cpu"))
        full_im = broadcast_volume(full_im, bdev, group, shape)
        if stats is not None:
            stats["broadcast_bytes"] = int(full_im.numel()) * 4 if world > 1 else 0
    elif full_im is None:
        raise L.BfmError("tiled_inference_distributed: this rank has no volume (pass broadcast=True on every rank)")
    shape = tuple(full_im.shape[2:])
    ranges = tiling_ranges(shape, stride, win_size)
    owner = assign_tiles(ranges, world)
    dev = full_im.device
    direct = hasattr(ops, "add_all")                           # HIP ops write straight into the send buffer
    nkeys = getattr(ops, "n_keys", None)
    if nkeys is None and hasattr(ops, "keys"):
        nkeys = len(ops.keys)
    if nkeys is None and session is not None:
        nkeys = len(session.stitch_keys())
    if nkeys is None:
        nkeys = len(STITCH_KEYS)
    # per rank: its tiles as batches of same-shape tiles (tile_batches: the deep levels of a batch run as one launch per
    # layer), largest batch first; slot[i] = (round, offset in that round's buffer) with round k = the k-th batch of
    # every rank.  Rank 0's own tiles never travel: they are packed into a private buffer per round, and the round's
    # (padded) size is set by the peers alone.
    nlanes = session.lanes if (session is not None and getattr(session, "use_graphs", False)) else 1
    # compact rows: a tile ships the voxels its mask keeps (known from the volume on every rank before any tile runs:
    # HipStitchOps.index_volume brings the counts to the host, the same on every rank) -- 1/5 of the bytes on a head in
    # a 256^3 box
    index = None
    if direct and COMPACT and GATHER_STITCH and dev.type == "cuda" and hasattr(ops, "index_volume"):
        index = ops.index_volume(full_im, ranges, counts=True)
    width = [index.nnz[i] if index is not None else tile_cost(r) for i, r in enumerate(ranges)]   # columns per row
    # a tile whose input is all zero keeps nothing of what it computes (scripts/demo_test.py:88-100) and adds +0 wherever
    # it is stitched: with the counts on the host it is not run at all (BFM_SKIP_EMPTY_TILES=0: run it)
    live = [index is None or not SKIP_EMPTY or width[i] > 0 for i in range(len(ranges))]
    batches_of = [tile_batches(ranges, [i for i in range(len(ranges)) if owner[i] == r and live[i]], min_batches=nlanes)
                  for r in range(world)]
    nrounds = max([len(b) for b in batches_of] + [1]) if rounds else 1
    round_of, off_of = {}, {}
    round_numel = [1] * nrounds
    own_numel = [1] * nrounds
    for i in range(len(ranges)):
        if not live[i]:
            round_of[i], off_of[i] = 0, 0                      # nothing computed, nothing shipped, nothing read
    for r in range(world):
        fill = [0] * nrounds
        for k, batch in enumerate(batches_of[r]):
            kk = k if rounds else 0
            for i in batch:
                round_of[i], off_of[i] = kk, fill[kk]
                fill[kk] += width[i] * nkeys
        for kk in range(nrounds):
            if r == 0:
                own_numel[kk] = max(own_numel[kk], fill[kk])
            else:
                round_numel[kk] = max(round_numel[kk], fill[kk])

    def _buf(name, numel):
        if session is not None and direct:
            return _exchange_buffer(session, name, numel, dev)     # persistent; padding is never read: no fill
        return torch.zeros(numel, dtype=torch.float32, device=dev)

    if direct and session is not None and dev.type == "cuda":
        agree_on_conv_variants(session, full_im, ranges, group)
    keys = None
    works, gathered, own, pending = [], [], [], []
    mine = batches_of[rank]
    lanes = session.lanes if (direct and session is not None and session.use_graphs and dev.type == "cuda") else 1
    start = None
    if lanes > 1:
        start = torch.cuda.Event()
        start.record(torch.cuda.current_stream(dev))           # send buffers are free, the input is in place
    lane_load = [0] * lanes                                    # this rank's batches go to its less loaded lane
    for kk in range(nrounds):
        sbuf = _buf("send%d" % kk, round_numel[kk])            # on rank 0: the padding the gather asks of its root
        dst = sbuf
        if rank == 0:
            dst = _buf("own%d" % kk, own_numel[kk])
            own.append(dst)
        todo = [b_ for k, b_ in enumerate(mine) if (k if rounds else 0) == kk]
        for batch in todo:
            ims = [full_im[:, :, ranges[i][0][0]:ranges[i][0][1], ranges[i][1][0]:ranges[i][1][1],
                           ranges[i][2][0]:ranges[i][2][1]] for i in batch]
            outs = [dst[off_of[i]:off_of[i] + width[i] * nkeys] for i in batch]
            cargs = dict(index=index, tiles_idx=batch, strides=[width[i] for i in batch]) if index is not None else {}
            lane = min(range(lanes), key=lambda j: (lane_load[j], j))
            lane_load[lane] += sum(tile_time(ranges[i]) for i in batch)
            if direct and session is not None and session.use_graphs and \
                    not session.has_graph(tuple(ims[0].shape[2:]), lane, len(batch)):
                for w in works:                                    # warm-up only: no transfer in flight while a graph
                    w.wait()                                       # is being captured
            if direct and lanes > 1:
                # this rank's batches run on its lanes' streams, independently of each other; the gather of a round waits
                # for that round's tiles only -- and on rank 0 for none: its receives are posted at once, whatever it is
                # still computing itself
                keys, _, done = ops.run_group(ims, outs, lane=lane, after=start, **cargs)
                if done is not None:
                    if rank == 0:
                        pending.append(done)
                    else:
                        torch.cuda.current_stream(dev).wait_event(done)
            elif direct:
                keys, _, _ = ops.run_group(ims, outs, **cargs)
            else:
                for i, im in zip(batch, ims):
                    keys, rows = ops.run_tile(im)
                    if len(keys) != nkeys:
                        raise RuntimeError("ops.run_tile returned %d maps, expected %d" % (len(keys), nkeys))
                    n = tile_cost(ranges[i]) * nkeys
                    dst[off_of[i]:off_of[i] + n] = rows.reshape(-1)
        g = [_buf("recv%d_%d" % (kk, r), round_numel[kk]) for r in range(world)] if rank == 0 else None
        gathered.append(g)
        works.append(dist.gather(sbuf, g, dst=0, group=group, async_op=True))
    if stats is not None:
        stats["world"] = world
        stats["rounds"] = nrounds
        stats["round_bytes_per_peer"] = [int(v) * 4 for v in round_numel]
        stats["bytes_sent_per_peer"] = int(sum(round_numel)) * 4 if world > 1 else 0
        stats["compact_rows"] = index is not None
    if rank == 0 and stats is not None and dev.type == "cuda":
        for ev in pending:                                    # rank 0's own tiles are done ...
            torch.cuda.current_stream(dev).wait_event(ev)
        stats["ev_own_done"] = torch.cuda.Event(enable_timing=True)
        stats["ev_own_done"].record(torch.cuda.current_stream(dev))
    for w in works:
        w.wait()
    if rank == 0 and stats is not None and dev.type == "cuda":
        stats["ev_gathers_done"] = torch.cuda.Event(enable_timing=True)      # ... and every peer's rows have arrived
        stats["ev_gathers_done"].record(torch.cuda.current_stream(dev))
    if rank != 0:
        return None, ranges, None
    if keys is None:
        keys = (session.stitch_keys() if session is not None else [k for k in STITCH_KEYS])[:nkeys]
    for ev in pending:                                        # rank 0's own tiles (lanes)
        torch.cuda.current_stream(dev).wait_event(ev)
    srcs = []
    for i, rng in enumerate(ranges):                          # reference tile order
        nv = width[i]
        src = own[round_of[i]] if owner[i] == 0 else gathered[round_of[i]][owner[i]]
        srcs.append(src[off_of[i]:off_of[i] + nv * nkeys].reshape(nkeys, nv))
    if session is not None and direct:
        cnt = _cached_count_volume(session, shape, ranges, stride, win_size, dev)
    else:
        cnt = count_volume(shape, ranges, dev)
    if hasattr(ops, "gather_all") and GATHER_STITCH:
        acc_buf = torch.empty((nkeys,) + shape, dtype=torch.float32, device=dev)
        if index is not None:
            ops.gather_all(acc_buf, srcs, ranges, shape, index=index)
        else:
            ops.gather_all(acc_buf, srcs, ranges, shape)      # one launch: sum in tile order, / count, write once
    else:
        acc_buf = torch.zeros((nkeys,) + shape, dtype=torch.float32, device=dev)
        for rng, rows in zip(ranges, srcs):
            if direct:
                ops.add_all(acc_buf, rows, rng, shape)
            else:
                for j in range(nkeys):
                    ops.add(acc_buf[j], rows[j], rng, shape)
        if direct:
            ops.finalize_all(acc_buf, cnt)
        else:
            for j in range(nkeys):
                ops.finalize(acc_buf[j], cnt)
    acc = OrderedDict((k, acc_buf[j]) for j, k in enumerate(keys))
    return acc, ranges, cnt
