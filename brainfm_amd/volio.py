"""Volume file I/O at the boundary of the inference / synthesis path: NIfTI-1 (.nii, .nii.gz) and
FreeSurfer MGH (.mgh, .mgz), read and write, without nibabel.

Mirrors what the reference does through nibabel: ``utils/misc.py:194-222`` (``MRIread`` -> ``nib.load(f).get_fdata()``
+ ``.affine``; ``MRIwrite`` -> ``nib.save(nib.Nifti1Image(volume, aff, nib.Nifti1Header()), f)``) and the generator's
cropped reads ``nib.load(f).dataobj[x1:x2, y1:y2, z1:z2]`` (``Generator/utils.py:296-305``).  The formats are public:
NIfTI-1 (348-byte header, data in Fortran order, sform / qform / pixdim affines, optional scl_slope / scl_inter) and
MGH (big-endian, 284-byte header, Mdc / c_ras geometry).  Host code (NumPy); the arrays go to the device in
``brainfm_amd.test_utils.prepare_image`` / the generator.

Parity note: nibabel is absent from this image, so these readers are pinned to the format definitions (hand-built
headers in tests/test_host_cpu.py), not to nibabel output.
"""
import gzip
import os
import struct
import zlib

import numpy as np

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                 768: np.uint32, 1024: np.int64, 1280: np.uint64}
_NIFTI_CODES = {np.dtype(v).str[1:]: k for k, v in _NIFTI_DTYPES.items()}
_MGH_DTYPES = {0: ">u1", 1: ">i4", 3: ">f4", 4: ">i2"}
_MGH_CODES = {"u1": 0, "i4": 1, "f4": 3, "i2": 4}


class _ParallelGzipWriter:
    """Write-only gzip file built from independent members compressed in a thread pool (zlib releases the GIL): a
    multi-member stream is plain RFC 1952 gzip, read back by gzip / nibabel / zcat like any other.  Level 1 is
    nibabel's own default for .nii.gz / .mgz (nibabel.openers.Opener.default_compresslevel); python's gzip.open default
    (9) took 23 s for one 256^3 float32 map -- 17 stitched maps per volume made writing the output cost 400 s against
    0.14 s of inference (reference quirk Q5 made worse).  With 8 threads: 0.3 s per map."""

    CHUNK = 4 << 20

    def __init__(self, filename, level=1, threads=None):
        self._f = open(filename, "wb")
        self._buf = []
        self._level = level
        self._threads = threads or min(16, os.cpu_count() or 1)

    def write(self, data):
        self._buf.append(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data)
        return len(data)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._f is None:
            return
        raw = b"".join(self._buf)
        self._buf = []
        view = memoryview(raw)
        chunks = [view[i:i + self.CHUNK] for i in range(0, len(raw), self.CHUNK)] or [view]
        level = self._level

        def comp(c):
            z = zlib.compressobj(level, zlib.DEFLATED, 31)          # wbits 31: gzip container
            return z.compress(c) + z.flush()

        if len(chunks) > 1 and self._threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(self._threads) as ex:
                parts = list(ex.map(comp, chunks))
        else:
            parts = [comp(c) for c in chunks]
        for p_ in parts:
            self._f.write(p_)
        self._f.close()
        self._f = None


def _open(filename, mode="rb"):
    if filename.endswith((".gz", ".mgz")):
        if "w" in mode:
            return _ParallelGzipWriter(filename)
        return gzip.open(filename, mode)
    return open(filename, mode)


# ----------------------------------------------------------------------------- NIfTI-1
def _quat_to_rot(b, c, d):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 1e-7 else 0.0
    if a2 <= 1e-7:                                   # special case: 180 degree rotation, renormalise
        s = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d = b * s, c * s, d * s
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])


def _rot_to_quat(R):
    """Quaternion (b, c, d) with a >= 0 of a proper rotation matrix (NIfTI-1 nifti_mat44_to_quatern)."""
    a = R[0, 0] + R[1, 1] + R[2, 2] + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b = 0.25 * (R[2, 1] - R[1, 2]) / a
        c = 0.25 * (R[0, 2] - R[2, 0]) / a
        d = 0.25 * (R[1, 0] - R[0, 1]) / a
    else:
        xd = 1.0 + R[0, 0] - (R[1, 1] + R[2, 2])
        yd = 1.0 + R[1, 1] - (R[0, 0] + R[2, 2])
        zd = 1.0 + R[2, 2] - (R[0, 0] + R[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c = 0.25 * (R[0, 1] + R[1, 0]) / b
            d = 0.25 * (R[0, 2] + R[2, 0]) / b
            a = 0.25 * (R[2, 1] - R[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b = 0.25 * (R[0, 1] + R[1, 0]) / c
            d = 0.25 * (R[1, 2] + R[2, 1]) / c
            a = 0.25 * (R[0, 2] - R[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b = 0.25 * (R[0, 2] + R[2, 0]) / d
            c = 0.25 * (R[1, 2] + R[2, 1]) / d
            a = 0.25 * (R[1, 0] - R[0, 1]) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return b, c, d


class _Nifti1:
    def __init__(self, filename):
        self.filename = filename
        with _open(filename) as f:
            hdr = f.read(348)
        if len(hdr) < 348:
            raise ValueError("%s: truncated NIfTI-1 header" % filename)
        for end in ("<", ">"):
            if struct.unpack(end + "i", hdr[0:4])[0] == 348:
                self.end = end
                break
        else:
            raise ValueError("%s: not a NIfTI-1 file (sizeof_hdr != 348)" % filename)
        e = self.end
        if hdr[344:347] not in (b"n+1", b"ni1"):
            raise ValueError("%s: bad NIfTI-1 magic %r" % (filename, hdr[344:348]))
        if hdr[344:347] == b"ni1":
            raise NotImplementedError("%s: two-file NIfTI (.hdr/.img) is not supported" % filename)
        dim = struct.unpack(e + "8h", hdr[40:56])
        self.shape = tuple(int(v) for v in dim[1:1 + dim[0]])
        self.datatype, self.bitpix = struct.unpack(e + "2h", hdr[70:74])
        if self.datatype not in _NIFTI_DTYPES:
            raise NotImplementedError("%s: NIfTI datatype %d" % (filename, self.datatype))
        self.dtype = np.dtype(_NIFTI_DTYPES[self.datatype]).newbyteorder(e)
        pixdim = struct.unpack(e + "8f", hdr[76:108])
        self.vox_offset = int(struct.unpack(e + "f", hdr[108:112])[0])
        self.scl_slope, self.scl_inter = struct.unpack(e + "2f", hdr[112:120])
        qform_code, sform_code = struct.unpack(e + "2h", hdr[252:256])
        qb, qc, qd, qx, qy, qz = struct.unpack(e + "6f", hdr[256:280])
        srow = np.array(struct.unpack(e + "12f", hdr[280:328]), dtype=np.float64).reshape(3, 4)
        aff = np.eye(4)
        if sform_code > 0:                                   # nibabel's get_best_affine: sform, then qform, then pixdim
            aff[:3, :] = srow
        elif qform_code > 0:
            R = _quat_to_rot(float(qb), float(qc), float(qd))
            qfac = -1.0 if pixdim[0] < 0 else 1.0
            zooms = np.array([pixdim[1], pixdim[2], pixdim[3] * qfac], dtype=np.float64)
            aff[:3, :3] = R * zooms
            aff[:3, 3] = [qx, qy, qz]
        else:
            zooms = np.array([abs(v) if v != 0 else 1.0 for v in pixdim[1:4]], dtype=np.float64)
            aff[:3, :3] = np.diag(zooms)
            aff[:3, 3] = -0.5 * (np.array(self.shape[:3] + (1,) * (3 - len(self.shape[:3]))) - 1) * zooms
            aff[0, :] *= -1                                  # nibabel's base affine flips x (radiological default)
        self.affine = aff

    def _scale(self, arr):
        s, i = self.scl_slope, self.scl_inter
        if s is None or np.isnan(s) or s == 0 or (s == 1 and (np.isnan(i) or i == 0)):
            return arr.astype(np.float64)
        return arr.astype(np.float64) * float(s) + (0.0 if np.isnan(i) else float(i))

    def _raw(self):
        n = int(np.prod(self.shape))
        with _open(self.filename) as f:
            f.seek(self.vox_offset)
            buf = f.read(n * self.dtype.itemsize)
        return np.frombuffer(buf, dtype=self.dtype, count=n).reshape(self.shape, order="F")

    def get_fdata(self):
        return self._scale(self._raw())

    @property
    def dataobj(self):
        return _Sliceable(self)


class _Sliceable:
    """``img.dataobj[x1:x2, y1:y2, z1:z2]``: memory-mapped for uncompressed files, full read otherwise."""

    def __init__(self, img):
        self.img = img

    def __array__(self, dtype=None, copy=None):
        """np.asarray(img.dataobj): the whole volume in its stored type (scaled like get_fdata when the header says so)."""
        arr = self[tuple(slice(None) for _ in self.img.shape)]
        return arr if dtype is None else arr.astype(dtype)

    def __getitem__(self, idx):
        img = self.img
        if isinstance(img, _Nifti1) and not img.filename.endswith(".gz"):
            mm = np.memmap(img.filename, dtype=img.dtype, mode="r", offset=img.vox_offset, shape=img.shape, order="F")
            return img._scale(np.asarray(mm[idx]))
        return img.get_fdata()[idx]


def _write_nifti1(volume, aff, filename):
    vol = np.asarray(volume)
    if vol.dtype == np.bool_:
        vol = vol.astype(np.uint8)
    key = vol.dtype.str[1:]
    if key not in _NIFTI_CODES:
        raise NotImplementedError("MRIwrite: dtype %s" % vol.dtype)
    aff = np.eye(4) if aff is None else np.asarray(aff, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [vol.ndim] + list(vol.shape) + [1] * (7 - vol.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<2h", hdr, 70, _NIFTI_CODES[key], vol.dtype.itemsize * 8)
    RZS = aff[:3, :3]
    zooms = np.sqrt(np.sum(RZS * RZS, axis=0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        qfac = -1.0
        R = R.copy()
        R[:, 2] *= -1
    # nearest proper rotation (polar decomposition), as nibabel does before taking the quaternion
    U, _, Vt = np.linalg.svd(R)
    b, c, d = _rot_to_quat(U @ Vt)
    pixdim = [qfac] + [float(z) for z in zooms] + [1.0] * 4
    struct.pack_into("<8f", hdr, 76, *pixdim)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, float("nan"), float("nan"))       # no intensity scaling
    struct.pack_into("<2h", hdr, 252, 0, 2)                              # qform 'unknown', sform 'aligned' (nibabel defaults)
    struct.pack_into("<6f", hdr, 256, b, c, d, aff[0, 3], aff[1, 3], aff[2, 3])
    struct.pack_into("<12f", hdr, 280, *aff[:3, :].reshape(-1))
    hdr[344:348] = b"n+1\x00"
    with _open(filename, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\x00\x00\x00\x00")                                     # extension flag -> data at byte 352
        if vol.dtype.byteorder == ">":
            vol = vol.astype(vol.dtype.newbyteorder("<"))
        # the file holds x fastest: a Fortran-contiguous array goes out as it lies in memory (its transpose is a
        # C-contiguous view), anything else is transposed once; no .tobytes() copy on top
        data = vol if vol.flags.f_contiguous else np.asfortranarray(vol)
        f.write(memoryview(np.ascontiguousarray(data.T)).cast("B"))


# ----------------------------------------------------------------------------- MGH / MGZ
class _Mgh:
    def __init__(self, filename):
        self.filename = filename
        with _open(filename) as f:
            hdr = f.read(284)
        version, w, h, d, nf, typ, dof = struct.unpack(">7i", hdr[0:28])
        if version != 1:
            raise ValueError("%s: not an MGH file (version %d)" % (filename, version))
        if typ not in _MGH_DTYPES:
            raise NotImplementedError("%s: MGH type %d" % (filename, typ))
        self.shape = (w, h, d) if nf == 1 else (w, h, d, nf)
        self.dtype = np.dtype(_MGH_DTYPES[typ])
        good = struct.unpack(">h", hdr[28:30])[0]
        if good:
            delta = np.array(struct.unpack(">3f", hdr[30:42]), dtype=np.float64)
            Mdc = np.array(struct.unpack(">9f", hdr[42:78]), dtype=np.float64).reshape(3, 3).T   # columns x, y, z
            c_ras = np.array(struct.unpack(">3f", hdr[78:90]), dtype=np.float64)
        else:
            delta = np.ones(3)
            Mdc = np.array([[-1., 0., 0.], [0., 0., 1.], [0., -1., 0.]])
            c_ras = np.zeros(3)
        M = Mdc * delta
        aff = np.eye(4)
        aff[:3, :3] = M
        aff[:3, 3] = c_ras - M @ (np.array([w, h, d], dtype=np.float64) / 2.0)
        self.affine = aff

    def get_fdata(self):
        n = int(np.prod(self.shape))
        with _open(self.filename) as f:
            f.seek(284)
            buf = f.read(n * self.dtype.itemsize)
        return np.frombuffer(buf, dtype=self.dtype, count=n).reshape(self.shape, order="F").astype(np.float64)

    @property
    def dataobj(self):
        return _Sliceable(self)


def _write_mgh(volume, aff, filename):
    vol = np.asarray(volume)
    key = vol.dtype.str[1:]
    if key not in _MGH_CODES:
        vol = vol.astype(np.float32)
        key = "f4"
    aff = np.eye(4) if aff is None else np.asarray(aff, dtype=np.float64)
    w, h, d = vol.shape[:3]
    nf = vol.shape[3] if vol.ndim > 3 else 1
    M = aff[:3, :3]
    delta = np.sqrt(np.sum(M * M, axis=0))
    Mdc = M / delta
    c_ras = aff[:3, 3] + M @ (np.array([w, h, d], dtype=np.float64) / 2.0)
    hdr = bytearray(284)
    struct.pack_into(">7i", hdr, 0, 1, w, h, d, nf, _MGH_CODES[key], 0)
    struct.pack_into(">h", hdr, 28, 1)
    struct.pack_into(">3f", hdr, 30, *delta)
    struct.pack_into(">9f", hdr, 42, *Mdc.T.reshape(-1))
    struct.pack_into(">3f", hdr, 78, *c_ras)
    with _open(filename, "wb") as f:
        f.write(bytes(hdr))
        f.write(np.asfortranarray(vol.astype(vol.dtype.newbyteorder(">"))).tobytes(order="F"))


# ----------------------------------------------------------------------------- reference-shaped entry points
def load(filename):
    """``nib.load`` stand-in: an object with ``.shape``, ``.affine``, ``.get_fdata()`` and a sliceable ``.dataobj``."""
    if filename.endswith((".nii", ".nii.gz")):
        return _Nifti1(filename)
    if filename.endswith((".mgz", ".mgh")):
        return _Mgh(filename)
    raise ValueError("Unknown data file: %s" % filename)


def MRIread(filename, dtype=None, im_only=False):
    """utils/misc.py:208-222."""
    assert filename.endswith((".nii", ".nii.gz", ".mgz")), "Unknown data file: %s" % filename
    x = load(filename)
    volume = x.get_fdata()
    aff = x.affine
    if dtype is not None:
        volume = volume.astype(dtype=dtype)
    if im_only:
        return volume
    return volume, aff


def MRIwrite(volume, aff, filename, dtype=None):
    """utils/misc.py:194-204 (always NIfTI-1 in the reference; .mgz / .mgh written as MGH here)."""
    volume = np.asarray(volume)
    if dtype is not None:
        volume = volume.astype(dtype=dtype)
    if filename.endswith((".mgz", ".mgh")):
        _write_mgh(volume, aff, filename)
    else:
        _write_nifti1(volume, aff, filename)


def write_device_volumes(volumes, aff, directory, ext=".nii", threads=None, names=None):
    """Write a dict of device (or host) volumes {name: (D,H,W) tensor} as <directory>/<name><ext> -- the tail of
    scripts/demo_test.py:108-119 (one utils.MRIwrite per output key).  The axis reversal NIfTI wants (x fastest) is done
    on the device, the copy to the host lands in one pinned staging buffer per map, and the files are written from a
    thread pool (file writes and zlib release the GIL).  'label'-like integer maps keep their dtype."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(directory, exist_ok=True)
    items = list(volumes.items())
    host = []
    for k, v in items:
        if isinstance(v, torch.Tensor):
            t = v
            if t.dtype == torch.int64:
                t = t.to(torch.int32)
            if t.is_cuda:
                tt = t.permute(*reversed(range(t.dim()))).contiguous()          # (W,H,D) C-order == (D,H,W) F-order
                pin = torch.empty(tt.shape, dtype=tt.dtype, pin_memory=True)
                pin.copy_(tt, non_blocking=True)
                host.append((k, pin, True))
            else:
                host.append((k, t.numpy(), False))
        else:
            host.append((k, np.asarray(v), False))
    if any(h[2] for h in host):
        torch.cuda.synchronize()
    paths = []

    def one(entry):
        k, a, rev = entry
        arr = a.numpy().T if rev else a                                         # .T of the reversed copy: F-contiguous view
        path = os.path.join(directory, (names[k] if names else k) + ext)
        MRIwrite(arr, aff, path)
        return path

    with ThreadPoolExecutor(threads or min(8, os.cpu_count() or 1)) as ex:
        paths = list(ex.map(one, host))
    return paths
