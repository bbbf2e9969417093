"""PARITY UNPINNED (nibabel absent): this fixture does NOT come from the reference's reader (utils/misc.py:194-222 MRIread /
MRIwrite = nibabel), which cannot be imported in this image.  It is the closest available substitute and is labelled as such
in DESIGN.md section 9 and README.md: row N3 stays "partial" until a nibabel-made fixture exists.

Pin brainfm_amd.volio to the one real volume file the reference's hot path reads:
files/gca.mgz (utils/test_utils.py:38-43 -> MNI, aff2).  nibabel is absent from this image, so the expected values are
produced HERE by an independent parse of the FreeSurfer MGH format definition (gzip + struct, nothing from volio):

  offset 0    int32  version (=1)            offset 4..16  int32 width, height, depth, nframes
  offset 20   int32  type (0 uchar, 1 int, 3 float, 4 short)      offset 24 int32 dof
  offset 28   int16  goodRASFlag             offset 30  3 x float32 spacing
  offset 42   9 x float32 direction cosines (x_r x_a x_s y_r ... stored column by column)   offset 78  3 x float32 c_ras
  offset 284  voxel data, big endian, x fastest (Fortran order)
  vox2ras = [Mdc * diag(spacing) | c_ras - Mdc * diag(spacing) * (dims / 2)]

Run in the build container only:  python tests/golden/make_golden_volio.py  ->  tests/golden/volio_gca.npz
(header bytes, the affine, a checksum of the voxel payload and an 8x-subsampled copy of the volume: data, no source).
"""
import gzip
import hashlib
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.environ.get("BRAINFM_REFERENCE", "/root/reference"), "files", "gca.mgz")

raw = gzip.open(SRC, "rb").read()
version, w, h, d, nf, typ, dof = struct.unpack(">7i", raw[:28])
good, = struct.unpack(">h", raw[28:30])
spacing = np.array(struct.unpack(">3f", raw[30:42]), dtype=np.float64)
mdc = np.array(struct.unpack(">9f", raw[42:78]), dtype=np.float64).reshape(3, 3).T      # columns = x, y, z cosines
cras = np.array(struct.unpack(">3f", raw[78:90]), dtype=np.float64)
assert version == 1 and good == 1
dt = {0: ">u1", 1: ">i4", 3: ">f4", 4: ">i2"}[typ]
n = w * h * d * nf
vox = np.frombuffer(raw, dtype=dt, count=n, offset=284).reshape((w, h, d) if nf == 1 else (w, h, d, nf), order="F")
M = mdc * spacing[None, :]
aff = np.eye(4)
aff[:3, :3] = M
aff[:3, 3] = cras - M @ (np.array([w, h, d]) / 2.0)
out = {"header": np.frombuffer(raw[:284], dtype=np.uint8), "dims": np.array([w, h, d, nf]), "type": np.array(typ),
       "affine": aff, "sha256_be_payload": np.frombuffer(hashlib.sha256(raw[284:284 + n * np.dtype(dt).itemsize]).digest(),
                                                         dtype=np.uint8),
       "sub8": np.ascontiguousarray(vox[::8, ::8, ::8]).astype(np.float32),
       "sum": np.array(float(vox.astype(np.float64).sum())), "max": np.array(float(vox.max()))}
np.savez_compressed(os.path.join(HERE, "volio_gca.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()}, "\naffine\n", aff, "\ntype", typ, "sum", out["sum"])
