"""Import harness for the upstream reference (this container only).

Only the golden-vector generators under tests/golden/ use this module. It
stubs the third-party modules the reference imports at module load time but
that are absent from this image (none of them is on the arithmetic path), and
puts /root/reference on sys.path with bytecode writing disabled so the
read-only mount stays untouched.  Nothing here travels to the GPU box:
tests read the committed .npz fixtures, never the reference.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("BRAINFM_REFERENCE", "/root/reference")
_MISSING = ["nibabel", "SimpleITK", "visdom", "torchvision", "iopath",
            "iopath.common", "iopath.common.file_io", "simplejson",
            "pytorch_msssim", "h5py", "future", "skimage", "seaborn",
            "torchvision.transforms", "torchvision.utils"]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Stub(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")


def setup():
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    for name in _MISSING:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Stub(name)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return REF_ROOT
