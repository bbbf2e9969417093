"""Build a DIAGNOSTICS variant of libbrainfm_hip.so: the named sources recompiled with extra -D flags, every other object
taken from the regular build.  Never loaded by the product path (brainfm_amd/_lib.py loads libbrainfm_hip.so only); the diag
scripts select it with BFM_DIAG_LIB.

    python scripts/build_variant.py ablate conv3d_wino4.hip -DBFM_W4_ABLATE
    python scripts/build_variant.py diag_pk synth_interp.hip -DBFM_DIAG --packed-fp32
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brainfm_amd import build as B   # noqa: E402


def main():
    tag, rest = sys.argv[1], sys.argv[2:]
    srcs = [a for a in rest if a.endswith(".hip")]
    flags = [a for a in rest if not a.endswith(".hip") and a != "--packed-fp32"]
    base_flags = list(B.FLAGS)
    if "--packed-fp32" in rest:          # the named sources WITH packed-FP32 instructions (the pre-round-6 build), for the
        i = base_flags.index(B.NO_PACKED_FP32[0])      # with / without comparisons of profiles/r06_hazard_root_cause.txt
        del base_flags[i:i + len(B.NO_PACKED_FP32)]
    B.build(verbose=False)
    objdir = os.path.join(B.OBJ, "variant_" + tag)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for s in B.sources():
        base = os.path.basename(s)
        o = os.path.join(B.OBJ, base[:-4] + ".o")
        if base in srcs:
            o = os.path.join(objdir, base[:-4] + ".o")
            cmd = [B.HIPCC] + base_flags + flags + ["-c", s, "-o", o]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        objs.append(o)
    lib = os.path.join(B.HERE, "libbrainfm_hip_%s.so" % tag)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
