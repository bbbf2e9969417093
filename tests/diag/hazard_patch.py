"""Round 6: assembly-level bisection of the grid_pull3d-beside-conv_wino4d hazard (HISTORY.md section 3.3).

Compiles brainfm_amd/csrc/synth_interp.hip to gfx950 assembly exactly as the library build does, then writes variants of the
code object in which ONLY grid_pull3d is edited (s_nop's behind one instruction class at a time) into
tests/diag/hazard_variants/<name>.hsaco; tests/diag/diag_hazard_r6.py launches them through hipModuleLaunchKernel in place of
the library's kernel (BFM_DIAG_HSACOS=...).  Runs on the build container (no GPU needed).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from brainfm_amd import build as B   # noqa: E402

OUT = os.path.join(ROOT, "tests", "diag", "hazard_variants")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN12_GLOBAL__N_111grid_pull3dEPKfiiiiiS1_iliiiiiPf"
NOP = "\ts_nop 7\n\ts_nop 7\n"


def device_asm(extra=()):
    s = os.path.join(OUT, "synth_interp.s")
    # the variants bisect the FAILING build: the library's flags of rounds 1-5, i.e. WITH packed-FP32 instructions ("nopk" adds
    # the switch back)
    flags = list(B.FLAGS)
    i = flags.index(B.NO_PACKED_FP32[0])
    del flags[i:i + len(B.NO_PACKED_FP32)]
    cmd = [B.HIPCC] + [f for f in flags if f != "-Wall"] + list(extra) + ["-S", "--cuda-device-only",
           os.path.join(B.CSRC, "synth_interp.hip"), "-o", s]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(s).read()


def split_kernel(asm):
    a = asm.index(KERNEL + ":")
    b = asm.index(".Lfunc_end", a)
    return asm[:a], asm[a:b], asm[b:]


def after(body, pattern, what=NOP, lo=None, hi=None):
    out = []
    lines = body.split("\n")
    for n, line in enumerate(lines):
        out.append(line)
        if re.match(r"\s+(%s)\b" % pattern, line) and (lo is None or lo <= n < hi):
            out.append(what.rstrip("\n"))
    return "\n".join(out)


def assemble(name, asm):
    if name == "dump":          # v76..v79: inside the 80 registers the wave is allocated anyway (granule of 8)
        a = asm.index(".amdhsa_kernel " + KERNEL)
        b = asm.index(".end_amdhsa_kernel", a)
        kd = asm[a:b].replace(".amdhsa_next_free_vgpr 75", ".amdhsa_next_free_vgpr 80").replace(".amdhsa_accum_offset 76", ".amdhsa_accum_offset 80")
        assert kd != asm[a:b]
        asm = asm[:a] + kd + asm[b:]
    s = os.path.join(OUT, name + ".s")
    open(s, "w").write(asm)
    o = os.path.join(OUT, name + ".o")
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", os.path.join(OUT, name + ".hsaco")])
    os.remove(o)
    os.remove(s)


def main():
    os.makedirs(OUT, exist_ok=True)
    head, body, tail = split_kernel(device_asm())
    lines = body.split("\n")
    loads = [n for n, l in enumerate(lines) if "global_load_dword" in l]
    assert len(loads) == 11, loads                                              # 3 coordinates + 8 corners
    loop = max(n for n in range(loads[3]) if re.match(r"\.LBB\d+_\d+:", lines[n]))   # the channel loop's label
    loop_end = next(n for n in range(loop, len(lines)) if "s_cbranch_scc0" in lines[n])
    assert loop_end > loads[-1]
    blk = max(n for n in range(loop) if re.match(r"\.LBB\d+_\d+:", lines[n]))          # the straight-line block in front of it
    variants = {
        "base": body,
        "mul": after(body, "v_mul_lo_u32|v_mul_hi_u32"),
        "cvt": after(body, "v_cvt_f32_i32_e32"),
        "pk": after(body, "v_pk_mul_f32|v_pk_add_f32"),
        "mad": after(body, "v_mad_u64_u32|v_mad_i64_i32"),
        "lshl": after(body, "v_lshl_add_u64"),
        "ld": after(body, "global_load_dword", lo=loop, hi=loop_end + 1),
        "loopall": after(body, r"[vs]_\w+|global_\w+", lo=loop, hi=loop_end),
        "all": after(body, "v_mul_lo_u32|v_mul_hi_u32|v_cvt_f32_i32_e32|v_pk_mul_f32|v_pk_add_f32|v_mad_u64_u32|v_mad_i64_i32|v_lshl_add_u64"),
        "blk": after(body, r"v_\w+", lo=blk, hi=loop),
        "rest": after(body, r"v_\w+", lo=0, hi=blk),
        "pre": "\n".join(lines[:loop + 1] + ["\ts_nop 7"] * 4 + lines[loop + 1:]),
    }
    # "dump": behind the loop's store, the raw bits of the registers that hold the corner signs / weights go to out + k * 0x8000
    # (the caller allocates 9 x 32 KiB); nothing in front of the loop changes
    st = next(n for n in range(loop, loop_end) if "global_store_dword" in lines[n])
    regs = ["v20", "v12", "v22", "v24", "v23", "v25", "v10", "v26"]     # sign c2, sign c4, weight c2, c4, c3, c5, c0, c6
    ins = ["\tv_mov_b32_e32 v78, 0x8000", "\tv_mov_b32_e32 v79, 0", "\tv_lshl_add_u64 v[76:77], v[40:41], 0, v[78:79]"]
    for k, r in enumerate(regs):
        ins.append("\tglobal_store_dword v[76:77], %s, off" % r)
        if k + 1 < len(regs):
            ins.append("\tv_lshl_add_u64 v[76:77], v[76:77], 0, v[78:79]")
    dump = "\n".join(lines[:st + 1] + ins + lines[st + 1:])
    variants["dump"] = dump
    # "aggr": the stand-alone aggressor (tests/diag/hazard_aggressor.hip) as a code object of its own
    src = os.path.join(ROOT, "tests", "diag", "hazard_aggressor.hip")
    subprocess.check_call([B.HIPCC, "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", os.path.join(OUT, "aggr.hsaco")],
                          stderr=subprocess.DEVNULL)
    print("aggr: hazard_aggressor code object")
    # "nopk": the same source compiled with the packed-FP32 instructions switched off (no v_pk_*_f32 anywhere)
    nopk = device_asm(["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"])
    assert not re.search(r"v_pk_\w+_f32", nopk)
    assemble("nopk", nopk)
    print("nopk: no packed-FP32 instruction in the code object")
    for name, b in variants.items():
        assemble(name, head + b + tail)
        print(name, b.count("s_nop 7") - body.count("s_nop 7"), "nops added")


if __name__ == "__main__":
    main()
