"""Round 6: the gather-beside-conv_wino4d hazard (HISTORY.md section 3.3) -- aggressor-side and placement experiments.

    python tests/diag/diag_hazard_r6.py <co-runner variant 0|2|3|4> <eager|graph> [rounds]
    BFM_DIAG_LIB=brainfm_amd/libbrainfm_hip_fullexec.so   a diagnostics build of the library (scripts/build_variant.py)
    BFM_DIAG_CUMASK=none|cu|xcd|same    victims and co-runner on streams with CU masks (hipExtStreamCreateWithCUMask):
        cu   : disjoint CUs of the SAME XCDs   xcd : disjoint XCDs   same : both confined to the same half of the CUs

Victims: interpol.grid_pull (bound zero) and fast_3D_interp_torch on the golden inputs, on two streams; co-runner: six launches
of a 128 -> 128 convolution on 40^3 on a third.  Reports wrong elements, and for grid_pull WHICH corner of WHICH lanes is
missing (every wrong value seen so far is the exact sum minus one corner's term).
"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from brainfm_amd import _lib as L
if os.environ.get("BFM_DIAG_LIB"):
    L.LIB_PATH = os.path.join(ROOT, os.environ["BFM_DIAG_LIB"])
from brainfm_amd import test_utils as TU
from brainfm_amd.engine import _Layer
from brainfm_amd.generator_utils import fast_3D_interp_torch
from brainfm_amd.interpol import grid_pull
DEV = "cuda:0"
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "synth_interp.npz")))
d2 = dict(np.load(os.path.join(ROOT, "tests", "golden", "synth_grid_pull.npz")))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
X = T(d["X1"])
tile = int(os.environ.get("BFM_DIAG_TILE", "1"))
if tile > 1:        # the golden grid repeated along its first spatial axis: more (and fuller) workgroups, same expected values
    d2["grid"] = np.concatenate([d2["grid"]] * tile, axis=1)
    d2["out_zero_0"] = np.concatenate([d2["out_zero_0"]] * tile, axis=2)
vol, grid = T(d2["vol"]), T(d2["grid"])
if os.environ.get("BFM_DIAG_VOLADDR"):
    # put the volume where the LOW 32 bits of its address read as the float 1000.0: a destination register that still holds
    # its own address (load issued, data never written) would then add weight * 1000 instead of dropping the term
    big = torch.empty(5 << 30, dtype=torch.uint8, device=DEV)
    off = (0x447A0000 - big.data_ptr()) % (1 << 32)
    vol2 = big[off:off + vol.numel() * 4].view(torch.float32).view(vol.shape)
    vol2.copy_(vol)
    vol = vol2
    print("vol at 0x%x" % vol.data_ptr())
ver = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else "eager"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 400
cumask = os.environ.get("BFM_DIAG_CUMASK", "none")


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not mapped")


def masked_stream(bits):
    hip = hip_runtime()
    words = (C.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


torch.cuda.init(); torch.zeros(1, device=DEV)
if cumask == "none":
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    side = torch.cuda.Stream()
else:
    allb = range(256)
    if cumask == "cu":
        va = [i for i in allb if (i // 8) % 2 == 0]; ag = [i for i in allb if (i // 8) % 2 == 1]
    elif cumask == "xcd":
        va = [i for i in allb if i % 8 < 4]; ag = [i for i in allb if i % 8 >= 4]
    elif cumask == "same":
        va = [i for i in allb if (i // 8) % 2 == 0]; ag = va
    else:
        raise SystemExit("BFM_DIAG_CUMASK?")
    streams = [masked_stream(va), masked_stream(va)]
    side = masked_stream(ag)
ga, ta = TU.default_inference_args(f_maps=64, num_levels=6)
eng = TU.InferenceSession(ga, ta, torch.device(DEV)).engine
cin = cout = 128
cd = (40, 40, 40)
cA = torch.randn(*cd, cin, device=DEV)
csc, csh, cbd = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1, torch.full((8,), 6.0, device=DEV)
cout_t, cws = torch.empty(*cd, cout, device=DEV), torch.empty(1 << 26, dtype=torch.uint8, device=DEV)
ly = _Layer(); ly.name, ly.cin, ly.cout, ly.groups = "corunner", cin, cout, 8
ly.w_raw = (torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.05).contiguous()
ly.packs, ly.kind, ly.wpacked, ly.wexp, ly.skip = {}, None, None, 0, None
ccfg = (C.c_int * 8)()
L.check(eng.lib.bfm_conv3x3x3_mfma_plan(cin, cout, cd[0], cd[1], cd[2], ccfg), "plan")
ccfg[6] = ver


def conv_beside():
    for _ in range(6):
        eng._conv_launch(ly, cA, cin, None, 0, cd, None, csc, csh, cbd, 8, ccfg, cout_t, cws)


conv_beside(); torch.cuda.synchronize()
ref_conv = cout_t.clone()

# BFM_DIAG_AGGR=<mode>: replace the library's convolution by the stand-alone aggressor of tests/diag/hazard_aggressor.hip
# (mode bits: 1 global loads into VGPRs, 2 LDS operand reads, 4 MFMAs; 7 = the whole tap loop)
aggr_mode = int(os.environ.get("BFM_DIAG_AGGR", "0"))
if aggr_mode:
    hipA = hip_runtime()
    modA = C.c_void_p(); fnA = C.c_void_p()
    dataA = open(os.path.join(ROOT, "tests", "diag", "hazard_variants", "aggr.hsaco"), "rb").read()
    bufA = C.create_string_buffer(dataA, len(dataA))
    assert hipA.hipModuleLoadData(C.byref(modA), bufA) == 0
    assert hipA.hipModuleGetFunction(C.byref(fnA), modA, b"hazard_aggressor") == 0
    assert hipA.hipFuncSetAttribute(fnA, 8, 80 * 1024) in (0, 1) or True      # hipFuncAttributeMaxDynamicSharedMemorySize
    nfragA = 64 * 1024
    wA = torch.full((nfragA * 1024 // 4,), 1.0009765625, dtype=torch.float32, device=DEV)
    sinkA = torch.zeros(1, device=DEV)
    stepsA = int(os.environ.get("BFM_DIAG_AGGR_STEPS", "400"))

    def conv_beside():
        vals = [C.c_void_p(wA.data_ptr()), C.c_int(nfragA), C.c_int(stepsA), C.c_int(aggr_mode), C.c_void_p(sinkA.data_ptr())]
        params = (C.c_void_p * len(vals))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
        for _ in range(6):
            rc = hipA.hipModuleLaunchKernel(fnA, 1024, 1, 1, 256, 1, 1, 76800, C.c_void_p(torch.cuda.current_stream().cuda_stream), params, None)
            assert rc == 0, rc
    conv_beside(); torch.cuda.synchronize()
    ref_conv = cout_t.clone()

# corner terms of the golden pull (bound zero, extrapolate no): want[i] = sum_c term[i, c]
volh, gridh = d2["vol"], d2["grid"]
Bn, Cn, nx, ny, nz = volh.shape
gh = gridh.reshape(Bn, -1, 3); nout = gh.shape[1]
terms = np.zeros((Bn, Cn, nout, 8), np.float64)
for b in range(Bn):
    for v in range(nout):
        g = gh[b, v]; f = np.floor(g).astype(int); w = g - f.astype(np.float32)
        for c8 in range(8):
            a, bb, dd = (c8 >> 2) & 1, (c8 >> 1) & 1, c8 & 1
            i = (f[0] + a, f[1] + bb, f[2] + dd)
            if 0 <= i[0] < nx and 0 <= i[1] < ny and 0 <= i[2] < nz:
                wt = (w[0] if a else 1 - w[0]) * (w[1] if bb else 1 - w[1]) * (w[2] if dd else 1 - w[2])
                terms[b, :, v, c8] = volh[b, :, i[0], i[1], i[2]] * wt
terms = terms.reshape(-1, 8)

KERNEL = b"_ZN12_GLOBAL__N_111grid_pull3dEPKfiiiiiS1_iliiiiiPf"
hsacos = [h for h in os.environ.get("BFM_DIAG_HSACOS", "").split(",") if h]
modfn = {"f": None}


def load_variant(path):
    hip = hip_runtime()
    mod = C.c_void_p(); fn = C.c_void_p()
    data = open(os.path.join(ROOT, path), "rb").read()
    buf = C.create_string_buffer(data, len(data))
    assert hip.hipModuleLoadData(C.byref(mod), buf) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, KERNEL) == 0
    modfn["f"] = fn; modfn["keep"] = (mod, buf); modfn["hip"] = hip


def pull_module(x, g, out):
    """grid_pull3d of a code-object variant (tests/diag/hazard_patch.py), launched like bfm_grid_pull3d_linear does"""
    Bi, Cc, n_x, n_y, n_z = x.shape
    Bg = g.shape[0]; no = g.shape[1] * g.shape[2] * g.shape[3]; Bb = max(Bi, Bg)
    vals = [C.c_void_p(x.data_ptr()), C.c_int(Bi), C.c_int(Cc), C.c_int(n_x), C.c_int(n_y), C.c_int(n_z),
            C.c_void_p(g.data_ptr()), C.c_int(Bg), C.c_int64(no), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(Bb),
            C.c_void_p(out.data_ptr())]
    params = (C.c_void_p * len(vals))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
    nb = min(max((Bb * no + 255) // 256, 1), 8192)
    rc = modfn["hip"].hipModuleLaunchKernel(modfn["f"], nb, 1, 1, 256, 1, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                          params, None)
    assert rc == 0, rc


lanes = []
for lane in range(2):
    ii, jj, kk = T(d["II"]), T(d["JJ"]), T(d["KK"])
    gcopy = grid.clone()
    outs = {}

    def body(ii=ii, jj=jj, kk=kk, gcopy=gcopy, outs=outs):
        outs["interp"] = fast_3D_interp_torch(X, ii, jj, kk, "linear")
        if modfn["f"] is None:
            outs["pull_zero"] = grid_pull(vol, gcopy, interpolation="linear", bound="zero", extrapolate=False, prefilter=False)
        else:
            shp = (vol.shape[0], vol.shape[1]) + tuple(gcopy.shape[1:4])
            nel = int(np.prod(shp))
            assert nel * 4 <= 0x8000
            big = torch.zeros(9 * 0x2000, device=DEV)           # the "dump" variant's register images behind the output
            o = big[:nel].view(shp)
            pull_module(vol, gcopy, o)
            outs["pull_zero"] = o
            outs["_dump"] = big

    g = None
    with torch.cuda.stream(streams[lane]):
        body(); streams[lane].synchronize()
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[lane]):
                body()
    lanes.append((g, outs, body))
probe = None
if hasattr(L.load(), "bfm_diag_pull_probe"):
    fn = L.load().bfm_diag_pull_probe
    fn.argtypes = [C.c_void_p] * 3; fn.restype = None
    p_want = T(d2["out_zero_0"]).contiguous()
    p_rec = torch.zeros(256 * 48, device=DEV)
    p_n = torch.zeros(1, dtype=torch.int32, device=DEV)
    fn(p_want.data_ptr(), p_rec.data_ptr(), p_n.data_ptr())
    probe = (p_want, p_rec, p_n)
def measure(tag):
    global bad, bad_rounds, corner_hist, lane_hist
    bad = {"interp": 0, "pull_zero": 0, "conv": 0}
    bad_rounds = 0
    corner_hist = np.zeros(9, int)       # [8] = not explained by one missing corner
    lane_hist = {}
    shown = {}
    for it in range(rounds):
        for _, outs, _ in lanes:
            for v in outs.values():
                v.fill_(float("nan"))
        cout_t.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            conv_beside()
        for lane, (g, _, body) in enumerate(lanes):
            with torch.cuda.stream(streams[lane]):
                if g is not None:
                    g.replay()
                else:
                    body()
        torch.cuda.synchronize()
        if not torch.equal(cout_t, ref_conv):
            bad["conv"] += int((cout_t != ref_conv).sum())
        rb = False
        for lane, (_, outs, _) in enumerate(lanes):
            a = outs["interp"].cpu().numpy(); w = d["lin1"]
            m = np.flatnonzero(a != w)
            if m.size:
                bad["interp"] += m.size; rb = True
                if bad["interp"] <= 3 * m.size:
                    print("it %d lane %d interp: %d wrong at %s got %s want %s" % (it, lane, m.size, m[:6], a[m[:6]], w[m[:6]]), flush=True)
            a = outs["pull_zero"].cpu().numpy().reshape(-1); w = d2["out_zero_0"].reshape(-1)
            m = np.flatnonzero(np.abs(a - w) > 1e-6 * np.abs(w).max())
            if m.size:
                bad["pull_zero"] += m.size; rb = True
                desc = []
                corner_hist_round = np.zeros(9, int)
                for i in m:
                    diff = float(a[i]) - float(w[i])
                    c8 = int(np.argmin(np.abs(terms[i] + diff)))
                    ok = abs(terms[i, c8] + diff) <= 2e-6 * max(1.0, abs(w).max())
                    corner_hist[c8 if ok else 8] += 1
                    corner_hist_round[c8 if ok else 8] += 1
                    bb, rem = divmod(int(i), Cn * nout); cc, vv = divmod(rem, nout)
                    thread = bb * nout + vv
                    lane_hist[thread % 64] = lane_hist.get(thread % 64, 0) + 1
                    desc.append("%d(t%d w%d l%d c%s)" % (i, thread, thread // 64, thread % 64, c8 if ok else "?"))
                if "_dump" in outs and "dump" in tag:
                    img = outs["_dump"].cpu().numpy().view(np.uint32).reshape(9, 0x2000)
                    names = ["sign c2 v20", "sign c4 v12", "w c2 v22", "w c4 v24", "w c3 v23", "w c5 v25", "w c0 v10", "w c6 v26"]
                    kind = int(np.argmax(corner_hist_round))
                    if shown.get(kind, 0) < 3:
                        shown[kind] = shown.get(kind, 0) + 1
                        for i in list(m[:3]) + [int(m[0]) - 3]:
                            print("   c%d missing, element %d%s: " % (kind, i, "" if i in m else " (correct)") +
                                  "  ".join("%s=0x%08x" % (names[k], img[k + 1, i]) for k in range(8)), flush=True)
                if bad_rounds < 3:
                    print("it %d stream %d pull_zero: %d wrong: %s" % (it, lane, m.size, " ".join(desc[:16])), flush=True)
        bad_rounds += rb
    print("[%s] lib %s, co-runner ver %d, %s, cumask %s: wrong elements in %d rounds x 2 streams: %s; rounds with a wrong element: %d"
          % (tag, os.path.basename(L.LIB_PATH), ver, mode, cumask, rounds, bad, bad_rounds))
    print("   grid_pull: missing corner histogram (0..7, unexplained):", corner_hist.tolist(), " lanes:", dict(sorted(lane_hist.items())), flush=True)



abls = [a for a in os.environ.get("BFM_DIAG_ABLS", "").split(",") if a]
if abls:                 # aggressor bisection (a -DBFM_W4_ABLATE build): conv_wino4d with phases compiled out, "old" = conv_wino4
    for a in abls:
        os.environ.pop("BFM_W4_ABL", None); os.environ.pop("BFM_W4_OLD", None)
        if a == "old":
            os.environ["BFM_W4_OLD"] = "1"
        else:
            os.environ["BFM_W4_ABL"] = a
        measure("aggressor ablation " + a)
elif hsacos:
    for h in hsacos:
        load_variant(h)
        measure(os.path.basename(h))
else:
    measure("library")

if probe is not None:
    torch.cuda.synchronize()
    n = int(probe[2].item())
    rec = probe[1].cpu().numpy().reshape(256, 48)
    print("probe records: %d" % n)
    for k in range(min(n, 40)):
        r = rec[k]
        hwid = int(r[1:2].view(np.uint32)[0]); xcc = int(r[2:3].view(np.uint32)[0])
        i = int(r[0])
        print(" rec %d: thread %d (wave %d lane %d) hw_id 0x%08x (wave_id %d simd %d cu %d sh %d se %d) xcc %d  got %.7g want %.7g mask %g" %
              (k, i, i // 64, i % 64, hwid, hwid & 15, (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7, xcc & 15, r[3], r[4], r[40]))
        print("    coords %s" % r[5:8])
        print("    signs   %s" % r[8:16])
        print("    weights %s" % r[16:24])
        print("    offsets %s" % r[24:32].astype(int))
        print("    re-read %s" % r[32:40])
        b = i // nout; v = i % nout
        fl = volh[b, Cn - 1].reshape(-1)
        print("    truth   %s" % fl[np.clip(r[24:32].astype(int), 0, fl.size - 1)])
        print("    golden terms %s" % terms[(b * Cn + Cn - 1) * nout + v])
