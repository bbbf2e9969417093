"""CPU: the oracle's restatement of the generator chains (oracle/gen_ref.py) against tests/golden/gen_chain.npz, the
outputs of the reference's own BrainIDGen / BaseGen __getitem__ (make_golden_gen.py), with the reference's torch draws
replayed and NumPy's stream reproduced by the seed."""
import json
import random

import numpy as np
import pytest

from conftest import load_npz
from oracle import gen_ref as G


def load_case(tag, d=None):
    d = d if d is not None else load_npz("gen_chain.npz")
    pre = tag + "/"
    case = {k: d[pre + "case/" + k] for k in ("Gen", "T1", "segmentation")}
    case["distance"] = [d[pre + "case/distance%d" % j] for j in range(4)]
    case["registration"] = [d[pre + "case/registration%d" % j] for j in range(3)]
    n = int(d[pre + "ndraws"])
    draws = []
    for i in range(n):
        kind = "randn" if (pre + "draw%03d_randn" % i) in d else "rand"
        draws.append((kind, d[pre + "draw%03d_%s" % (i, kind)]))
    cfg = json.loads(str(d[pre + "cfg_json"]))
    target = {k[len(pre) + 7:]: d[k] for k in d if k.startswith(pre + "target/")}
    samples = []
    i = 0
    while (pre + "sample%d/input" % i) in d:
        samples.append({k.split("/")[-1]: d[k] for k in d if k.startswith(pre + "sample%d/" % i)})
        i += 1
    return dict(case=case, draws=draws, cfg=cfg, target=target, samples=samples, seed=int(d[pre + "seed"]),
                mode=str(d[pre + "mode"]), cls=str(d[pre + "cls"]), t1_prob=float(d[pre + "t1_prob"]))


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max()) / max(1e-6, float(np.abs(b).max()))


def check_item(tag, target, samples, c, tol_img, tol_t):
    assert len(samples) == len(c["samples"])
    for k, ref in c["target"].items():
        got = target[k]
        if np.ndim(ref) == 0:
            assert np.ndim(got) == 0 and float(got) == float(ref), (tag, k, got, ref)
            continue
        got = np.asarray(got)
        assert got.shape == ref.shape, (tag, k, got.shape, ref.shape)
        if k in ("segmentation", "pathology"):
            assert np.mean(got != ref) <= 1e-4, (tag, k, float(np.mean(got != ref)))
        else:
            assert relerr(got, ref) <= tol_t, (tag, k, relerr(got, ref))
    for i, (s, r) in enumerate(zip(samples, c["samples"])):
        assert sorted(s.keys()) == sorted(r.keys()), (tag, i, sorted(s.keys()), sorted(r.keys()))
        for k in r:
            assert np.asarray(s[k]).shape == r[k].shape, (tag, i, k)
            e = relerr(s[k], r[k])
            assert e <= tol_img, (tag, i, k, e)


@pytest.mark.parametrize("tag", ["A", "B", "C"])
def test_generator_chain_oracle_vs_reference_getitem(tag):
    c = load_case(tag)
    np.random.seed(c["seed"])
    random.seed(c["seed"])
    o = G.GenOracle(c["cfg"], c["case"], c["draws"], t1_prob=c["t1_prob"], brain_id=c["cls"] == "BrainIDGen")
    mode, target, samples, setups = o.getitem()
    assert mode == c["mode"]
    assert o.draws.pos == len(c["draws"])                     # every draw the reference made was consumed, in its order
    check_item(tag, target, samples, c, tol_img=2e-5, tol_t=1e-6)
