"""HBM traffic of the synthesis kernels and of whole generator items from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; separate passes, --kernel-trace only) over scripts/synth_cold_once.py.  Units: KB per dispatch.  On gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): `fetch_bytes` is the
doubled value, `fetch_bytes_raw` the counter's own; scattered 8-byte gathers are uncalibrated, so for the gather kernels
the truth lies between the two.  Infinity-Cache hits are counted by these memory-side counters, which is why every
launch of a section works on its own buffers (>= 1.2 GB per section).
usage: python scripts/pmc_synth.py <fetch counter_collection.csv> <write counter_collection.csv> <sections.json> <out.json>"""
import csv
import json
import sys


def sections_of(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "bbox_init_kernel" in r["Kernel_Name"]]
    out = []
    for a, b in zip(marks[0::2], marks[1::2]):
        out.append([r for r in rows[a + 1:b] if "bbox_kernel" not in r["Kernel_Name"] and "bbox_init_kernel" not in r["Kernel_Name"]])
    return out


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()[:60] or name[:60]


fetch = sections_of(sys.argv[1], "FETCH_SIZE")
write = sections_of(sys.argv[2], "WRITE_SIZE")
names = json.load(open(sys.argv[3]))
assert len(fetch) == len(write) == len(names), (len(fetch), len(write), len(names))
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes, program directly after --) on "
                 "scripts/synth_cold_once.py; FETCH_SIZE doubled (gfx950 wide-read correction), raw value kept beside it; every launch of a "
                 "kernel section on its own buffers (cold)", "sections": {}}
for meta, fr, wr in zip(names, fetch, write):
    e = {"calls": meta["calls"], "dispatches": len(fr)}
    fb = sum(float(r["Counter_Value"]) for r in fr) * 1024.0
    wb = sum(float(r["Counter_Value"]) for r in wr) * 1024.0
    e["fetch_bytes_raw_per_call"] = fb / meta["calls"]
    e["fetch_bytes_per_call"] = 2.0 * fb / meta["calls"]
    e["write_bytes_per_call"] = wb / meta["calls"]
    e["hbm_bytes_per_call"] = e["fetch_bytes_per_call"] + e["write_bytes_per_call"]
    if "algorithmic_bytes_per_call" in meta:
        e["algorithmic_bytes_per_call"] = meta["algorithmic_bytes_per_call"]
        e["hbm_over_algorithmic"] = e["hbm_bytes_per_call"] / meta["algorithmic_bytes_per_call"]
        e["hbm_over_algorithmic_raw_fetch"] = (fb + wb) / meta["calls"] / meta["algorithmic_bytes_per_call"]
    if "steps" in meta:
        e["dopri5_steps"] = meta["steps"]
    per = {}
    for rows, key, mult in ((fr, "fetch_bytes", 2.0), (wr, "write_bytes", 1.0)):
        for r in rows:
            k = per.setdefault(short(r["Kernel_Name"]), {"dispatches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
            k[key] += float(r["Counter_Value"]) * 1024.0 * mult
            if key == "fetch_bytes":
                k["dispatches"] += 1
    e["kernels"] = dict(sorted(per.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))[:12])
    res["sections"][meta["section"]] = e
json.dump(res, open(sys.argv[4], "w"), indent=1)
for k, e in res["sections"].items():
    print("%-20s calls %3d  dispatches %4d  HBM %9.1f MB/call (fetch x2 %9.1f + write %8.1f)%s" % (
        k, e["calls"], e["dispatches"], e["hbm_bytes_per_call"] / 1e6, e["fetch_bytes_per_call"] / 1e6, e["write_bytes_per_call"] / 1e6,
        "  = %.2fx algorithmic (%.2fx with the raw fetch counter)" % (e["hbm_over_algorithmic"], e["hbm_over_algorithmic_raw_fetch"])
        if "hbm_over_algorithmic" in e else ""))
