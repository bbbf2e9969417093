"""HBM traffic of the conv kernels of one bench step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes as MI355X_MICROARCH.md prescribes: the TCC block cannot hold both).  Units: KB per dispatch; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled here.  Only the dispatches of the LAST step
(between the last two step-closing stitch kernels) are kept.
usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import sys


def last_step(path, counter):
    """Dispatches of the LAST step: after the stitch kernel that closed the step before it (stitch_gather[_compact]_kernel runs once
    per volume in the batched flow; the tile-by-tile flow ends a volume with divide_multi_kernel)."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "stitch_gather" in r["Kernel_Name"] or "divide_multi_kernel" in r["Kernel_Name"]]
    if len(ends) < 2:
        raise SystemExit("need at least two steps in the trace (found %d step-closing kernels)" % len(ends))
    return rows[ends[-2] + 1:ends[-1] + 1]


def family(name):
    """bench.py's kernel names (CONV_KERNELS); longest match first."""
    if "conv_wino4_masked" in name or "conv_wino4d_masked" in name:   # bench.py counts a masked launch as conv_wino_masked
        return "conv_wino_masked"                                      # whichever variant ran
    if "conv_wino4d_uniform" in name:          # ... and a uniform-box pair as conv_wino_uniform (F(2,3) or F(4,3))
        return "conv_wino_uniform"
    for k in ("conv_upfold", "conv_wino_masked", "conv_wino_uniform", "conv_wino4", "conv_wino", "conv_mfma16", "conv_mfma_ws", "conv_mfma",
              "conv_stem", "tail_kernel"):
        if k in name:
            return k
    return None


fetch = last_step(sys.argv[1], "FETCH_SIZE")
write = last_step(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) on bench.py --steps 1 --warmup 1 "
                 "--no-graphs --roofline-reps 0 --no-cpu-baseline; last step only; FETCH_SIZE x2 (gfx950 correction); per kernel",
       "kernels": {}}
for rows, key, mult in ((fetch, "fetch_bytes", 2.0), (write, "write_bytes", 1.0)):
    for r in rows:
        # conv_wino_rest + conv_wino_uniform are the two halves of ONE bfm_conv3x3x3_wino_uniform launch (disjoint boxes):
        # bench.py counts the pair once, as conv_wino_uniform
        rest = "conv_wino_rest" in r["Kernel_Name"] or "conv_wino4d_rest" in r["Kernel_Name"]
        fam = "conv_wino_uniform" if rest else family(r["Kernel_Name"])
        if fam is None:
            continue
        e = out["kernels"].setdefault(fam, {"launches_fetch_pass": 0, "launches_write_pass": 0, "fetch_bytes": 0.0,
                                            "write_bytes": 0.0})
        e[key] += float(r["Counter_Value"]) * 1024.0 * mult
        if not rest:
            e["launches_fetch_pass" if key == "fetch_bytes" else "launches_write_pass"] += 1
for fam, e in out["kernels"].items():
    n = max(e["launches_fetch_pass"], 1)
    e["launches"] = e["launches_fetch_pass"]            # bench.py refuses this file when its own launch counts differ
    e["hbm_bytes_per_step"] = e["fetch_bytes"] + e["write_bytes"]
    e["hbm_bytes_per_launch"] = e["hbm_bytes_per_step"] / n
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
