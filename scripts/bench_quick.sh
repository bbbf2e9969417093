# Per-kernel view of one bench run: python3 bench.py summarised (ms per step, dense volume, conv kernels).
# usage (GPU box): bash scripts/bench_quick.sh [out.json] [extra bench args...]
cd ${GRAFT_REPO_ROOT:-.}
out=${1:-gpurun_out/bench_quick.json}; shift || true
timeout -k 10 900 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fast-mode --no-config4 --no-config5 "$@" > $out 2> ${out%.json}.err || { tail -5 ${out%.json}.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$out").read().strip().splitlines()[-1])
print("ms_per_step %.2f  dense %s" % (d["ms_per_step"], (d.get("dense_volume") or {}).get("ms_per_step")))
for k, v in d["roofline"]["per_kernel"].items():
    print("  %-20s launches %3d  ms %6.2f  achieved %6.1f" % (k, v["launches_per_step"], v["ms_per_step"], v["achieved"]))
PY
