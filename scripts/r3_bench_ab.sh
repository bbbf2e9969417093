cd $GRAFT_REPO_ROOT
for v in 2 1; do
  echo "BFM_WINO_V=$v"
  BFM_WINO_V=$v timeout -k 10 500 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_v$v.json 2> gpurun_out/r3_bench_v$v.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r3_bench_v$v.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "dense", d.get("dense_volume", {}).get("ms_per_step"))
for k, v in d["roofline"]["per_kernel"].items():
    print("  %-20s launches %3d  ms %6.2f  achieved %6.1f" % (k, v["launches_per_step"], v["ms_per_step"], v["achieved"]))
PY
done
