# kernel trace of scripts/bench_wgrad.py.  usage: bash scripts/prof_wgrad.sh <tag> [size=128]
tag=$1; N=${2:-128}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pw_$tag -o tr --output-format csv -- python3 $R/scripts/bench_wgrad.py $N 3 > $R/gpurun_out/wgrad_${tag}_prof.log 2>&1
f=$(find /tmp/pw_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/wgrad_${tag}_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Kernel_Name"]
    if "wgrad" in n:
        print("%-60s grid %-8s %9.1f us" % (n[:60], r.get("Grid_Size", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
tail -40 $R/gpurun_out/wgrad_${tag}_trace.txt
