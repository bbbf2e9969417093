# kernel trace of the timed bench steps.  usage: bash scripts/prof_bench.sh <tag> [env assignments...]
# writes gpurun_out/prof_<tag>_summary.txt (scripts/prof_summary.py over the last 3 steps, marker: stitch_gather)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench --output-format rocpd -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-check --no-synthesis --no-training --no-fast-mode --no-config4 --no-config5 --roofline-reps 0 > $R/gpurun_out/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 3 0 stitch_gather > gpurun_out/prof_${tag}_summary.txt 2>&1 || python3 scripts/prof_summary.py $db 3 450 > gpurun_out/prof_${tag}_summary.txt 2>&1
head -40 gpurun_out/prof_${tag}_summary.txt
