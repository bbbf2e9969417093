# kernel-only durations of scripts/bench_synth_micro.py.  usage: bash scripts/prof_micro.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pm_$tag -o m --output-format rocpd -- python3 $R/scripts/bench_synth_micro.py > $R/gpurun_out/micro_${tag}.log 2>&1
db=$(find /tmp/pm_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 1 > gpurun_out/micro_${tag}_trace.txt 2>&1
head -30 gpurun_out/micro_${tag}_trace.txt
