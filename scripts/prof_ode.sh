# kernel trace of scripts/bench_ode.py (both controller modes).  usage: bash scripts/prof_ode.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/po_$tag
rocprofv3 --kernel-trace --stats -d /tmp/po_$tag -o ode --output-format rocpd -- python3 $R/scripts/bench_ode.py 160 3 > $R/gpurun_out/ode_${tag}.log 2>&1
db=$(find /tmp/po_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 1 > gpurun_out/ode_${tag}_trace.txt 2>&1
cat gpurun_out/ode_${tag}.log
head -40 gpurun_out/ode_${tag}_trace.txt
