# kernel trace of scripts/bench_config5.py (generator item -> training iteration).  usage: bash scripts/prof_config5.sh <tag> [items=2] [size=160]
tag=$1; IT=${2:-2}; N=${3:-160}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pc5_$tag -o tr --output-format rocpd -- python3 $R/scripts/bench_config5.py $IT $N > $R/gpurun_out/config5_${tag}_prof.log 2>&1
db=$(find /tmp/pc5_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 1 450 > gpurun_out/config5_${tag}_trace.txt 2>&1
head -70 gpurun_out/config5_${tag}_trace.txt
