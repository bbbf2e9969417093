# kernel trace of one training iteration (scripts/bench_train.py).  usage: bash scripts/prof_train.sh <tag> [size=128]
tag=$1; N=${2:-128}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 scripts/bench_train.py $N 1 3 > gpurun_out/train_${tag}.txt 2>&1
cat gpurun_out/train_${tag}.txt | tail -12
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pt_$tag -o tr --output-format rocpd -- python3 $R/scripts/bench_train.py $N 1 2 > $R/gpurun_out/train_${tag}_prof.log 2>&1
db=$(find /tmp/pt_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 1 600 > gpurun_out/train_${tag}_trace.txt 2>&1
head -60 gpurun_out/train_${tag}_trace.txt
