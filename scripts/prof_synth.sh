# synthesis hot path: wall + phases of one generator item, rocprofv3 kernel trace of the item, per-kernel table of
# scripts/bench_synth.py.  usage: bash scripts/prof_synth.sh <tag>   -> gpurun_out/synth_<tag>_*.txt
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 scripts/prof_synth_item.py 5 160 1 > gpurun_out/synth_${tag}_item_wall.txt 2>&1
cat gpurun_out/synth_${tag}_item_wall.txt
python3 scripts/bench_synth.py 10 > gpurun_out/synth_${tag}_kernels.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$tag /tmp/pk_$tag
rocprofv3 --kernel-trace --stats -d /tmp/ps_$tag -o item --output-format rocpd -- python3 $R/scripts/prof_synth_item.py 3 160 0 > $R/gpurun_out/synth_${tag}_item_prof.log 2>&1
db=$(find /tmp/ps_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 3 0 bbox_kernel > gpurun_out/synth_${tag}_item_trace.txt 2>&1
head -60 gpurun_out/synth_${tag}_item_trace.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pk_$tag -o ks --output-format rocpd -- python3 $R/scripts/bench_synth.py 10 > $R/gpurun_out/synth_${tag}_kernels_prof.log 2>&1
db=$(find /tmp/pk_$tag -name "*.db" | head -1)
cd $R
python3 scripts/prof_summary.py $db 1 > gpurun_out/synth_${tag}_kernel_trace.txt 2>&1
