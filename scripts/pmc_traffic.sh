# HBM traffic of the conv kernels of one eager bench step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
# --kernel-trace only, as this pool requires) + scripts/pmc_traffic.py -> profiles/r06_conv_hbm_traffic.json
# usage (GPU box): bash scripts/pmc_traffic.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graphs --roofline-reps 0 --no-cpu-baseline --no-dense-check --no-synthesis --no-training --no-fast-mode --no-config4 --no-config5 > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-graphs --roofline-reps 0 --no-cpu-baseline --no-dense-check --no-synthesis --no-training --no-fast-mode --no-config4 --no-config5 > $R/gpurun_out/pmc_w.log 2>&1
cd $R
python3 scripts/pmc_traffic.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) gpurun_out/r06_conv_hbm_traffic.json | tail -60
