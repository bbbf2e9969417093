# HBM traffic of hot path B: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, the program
# directly after --) over scripts/synth_cold_once.py + scripts/pmc_synth.py -> gpurun_out/r05_synth_hbm_traffic.json
# usage (GPU box): bash scripts/pmc_synth.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_f /tmp/pmcs_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcs_f -o f --output-format csv -- python3 $R/scripts/synth_cold_once.py $R/gpurun_out/synth_sections_f.json > $R/gpurun_out/pmcs_f.log 2>&1 || { tail -5 $R/gpurun_out/pmcs_f.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcs_w -o w --output-format csv -- python3 $R/scripts/synth_cold_once.py $R/gpurun_out/synth_sections_w.json > $R/gpurun_out/pmcs_w.log 2>&1 || { tail -5 $R/gpurun_out/pmcs_w.log; exit 1; }
cd $R
python3 scripts/pmc_synth.py $(find /tmp/pmcs_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmcs_w -name "*counter_collection.csv" | head -1) gpurun_out/synth_sections_f.json gpurun_out/r05_synth_hbm_traffic.json
