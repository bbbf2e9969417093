cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 7 8 16 24 31; do
  echo "dbg=$d"; BFM_W2_DBG=$d timeout -k 10 120 python3 scripts/run_one_conv.py 3 160 64 64 5 2>&1 | grep ver
done > gpurun_out/r3_w2_ablate.log 2>&1
cat gpurun_out/r3_w2_ablate.log
bash scripts/pmc_wino.sh gpurun_out/pmc_w2 > gpurun_out/r3_w2_pmc.log 2>&1
cat gpurun_out/pmc_w2/summary.txt
