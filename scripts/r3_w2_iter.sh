cd $GRAFT_REPO_ROOT
for d in 0 64 16 80; do
  echo "dbg=$d"; BFM_W2_DBG=$d timeout -k 10 120 python3 scripts/run_one_conv.py 3 160 64 64 5 2>&1 | grep ver
done > gpurun_out/r3_w2_ablate.log 2>&1
cat gpurun_out/r3_w2_ablate.log
echo "v1:"; BFM_WINO_V=1 timeout -k 10 120 python3 scripts/run_one_conv.py 3 160 64 64 5 2>&1 | grep ver
