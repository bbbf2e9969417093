set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_infer.py -x -q -k "winograd or moment_rows or masked_last or uniform_background or mfma_network or full_width_blocks" > gpurun_out/r3_w2_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3_w2_tests.log
tail -5 gpurun_out/r3_w2_tests.log
bash scripts/ab_wino.sh 160 64 64 > gpurun_out/r3_w2_ab.log 2>&1; echo "ab rc=$?" >> gpurun_out/r3_w2_ab.log
bash scripts/ab_wino.sh 160 32 64 >> gpurun_out/r3_w2_ab.log 2>&1
bash scripts/ab_wino.sh 80 128 128 >> gpurun_out/r3_w2_ab.log 2>&1
bash scripts/ab_wino.sh 80 64 128 >> gpurun_out/r3_w2_ab.log 2>&1
cat gpurun_out/r3_w2_ab.log
