#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_synth.py -m gpu -x -q -k "beside or captured_graphs or l1_reuse" > $O/pytest_haz.txt 2>&1; tail -n 5 $O/pytest_haz.txt
bash scripts/bench_quick.sh $O/bench_nopk.json
