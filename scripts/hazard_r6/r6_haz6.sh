#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=tests/diag/diag_hazard_r6.py
BFM_DIAG_LIB=brainfm_amd/libbrainfm_hip_ablate.so BFM_DIAG_ABLS=0,1,2,4,8,16,64,128,256,old,12,28,0 BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 600 python $D 4 eager 200 > $O/ablate1.txt 2>&1
grep "^\[" -A1 $O/ablate1.txt | cut -c1-260
