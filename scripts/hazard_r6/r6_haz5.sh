#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=tests/diag/diag_hazard_r6.py
V=tests/diag/hazard_variants
BFM_DIAG_HSACOS=$V/dump.hsaco,$V/base.hsaco BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 600 python $D 4 eager 250 > $O/dump1.txt 2>&1
grep -v "^it " $O/dump1.txt | tail -n 60
