#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=tests/diag/diag_hazard_r6.py
V=tests/diag/hazard_variants
BFM_DIAG_HSACOS=$V/nopk.hsaco,$V/base.hsaco,$V/nopk.hsaco BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 600 python $D 4 eager 400 > $O/nopk1.txt 2>&1
grep "^\[" -A1 $O/nopk1.txt | cut -c1-260
timeout -k 10 300 ./scripts/micro/pk_f32_mfma_hazard 200 > $O/pkmicro1.txt 2>&1; cat $O/pkmicro1.txt
