#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=tests/diag/diag_hazard_r6.py
V=tests/diag/hazard_variants
L=""
for n in base pre all blk loopall rest mul cvt pk mad lshl ld base; do L="$L,$V/$n.hsaco"; done
BFM_DIAG_HSACOS=${L#,} BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 600 python $D 4 eager 250 > $O/variants1.txt 2>&1
grep "^\[" -A1 $O/variants1.txt
