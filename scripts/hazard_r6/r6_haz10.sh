#!/bin/bash
# round 6: which side of the pair a library-free reproducer is missing -- the REAL victim (grid_pull3d, packed build, from its
# code object) beside the STAND-ALONE aggressor (tests/diag/hazard_aggressor.hip), mode bits 1 loads 2 LDS reads 4 MFMAs
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
D=tests/diag/diag_hazard_r6.py
V=tests/diag/hazard_variants
for m in 7 5 6 4; do
  BFM_DIAG_AGGR=$m BFM_DIAG_HSACOS=$V/base.hsaco BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 300 python $D 4 eager 300 > $O/aggr_$m.txt 2>&1
  echo "aggressor mode $m:"; grep "^\[" -A1 $O/aggr_$m.txt | cut -c1-230
done
BFM_DIAG_HSACOS=$V/base.hsaco BFM_DIAG_TILE=16 BFM_DIAG_CUMASK=same timeout -k 10 300 python $D 4 eager 300 > $O/aggr_lib.txt 2>&1
echo "library conv_wino4d:"; grep "^\[" -A1 $O/aggr_lib.txt | cut -c1-230
