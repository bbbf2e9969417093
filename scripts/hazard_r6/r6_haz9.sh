#!/bin/bash
set -o pipefail
O=gpurun_out/r6haz; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
BFM_BISECT_SPREAD=0.1 BFM_DIAG_LIB=brainfm_amd/libbrainfm_hip_diag_pk.so timeout -k 10 300 python tests/diag/diag_atlas_bisect.py 60 > $O/atlas_pk.txt 2>&1; grep "wrong voxels in" $O/atlas_pk.txt
BFM_BISECT_SPREAD=0.1 BFM_DIAG_LIB=brainfm_amd/libbrainfm_hip_diag_nopk.so timeout -k 10 300 python tests/diag/diag_atlas_bisect.py 60 > $O/atlas_nopk.txt 2>&1; grep "wrong voxels in" $O/atlas_nopk.txt
