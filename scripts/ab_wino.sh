# A/B of the Winograd kernels on one layer shape: round 2's conv_wino (BFM_WINO_V=1) against conv_wino2 (default).
# usage: bash scripts/ab_wino.sh [size cin cout]
set -e
S=${1:-160}; CI=${2:-64}; CO=${3:-64}
for v in 1 2 1 2; do
  BFM_WINO_V=$v timeout -k 10 120 python3 scripts/run_one_conv.py 3 $S $CI $CO 5
done
