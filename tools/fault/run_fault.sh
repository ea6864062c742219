#!/bin/bash
# runs the per-section check on each variant library of the failing tree
cd "$(dirname "$0")"
L=$PWD/compressed-vec_amd/lib
for v in "$@"; do
  echo "=== variant $v"
  CVEC_LIB_PATH=$L/libcvec_$v.so timeout -k 10 150 python3 -u tools/${PROBE:-fault_probe.py} ${CFG:-cfg4} 6 || { echo "variant $v exit $?"; }
done
