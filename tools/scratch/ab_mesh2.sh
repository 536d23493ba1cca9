#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for W in 4 5 6; do
  if [ $W = 4 ]; then unset RTX_HIP_LIB; else export RTX_HIP_LIB=$PWD/tools/scratch/librtx_w$W.so; fi
  echo "waves/SIMD $W"; timeout -k 10 200 python tools/scratch/ab_mesh.py
done
