#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or seeded or triangle or fuzz or edge or c3_full or c5_band or axis or joint" 2>&1 | tail -8
RTX_HIP_BVH_CLASSIC=1 timeout -k 10 200 python tools/scratch/ab_mesh.py
timeout -k 10 200 python tools/scratch/ab_mesh.py
