#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "--- 96-byte nodes"; RTX_HIP_NO_QNODES=1 python tools/scratch/ab_mesh.py
echo "--- 64-byte nodes"; python tools/scratch/ab_mesh.py
