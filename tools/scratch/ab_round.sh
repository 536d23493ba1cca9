#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or seeded or identical or edge or fuzz or c2_full or c4_shaped or overflow or spheres_kernel" 2>&1 | tail -3
for R in 0 8 12 16 24 32 48; do
  RTX_HIP_BVH_ROUND=$R python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-lds-sweep --no-pmc 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); a=d['roofline']['algorithmic']
print('round $R', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms', 'box/seg', round(a['box_tests_per_segment'],2), 'exact/seg', round(a['exact_tests_per_segment'],3), 'mean', d['image_mean'])"
done
RTX_HIP_BVH_ROUND=16 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or seeded or identical or edge or fuzz or c2_full or c4_shaped or overflow or spheres_kernel" 2>&1 | tail -3
