#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or seeded or identical or edge or fuzz or c2_full or c4_shaped or overflow" 2>&1 | tail -3
for w in classic 4 5 6 8; do
  if [ $w = classic ]; then export RTX_HIP_BVH_CLASSIC=1; else unset RTX_HIP_BVH_CLASSIC; export RTX_HIP_BVH_WPS=$w; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-lds-sweep --no-pmc 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); a=d['roofline']['algorithmic']
print('$w', round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms', 'box/seg', round(a['box_tests_per_segment'],2), 'exact/seg', round(a['exact_tests_per_segment'],3), 'mean', d['image_mean'])"
done
