#!/bin/bash
# A/B of the MIXED kernel's launch shapes on C2 at 16 spp (one process per variant, same box).
for v in ${VARIANTS:-0 1 2 3 4 5}; do
  echo -n "variant $v: "
  RTX_HIP_MIXED_VARIANT=$v timeout -k 10 120 python bench.py --steps 3 --warmup 1 --spp ${SPP:-16} --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.1f Mrays/s  %.2f ms/step  valu_frac %.3f  mean %.9f' % (d['value'], d['ms_per_step'], d['roofline']['valu_frac'], d['image_mean']))
    elif l.strip(): print(l.rstrip())
"
done
