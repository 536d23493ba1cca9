#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/${TAG:-r04}
timeout -k 10 500 python tools/soak_parity.py 1.0 c5 > gpurun_out/${TAG:-r04}/soak.txt 2>&1; echo "soak rc $?"; tail -4 gpurun_out/${TAG:-r04}/soak.txt
for c in C3 C4; do timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --detail-out gpurun_out/${TAG:-r04}/full_${c}_detail.json 2>/dev/null > gpurun_out/${TAG:-r04}/full_$c.json; python -c "
import json; d=json.loads(open('gpurun_out/${TAG:-r04}/full_$c.json').read().strip().splitlines()[-1]); print('$c', d['metric'], round(d['value'],1), d['ms_per_step'], d['segments_per_primary_ray'])"; done
timeout -k 10 400 python bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc 2>/dev/null > gpurun_out/${TAG:-r04}/full_C5.json; python -c "
import json; d=json.loads(open('gpurun_out/${TAG:-r04}/full_C5.json').read().strip().splitlines()[-1]); print('C5', d['metric'], round(d['value'],1), d['ms_per_step'], d['segments_per_primary_ray'])"
