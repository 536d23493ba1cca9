#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/${TAG:-r03}
timeout -k 10 500 python tools/soak_parity.py 1.0 c5 > gpurun_out/${TAG:-r03}/soak.txt 2>&1; echo "soak rc $?"; tail -4 gpurun_out/${TAG:-r03}/soak.txt
for c in C3 C4; do timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null > gpurun_out/${TAG:-r03}/full_$c.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG:-r03}/full_$c.json')); print('$c', d['metric'], round(d['value'],1), d['ms_per_step'], d['segments_per_primary_ray'])"; done
timeout -k 10 400 python bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc 2>/dev/null > gpurun_out/${TAG:-r03}/full_C5.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG:-r03}/full_C5.json')); print('C5', d['metric'], round(d['value'],1), d['ms_per_step'], d['segments_per_primary_ray'])"
