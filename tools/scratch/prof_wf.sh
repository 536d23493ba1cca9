#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wfprof -- python3 $R/tools/scratch/ab_mesh.py > /tmp/wf.log 2>&1
f=$(find /tmp/wfprof -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if 'rtx::' in r['Name']:
        print(r['Name'].split('(')[0][-45:], r['Calls'], 'total %.1f ms' % (float(r['TotalDurationNs'])/1e6), 'avg %.3f ms' % (float(r['AverageNs'])/1e6), 'max %.2f' % (float(r['MaxNs'])/1e6))
PY
f=$(find /tmp/wfprof -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if 'wf_' in r['Kernel_Name']]
# last frame of C3 = launches: find sequences; print durations of the first 24 wf kernels after the first generate of the last C3 frame
gens=[i for i,r in enumerate(rows) if 'generate' in r['Kernel_Name']]
for gi in (gens[2], gens[5]):
    seq=rows[gi:gi+23]
    print(' | '.join('%s %.1f' % (r['Kernel_Name'].split('::')[1][:8], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6) for r in seq))
PY
