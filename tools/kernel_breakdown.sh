#!/bin/bash
# rocprofv3 --kernel-trace --stats of one config under one kernel id: per-kernel totals (the wavefront form is ~23 kernels per launch).
#   tools/kernel_breakdown.sh TAG CONFIG SPP KERNEL [band]      -> gpurun_out/${TAG:-r03}/TAG/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG:-r03}/$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/ab_kernels.py $2 $3 $4 $4 $5 > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s total %10.3f ms avg %9.3f ms  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
grep kernel $OUT/log.txt
