#!/bin/bash
# PMC profile of both BVH schedules on C3 (100k triangles) at 4 spp (counters only, separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_c3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/dev_regroup.py 100000 4 > $OUT/p1.log 2>&1 || tail -5 $OUT/p1.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/dev_regroup.py 100000 4 > $OUT/p2.log 2>&1 || tail -5 $OUT/p2.log
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for d in ("p1","p2"):
    for f in glob.glob(R + "/gpurun_out/pmc_c3/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-44:]
            if "trace_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in acc:
            print(k)
            for c, v in sorted(acc[k].items()):
                print("   %-28s %.4g (per launch, %d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
