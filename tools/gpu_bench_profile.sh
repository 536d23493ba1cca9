#!/bin/bash
# Runs on the GPU box: bench line, rocprofv3 kernel-trace stats, and the two PMC passes (HBM bytes).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r01
mkdir -p $OUT
cd $R
python bench.py --steps 3 --warmup 1 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
cat $OUT/bench_n1.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
find $OUT -name "*.csv" | head -20
