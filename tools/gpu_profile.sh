#!/bin/bash
# Runs on the GPU box (gpurun): the bench line (with its live rocprofv3 counter passes), then the rocprofv3 kernel-trace
# summary of the same command.  Outputs under gpurun_out/$TAG/; tools/store_profiles.py copies the judged files to profiles/.
#   tools/gpu_profile.sh r02
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "[gpu_profile] bench.py (N=1, default flags)"
python bench.py --steps 3 --warmup 1 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -30 $OUT/bench_n1.err; exit 1; }
python - <<PY
import json
d = json.load(open("$OUT/bench_n1.json"))
r = d["roofline"]
print("value %.1f Mrays/s  %.2f ms/step  frac %.3f (%s)  lane_util %s  hbm_frac %s  cpu %.4f Mrays/s x%.0f" % (
    d["value"], d["ms_per_step"], r["frac"], r["frac_source"], r["lane_utilisation"], r["hbm_frac"],
    d["cpu_baseline"]["value"], d["speedup_vs_cpu"]["primary_rays"]))
for o in d.get("other_configs", []):
    print("  %s %.1f Mrays/s  %.1f Mseg/s  frac %.3f  hbm_frac %s" % (o["config"], o["value"], o["Msegments_per_s"], o["roofline"]["frac"], o["roofline"]["hbm_frac"]))
if "lds_sweep" in d:
    print("  lds_sweep %.1f Mrays/s frac %.3f" % (d["lds_sweep"]["value"], d["lds_sweep"]["roofline"]["frac"]))
print("log:", d.get("log"))
PY
export TMPDIR=/tmp
cd /tmp
echo "[gpu_profile] rocprofv3 --kernel-trace --stats of the same command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200
