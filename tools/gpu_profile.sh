#!/bin/bash
# Runs on the GPU box (gpurun): the bench line (with its live rocprofv3 counter passes), then the rocprofv3 kernel-trace
# summary of the same command.  Outputs under gpurun_out/$TAG/; tools/store_profiles.py copies the judged files to profiles/.
#   tools/gpu_profile.sh r04
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
echo "[gpu_profile] bench.py (N=1, default flags)"
python bench.py --steps 3 --warmup 1 --detail-out $OUT/bench_detail_n1.json > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -30 $OUT/bench_n1.err; exit 1; }
python - <<PY
import json
line = open("$OUT/bench_n1.json").read().strip().splitlines()[-1]
print("compact line: %d characters" % len(line)); json.loads(line)
d = json.load(open("$OUT/bench_detail_n1.json"))
r = d["roofline"]
iss = (r.get("issued") or {}).get("frac")
print("value %.1f Mrays/s  %.2f ms/step  useful frac %.3f  issued frac %s  lane_util %s  valu_busy %s  hbm_frac %s  cpu %.4f Mrays/s (%s threads' worth) x%.0f" % (
    d["value"], d["ms_per_step"], r["frac"], iss, r["lane_utilisation"], r["valu_busy"], r["hbm_frac"],
    d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["speedup_vs_cpu"]["primary_rays"]))
for s_ in r.get("stages", []):
    print("    %s: %.2f ms  useful %.3f issued %s lanes %s busy %s salu %s" % (s_["stage"], s_["ms"], s_["frac"], s_.get("issued_frac"), s_.get("lane_utilisation"), s_.get("valu_busy"), s_.get("salu_busy")))
for o in d.get("other_configs", []):
    print("  %s %.1f Mrays/s  %.1f Mseg/s  useful %.3f issued %s hbm_frac %s %s" % (o["config"], o["value"], o["Msegments_per_s"], o["roofline"]["frac"], (o["roofline"].get("issued") or {}).get("frac"), o["roofline"]["hbm_frac"], "band/full %.3f" % o["band_rate_over_full_frame_rate"] if "band_rate_over_full_frame_rate" in o else ("band" if "band" in o["workload"] else "full")))
    for k in o["roofline"].get("kernels", []):
        print("      %-45s %.2f ms issued %s lanes %s busy %s" % (k["kernel"][:45], k["avg_ms_per_launch"], k.get("issued_frac"), k.get("lane_utilisation"), k.get("valu_busy")))
if "lds_sweep" in d:
    print("  lds_sweep %.1f Mrays/s useful %.3f issued %s" % (d["lds_sweep"]["value"], d["lds_sweep"]["roofline"]["frac"], (d["lds_sweep"]["roofline"].get("issued") or {}).get("frac")))
print("log:", d.get("log"))
PY
export TMPDIR=/tmp
cd /tmp
echo "[gpu_profile] rocprofv3 --kernel-trace --stats of the same command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-other-configs > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -3
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -8 "$f" | cut -c1-200

for leg in "C3 64" "C5 4 band"; do
  set -- $leg
  echo "[gpu_profile] rocprofv3 --kernel-trace --stats of $leg"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1 -- python3 $R/tools/ab_kernels.py $1 $2 0 $3 > $OUT/trace_$1.log 2>&1 || { tail -20 $OUT/trace_$1.log; exit 1; }
  f=$(find $OUT/trace_$1 -name "*kernel_stats.csv" | head -1)
  head -6 "$f" | cut -c1-160
done
