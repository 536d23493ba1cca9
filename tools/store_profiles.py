#!/usr/bin/env python3
"""Copies what tools/gpu_profile.sh left under gpurun_out/<tag>/ into profiles/ (tracked): the bench line, the rocprofv3
kernel-trace summary of the same command, and profiles/pmc_counters.json = the per-launch counters of that bench run
stamped with the hash of the kernel sources they were measured on (bench.py falls back to it when rocprofv3 is absent).

    python tools/store_profiles.py r02
"""
import glob
import importlib.util
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")

spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

line = json.load(open(os.path.join(G, "bench_detail_n1.json")))                 # the full record; bench_n1.json = the compact stdout line
shutil.copy(os.path.join(G, "bench_n1.json"), os.path.join(P, "%s_bench_n1.json" % tag))
shutil.copy(os.path.join(G, "bench_detail_n1.json"), os.path.join(P, "%s_bench_detail_n1.json" % tag))
stats = max(glob.glob(os.path.join(G, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
shutil.copy(stats, os.path.join(P, "%s_bench_kernel_stats.csv" % tag))

for name in ("C3", "C5"):                      # the kernel-trace summaries of the C3 and C5-band legs
    fs = glob.glob(os.path.join(G, "trace_%s" % name, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(max(fs, key=os.path.getmtime), os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, name.lower() + ("_band" if name == "C5" else ""))))

legs, legs_pk = {}, {}


def per_kernel(roof):
    """roofline['kernels'] back into {symbol: {counter-like figures}} is not needed: bench stores the raw per-kernel counters"""
    return roof.get("counters_per_kernel")


spp = line["config"]["rays_per_pixel"]
if line["roofline"].get("counters"):
    legs["C2:%d:full:0" % spp] = line["roofline"]["counters"]
    legs_pk["C2:%d:full:0" % spp] = per_kernel(line["roofline"])
if line.get("lds_sweep", {}).get("roofline", {}).get("counters"):
    legs["C2:%d:full:2" % spp] = line["lds_sweep"]["roofline"]["counters"]
    legs_pk["C2:%d:full:2" % spp] = per_kernel(line["lds_sweep"]["roofline"])
for o in line.get("other_configs", []):
    if o["roofline"].get("counters"):
        s = o["rays_per_pixel"]
        key = "%s:%d:%s:0" % (o["config"], s, "band" if "band" in o["workload"] else "full")
        legs[key] = o["roofline"]["counters"]
        legs_pk[key] = per_kernel(o["roofline"])
json.dump({"kernel_source_hash": bench.kernel_source_hash(), "collected": "%s, %s" % (tag, time.strftime("%Y-%m-%d")),
           "note": "per launch, as rocprofv3 reported them (FETCH_SIZE / WRITE_SIZE in KB); legs_per_kernel: the same per kernel "
                   "symbol of the launch (+ _ms: its average duration under the profiler, _calls: rows per launch); passes: " +
                   "; ".join(" ".join(c) for _, c in bench.PMC_PASSES), "legs": legs, "legs_per_kernel": legs_pk},
          open(os.path.join(P, "pmc_counters.json"), "w"), indent=1)
print("stored", sorted(legs), "kernel sources", bench.kernel_source_hash())
for l in open(stats).read().splitlines()[:6]:
    print(l[:220])
