#!/usr/bin/env python3
"""Copies the newest outputs of tools/gpu_bench_profile.sh from gpurun_out/r01 into profiles/ (tracked)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r01")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):
    files = glob.glob(os.path.join(G, pattern))
    return max(files, key=os.path.getmtime)


stats = newest("trace/*/*_kernel_stats.csv")
fetch = newest("pmc_fetch/*/*_counter_collection.csv")
write = newest("pmc_write/*/*_counter_collection.csv")
shutil.copy(stats, os.path.join(P, "%s_c2_64spp_kernel_stats.csv" % tag))
shutil.copy(os.path.join(G, "bench_n1.json"), os.path.join(P, "%s_bench_n1.json" % tag))
res = {}
with open(os.path.join(P, "%s_c2_64spp_pmc_hbm.csv" % tag), "w") as o:
    first = True
    for f, cname in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        for i, l in enumerate(open(f)):
            if i == 0:
                if first:
                    o.write(l); first = False
            elif "rtx::" in l:
                o.write(l)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rtx::trace" in k and r["Counter_Name"] == cname:
                res.setdefault("bvh" if "bvh" in k else "mixed", {}).setdefault(cname, []).append(float(r["Counter_Value"]))
tj = {}
for name, key in (("bvh", "c2_64spp_kernel4"), ("mixed", "c2_64spp_kernel2")):
    if name not in res:
        continue
    f = sum(res[name]["FETCH_SIZE"]) / len(res[name]["FETCH_SIZE"])
    w = sum(res[name]["WRITE_SIZE"]) / len(res[name]["WRITE_SIZE"])
    tj[key] = {"kernel": "rtx::trace_%s_kernel" % name, "round": tag, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
               "hbm_bytes_per_launch": (f + w) * 1024,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averaged over the launches of one bench.py "
                       "run; one launch = one C2 frame (1920x1080x64spp). FETCH_SIZE is quoted uncorrected: MI355X_MICROARCH.md's x2 "
                       "correction is calibrated for 16 B/lane streaming loads only; these kernels' memory-side reads are 8-32 B/lane "
                       "gathers and SoA slot loads.",
               "source": "profiles/%s_c2_64spp_pmc_hbm.csv" % tag}
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
# the bench line of this run read the PREVIOUS traffic.json; the PMC passes of this very run (same box, same command)
# are the figure that belongs to it
bj = json.load(open(os.path.join(P, "%s_bench_n1.json" % tag)))
if "c2_64spp_kernel4" in tj:
    bj["roofline"]["traffic"] = tj["c2_64spp_kernel4"]["hbm_bytes_per_launch"]
if "c2_64spp_kernel2" in tj and "lds_sweep" in bj:
    bj["lds_sweep"]["roofline"]["traffic"] = tj["c2_64spp_kernel2"]["hbm_bytes_per_launch"]
json.dump(bj, open(os.path.join(P, "%s_bench_n1.json" % tag), "w"))
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in tj.items()}))
for l in open(stats).read().splitlines()[:4]:
    print(l[:230])
b = json.load(open(os.path.join(G, "bench_n1.json")))
print("value", b["value"], "lds", b["lds_sweep"]["value"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["faithful_Mrays_s"],
      "bench avg ms", b["roofline"]["avg_launch_ms"], b["lds_sweep"]["roofline"]["avg_launch_ms"])
