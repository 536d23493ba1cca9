"""Per-kernel hardware counters of an A/B run (on the GPU box): one rocprofv3 --pmc pass per counter group around
tools/ab_kernels.py, then per kernel symbol the average per launch of every counter and a few derived figures.
    tools/pmc_ab.py OUTDIR "CTR CTR ...;CTR CTR ..." -- CONFIG SPP SPEC [SPEC ...] [band]
Never combines --pmc with other trace domains than --kernel-trace (the pool refuses that)."""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, groups = os.path.abspath(sys.argv[1]), [g.split() for g in sys.argv[2].split(";") if g.strip()]
rest = sys.argv[sys.argv.index("--") + 1:]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
acc, calls, dur = {}, {}, {}
for gi, ctrs in enumerate(groups):
    d = os.path.join(out, "g%d" % gi)
    cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                          os.path.join(ROOT, "tools", "ab_kernels.py")] + rest
    p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
    print("pass %d (%s): rc %d" % (gi, " ".join(ctrs), p.returncode), flush=True)
    if p.returncode != 0:
        print((p.stderr or p.stdout)[-600:])
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "rtx::" not in name:
                continue
            k = (name, r["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            calls[k] = calls.get(k, 0) + 1
            if r["Counter_Name"] == ctrs[0]:
                dur.setdefault((name, gi), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
names = sorted({k[0] for k in acc})
for n in names:
    c = {k[1]: acc[k] / calls[k] for k in acc if k[0] == n}
    ds = [sum(v) / len(v) for (nn, gi), v in dur.items() if nn == n]
    ms = sum(ds) / len(ds) if ds else 0.0
    line = "%s\n    avg %.3f ms per launch (under the profiler)" % (n, ms)
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        line += "  lane_util %.3f  valu_busy %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
                                                    c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * ms * 1e-3 * 2.4e9) if ms else 0)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in c:
                line += "  %s/wave_cycles %.3f" % (k[3:].lower(), c[k] / c["SQ_WAVE_CYCLES"])
    print(line)
    print("    " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(c.items())), flush=True)
