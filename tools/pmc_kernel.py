"""SQ + HBM counters of single kernels of a bench leg (bench.py collects them per launch; this lists them per kernel symbol).
    tools/pmc_kernel.py LEG SYMBOL [SYMBOL...]     LEG = CONFIG:SPP:full|band:KERNEL, e.g.  C3:8:full:6 wf_trace_packet_kernel wf_shade_kernel
Values are per kernel launch (the sum over the symbol's rows divided by their number)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
leg = sys.argv[1]
for sym in sys.argv[2:]:
    log = []
    c, src = bench.pmc_collect(leg, (sym,), log)
    if not c:
        print(sym, "FAILED", src, log); continue
    t = c.get("_pmc_launch_s", 0)
    lane = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    print("%s: avg launch %.3f ms  lane_util %.3f  valu_busy %.3f  insts_valu/launch %.3g  wait_any/wave_cycles %.3f  active_any/wave_cycles %.3f  fetch2x %.2f GB write %.2f GB (per launch)" % (
        sym, t * 1e3, lane, c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * t * 2.4e9) if t else 0, c["SQ_INSTS_VALU"],
        c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 2 * c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9), flush=True)
    print("   ", {k: ("%.4g" % v) for k, v in c.items()}, flush=True)
