#!/usr/bin/env python3
"""Timeline of the kernels of ONE render (start / end in ms since the first kernel of the last render): which launches overlap.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/halves_timeline.py render [band] [TUNING]
    python3 tools/halves_timeline.py show gpurun_out/tl
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "render":
    import torch
    import bench
    import rust_raytracing_amd as rtx
    from rust_raytracing_amd import scenes, tiles
    cfg = bench.CONFIGS["C2"]
    objs = bench.make_objects(cfg)
    world = 8 if "band" in sys.argv else 1
    tune = int(sys.argv[-1], 0) if sys.argv[-1].startswith("0x") else 0
    part = tiles.Partition(cfg["h"], 0, world, tiles.ROW_BLOCK)
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=64, seed=scenes.RENDER_SEED, kernel=rtx.RTX_KERNEL_AUTO, tuning=tune),
                                rtx.Camera(*scenes.CAMERA), objs).upload(0)
    band = part.alloc_band(cfg["w"], torch.device("cuda", 0))
    for _ in range(3):
        part.render(hnd, cfg["w"], band, want_stats=False)
        torch.cuda.synchronize()
    hnd.close()
else:
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "trace_" in r["Kernel_Name"] or "resolve" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = max(i for i, r in enumerate(rows) if "resolve" in r["Kernel_Name"])
    first = max([i for i, r in enumerate(rows[:last]) if "resolve" in r["Kernel_Name"]] + [-1]) + 1
    t0 = int(rows[first]["Start_Timestamp"])
    for r in rows[first:last + 1]:
        print("%8.3f .. %8.3f ms  queue %-4s %s" % ((int(r["Start_Timestamp"]) - t0) * 1e-6, (int(r["End_Timestamp"]) - t0) * 1e-6,
                                                  r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
