#!/usr/bin/env python3
"""Registers, spills, LDS and scratch of every kernel in the ISA files hipcc --save-temps leaves (amdhsa.kernels metadata):
    tools/kernel_table.py DIR_WITH_.s_FILES      (build them with: hipcc --offload-arch=gfx950 ... --save-temps -c csrc/X.hip)"""
import glob, os, re, subprocess, sys
d = sys.argv[1] if len(sys.argv) > 1 else "."
filt = "c++filt"
for path in sorted(glob.glob(os.path.join(d, "rtx_*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    s = open(path).read()
    for b in s.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        try:
            dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        dem = dem.split("(")[0].replace("void ", "").replace("rtx::", "")
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        print("%-52s VGPR %3d  spilled %3d  SGPR spilled %3d  LDS %6d B  scratch %4d B/lane" % (
            dem[:52], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
