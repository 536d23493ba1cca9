#!/usr/bin/env python3
"""Kernel instances of a built library: unbundles the gfx950 code objects of a .so (clang offload bundles in .hip_fatbin) and
lists their kernel descriptors (`*.kd` symbols) with registers, spills, LDS and scratch from the code object's notes.

    tools/kernel_instances.py [rust-raytracing_amd/librtx_hip.so]          # table + count
    tools/kernel_instances.py LIB --count                                   # just the number (tests use kernel_names())
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(lib):
    """The gfx950 code objects (bytes) of every offload bundle in `lib`."""
    data = open(lib, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        i = m.start()
        n_entries = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n_entries):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tlen].decode(errors="replace")
            off += tlen
            if "gfx950" in triple and size:
                out.append(data[i + o:i + o + size])
    return out


def demangle(names):
    if not names:
        return []
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    except OSError:
        out = names
    return [d.split("(")[0].replace("void ", "").replace("rtx::", "") for d in out[:len(names)]]


def kernels(lib):
    """[{name, vgpr, vgpr_spill, sgpr_spill, lds, scratch}] of every kernel in `lib`."""
    rows = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for b in notes.split("- .agpr_count:")[1:]:
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
            rows.append({"symbol": re.search(r"\.name:\s+(\S+)", b).group(1), "vgpr": g("vgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                         "sgpr_spill": g("sgpr_spill_count"), "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size")})
    for r, d in zip(rows, demangle([r["symbol"] for r in rows])):
        r["name"] = d
    return rows


def kernel_names(lib):
    return sorted(r["name"] for r in kernels(lib))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "rust-raytracing_amd", "librtx_hip.so")
    rows = kernels(lib)
    if "--count" in sys.argv:
        print(len(rows))
        sys.exit(0)
    for r in sorted(rows, key=lambda r: r["name"]):
        print("%-64s VGPR %3d  spilled %3d  SGPR spilled %3d  LDS %6d B  scratch %4d B/lane" % (
            r["name"][:64], r["vgpr"], r["vgpr_spill"], r["sgpr_spill"], r["lds"], r["scratch"]))
    print("%d kernel instances, %d bytes (%s)" % (len(rows), os.path.getsize(lib), os.path.relpath(lib, ROOT)))
