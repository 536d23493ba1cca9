#!/usr/bin/env python3
"""Where does a kernel touch its spilled registers?  Reads hipcc --save-temps ISA (.s), takes one kernel (substring of its
mangled name), finds the loops from backward branches and lists, per loop, the scratch loads/stores (VGPR spills) and
v_writelane/v_readlane (SGPR spills) inside it -- so that "the spills are outside the traversal loop" is a statement
one can check rather than hope.

    tools/isa_spills.py file.s KERNEL_SUBSTRING [--dump]
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(?:\S+,\s*)?(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i))
    # merge identical heads: one loop per head, widest extent
    by_head = {}
    for a, b in loops:
        by_head[a] = max(by_head.get(a, a), b)
    loops = sorted(by_head.items())

    def is_instr(l):
        s = l.strip()
        return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")

    def count(a, b, pat):
        return sum(1 for l in body[a:b + 1] if re.search(pat, l))

    n_instr = sum(1 for l in body if is_instr(l))
    print("kernel %s: %d ISA lines, %d instructions, %d loops" % (key, len(body), n_instr, len(loops)))
    print("whole kernel: scratch_load %d scratch_store %d v_writelane %d v_readlane %d" % (
        count(0, len(body) - 1, r"\bscratch_load"), count(0, len(body) - 1, r"\bscratch_store"),
        count(0, len(body) - 1, r"\bv_writelane"), count(0, len(body) - 1, r"\bv_readlane")))
    print("%-8s %-8s %6s %5s %5s %5s %5s %5s %5s %5s %s" % ("head", "tail", "instr", "depth", "sld", "sst", "wrl", "rdl", "gld", "f64", "ds"))
    for a, b in loops:
        depth = sum(1 for (c, d) in loops if c <= a and d >= b) - 1
        ni = sum(1 for l in body[a:b + 1] if is_instr(l))
        print("%-8d %-8d %6d %5d %5d %5d %5d %5d %5d %5d %d" % (
            a, b, ni, depth, count(a, b, r"\bscratch_load"), count(a, b, r"\bscratch_store"),
            count(a, b, r"\bv_writelane"), count(a, b, r"\bv_readlane"), count(a, b, r"\bglobal_load"),
            count(a, b, r"\bv_\w+_f64"), count(a, b, r"\bds_")))
    if "--dump" in sys.argv:
        for i, l in enumerate(body):
            print("%6d %s" % (i, l))


if __name__ == "__main__":
    main()
