#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a `hipcc -S` listing:
    python tools/isa_loops.py ab/riccati.s <mangled kernel name prefix> [--dump LABEL]
Lists every backward branch (label, line range, instruction counts by class); --dump prints the opcode histogram of one loop."""
import collections
import re
import sys


def kernel_lines(path, prefix):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(re.escape(prefix) + r"[^\s]*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
    k = kernel_lines(path, prefix)
    labels = {}
    for i, l in enumerate(k):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    for i, l in enumerate(k):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = k[labels[m.group(1)]:i]
            ins = [x.strip().split()[0] for x in body if x.strip() and not x.strip().startswith((".", ";")) and not x.strip().endswith(":")]
            c = collections.Counter(classify(x) for x in ins)
            print(m.group(1), "lines", labels[m.group(1)], i, "n", len(ins), dict(c))
            if dump == m.group(1):
                h = collections.Counter(ins)
                for op, cnt in h.most_common():
                    print(f"    {cnt:5d} {op}")


if __name__ == "__main__":
    main()
