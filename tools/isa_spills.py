#!/usr/bin/env python3
"""Scratch (spill) instructions of one kernel in a hipcc -S listing, marked IN / OUT of its largest loop:
    python tools/isa_spills.py file.s <substring of kernel name>"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    label = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            label[m.group(1)] = i
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i and i - label[m.group(1)] > best[0]:
            best = (i - label[m.group(1)], label[m.group(1)], i)
    _, a, b = best
    n_in = 0
    for i, l in enumerate(body):
        if "scratch_" in l:
            inside = a <= i <= b
            n_in += inside
            print("%s %5d (loop %d..%d) %s" % ("IN " if inside else "OUT", i, a, b, l.strip()[:90]))
    print("%d scratch instructions inside the loop" % n_in)


if __name__ == "__main__":
    main()
