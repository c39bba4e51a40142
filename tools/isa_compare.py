#!/usr/bin/env python3
"""Are the kernels of two `hipcc -S --cuda-device-only` outputs the same instruction for instruction?
    python tools/isa_compare.py before.s after.s
Compares, per kernel symbol, the instruction lines between the label and the kernel's .Lfunc_end (comments, directives and
basic-block labels' numbering and the operand order of commutative scalar and/or/xor aside).  Used when source is refactored without an intended change of the code (round 4:
the resolved A/B switches of csrc/photo_train.hip)."""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m and name is None:
            name, body = m.group(1), []
            continue
        if name is not None:
            if ln.startswith(".Lfunc_end"):
                out[name] = body
                name = None
                continue
            t = ln.split(";")[0].strip()
            if not t or t.startswith("."):
                t = re.sub(r"^\.LBB\d+_(\d+):", r"L\1:", t) if t.startswith(".LBB") else ""
                if not t:
                    continue
            t = re.sub(r"\.LBB\d+_(\d+)", r"L\1", t)
            m2 = re.match(r"^(s_(?:and|or|xor)_b(?:32|64)) (\S+), (\S+), (\S+)$", t)
            if m2:                                   # commutative scalar ops: operand order is not a difference
                x, y = sorted([m2.group(3), m2.group(4)])
                t = "%s %s %s, %s" % (m2.group(1), m2.group(2), x, y)
            body.append(t)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    bad = 0
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b:
            print("ONLY IN %s  %s" % ("first" if k in a else "second", k))
            bad += 1
            continue
        same = a[k] == b[k]
        bad += not same
        print("%s %-78s %6d %6d instructions" % ("same" if same else "DIFF", k[:78], len(a[k]), len(b[k])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
