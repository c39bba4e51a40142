#!/usr/bin/env python3
"""Where, in the largest loop of a kernel, do VALU instructions read the result of the VALU instruction issued just before
them (on gfx950 such a pair costs ~2 extra cycles and other waves do not fill the bubble: profiles/r02_micro_valu_dep.txt)?
    python tools/isa_depmap.py file.s <substring of kernel name> [--dump]
Prints the dependent runs (length >= 2 back-to-back dependent instructions) with their opcodes, and totals by run length."""
import collections
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def loop_of(lines, key):
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
    return body[best[1]:best[2] + 1]


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    body = loop_of(open(path).read().split("\n"), key)
    prev_dst, prev_valu = set(), False
    run, runs, n, dep = [], [], 0, 0
    for l in body:
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        t = l.split(";")[0].strip()
        op = t.split()[0]
        if not op.startswith("v_"):
            if op.startswith(("s_waitcnt", "s_nop")) or not op.startswith("s_"):
                pass
            if dump:
                print("      " + t)
            if not op.startswith("s_") or op.startswith(("s_cbranch", "s_branch")):
                prev_valu, prev_dst = False, set()
            continue
        ops = t[len(op):].split(",")
        dst = regs(ops[0]) if not op.startswith("v_cmp") else set()
        srcs = set()
        for x in (ops[1:] if not op.startswith("v_cmp") else ops):
            srcs |= regs(x)
        if op.startswith(("v_fmac", "v_mac")):
            srcs |= dst
        n += 1
        d = prev_valu and bool(srcs & prev_dst)
        dep += d
        if dump:
            print(("DEP   " if d else "      ") + t)
        if d:
            run.append(op)
        else:
            if run:
                runs.append(run)
            run = []
        prev_dst, prev_valu = dst, True
    if run:
        runs.append(run)
    hist = collections.Counter(len(r) for r in runs)
    print("%s: loop of %d VALU instructions, %d (%.1f%%) dependent on their predecessor" % (key, n, dep, 100.0 * dep / max(n, 1)))
    print("dependent runs by length:", dict(sorted(hist.items())))
    ops = collections.Counter(o for r in runs for o in r)
    print("opcodes in dependent position:", ops.most_common(12))


if __name__ == "__main__":
    main()
