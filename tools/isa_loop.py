"""Instruction mix of the LARGEST loop (longest backward branch span) of one kernel in a hipcc -S listing:
       python tools/isa_loop.py file.s <substring of kernel name>"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
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
    c = collections.Counter(l.strip().split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";")))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    lane = sum(v for k, v in c.items() if k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
    print("%s: largest loop lines %d..%d: %d VALU (%d lane reads/writes, %d DPP, %d v_mov), %d SALU, %d s_nop, %d s_waitcnt, %d SMEM, %d VMEM" % (
        key, a, b, valu, lane, sum(v for k, v in c.items() if k.endswith("_dpp")), c.get("v_mov_b32_e32", 0),
        sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_load", "s_nop", "s_waitcnt"))), c.get("s_nop", 0),
        c.get("s_waitcnt", 0), sum(v for k, v in c.items() if k.startswith("s_load")),
        sum(v for k, v in c.items() if k.startswith(("global_", "buffer_")))))


if __name__ == "__main__":
    main()
