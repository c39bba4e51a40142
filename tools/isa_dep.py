"""How many VALU instructions of a kernel read a VGPR written by the instruction issued just before them (on gfx950
such an instruction costs ~4 cycles instead of ~2 and other waves do not fill the bubble: profiles/r02_micro_valu_dep.txt).
    python tools/isa_dep.py file.s <substring of kernel name> [first_line last_line]"""
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


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    if len(sys.argv) > 4:
        start, end = start + int(sys.argv[3]), start + int(sys.argv[4])
    prev_dst, prev_was_valu = set(), False
    n = dep = dpp = dep_dpp = trans = 0
    for l in lines[start:end + 1]:
        if not l.startswith("\t") or l.strip().startswith((".", ";")):
            continue
        body = l.split(";")[0].strip()
        op = body.split()[0]
        if not op.startswith("v_"):
            prev_was_valu = False
            prev_dst = set()
            continue
        ops = body[len(op):].split(",")
        nd = 2 if op.startswith(("v_div_scale", "v_cmp")) is False and False else 1
        dst = regs(ops[0]) if not op.startswith("v_cmp") else set()
        srcs = set()
        for t in ops[1:] if not op.startswith("v_cmp") else ops:
            srcs |= regs(t)
        if op.startswith(("v_fmac", "v_mac")) or "_dpp" in op and False:
            srcs |= dst
        if op.startswith(("v_fmac", "v_mac")):
            srcs |= dst
        n += 1
        is_dpp = op.endswith("_dpp")
        dpp += is_dpp
        trans += op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos"))
        if prev_was_valu and (srcs & prev_dst):
            dep += 1
            dep_dpp += is_dpp
        prev_dst, prev_was_valu = dst, True
    print("%s: %d VALU, %d (%.1f%%) read the previous instruction's result; %d DPP (%d of them dependent); %d transcendental"
          % (key, n, dep, 100.0 * dep / max(n, 1), dpp, dep_dpp, trans))


if __name__ == "__main__":
    main()
