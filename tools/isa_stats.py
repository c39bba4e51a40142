"""Static instruction statistics of one kernel in a hipcc -S listing:  python tools/isa_stats.py file.s <substring of kernel name>"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ins = []
    for l in lines[start:end + 1]:
        if l.startswith("\t") and not l.strip().startswith((".", ";")):
            ins.append(l.strip().split()[0])
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print("%s: %d instructions, %d VALU, %d DPP, %d SALU" % (lines[start].split(":")[0], len(ins), valu,
          sum(v for k, v in c.items() if k.endswith("_dpp")), sum(v for k, v in c.items() if k.startswith("s_"))))
    for title, pred in (("scratch", lambda k: "scratch" in k), ("dpp", lambda k: k.endswith("_dpp")),
                        ("memory", lambda k: k.startswith(("global_", "ds_", "buffer_", "s_load", "flat_")))):
        print("  %s: %s" % (title, dict((k, v) for k, v in sorted(c.items()) if pred(k))))
    print("  top: %s" % c.most_common(30))


if __name__ == "__main__":
    main()
