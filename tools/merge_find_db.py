#!/usr/bin/env python3
"""Merge the lines MIOpen appended to a recorded user db (gpurun_out/miopen_db_<tag>/) into the shipped one
(digging-into-self-supervised-monocular-depth-estimation_amd/miopen_db/): per file, a line replaces the shipped line with the
same key (the text before '='), new keys are appended.  Only lines the recording itself changed are taken: tools/find_db.sh keeps
the files it started from under <dir>/base/, and a line equal to its base line is left alone (two recordings started from the same
shipped db would otherwise undo each other's results when merged one after the other).  Prints what changed.

    python tools/merge_find_db.py gpurun_out/miopen_db_nhwc_fp32 [more directories]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "digging-into-self-supervised-monocular-depth-estimation_amd", "miopen_db")


def read(path):
    rows = {}
    if os.path.exists(path):
        for line in open(path):
            line = line.rstrip("\n")
            if "=" in line:
                rows[line.split("=", 1)[0]] = line
    return rows


def main():
    for src in sys.argv[1:]:
        for name in sorted(os.listdir(src)):
            if not name.endswith(".txt"):
                continue
            have, new = read(os.path.join(DST, name)), read(os.path.join(src, name))
            base = read(os.path.join(src, "base", name))
            new = {k: v for k, v in new.items() if base.get(k) != v}
            added = [k for k in new if k not in have]
            changed = [k for k in new if k in have and have[k] != new[k]]
            have.update(new)
            with open(os.path.join(DST, name), "w") as f:
                f.write("\n".join(have.values()) + "\n")
            print("%s: +%d new, %d replaced, %d total" % (name, len(added), len(changed), len(have)))


if __name__ == "__main__":
    main()
