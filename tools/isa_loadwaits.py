"""Per kernel of an ISA dump (hipcc -S): vector-memory loads against the s_waitcnt vmcnt(...) that wait for them.  A kernel
whose waits mostly come right behind ONE load each (last column) walks memory one round trip at a time -- what an `if` around every
neighbour load compiles to (an exec-masked block per load with s_waitcnt vmcnt(0) behind it).
    python tools/isa_loadwaits.py file.s [...]"""
import re
import sys

for path in sys.argv[1:]:
    name, loads_since, loads, waits, serial = None, 0, 0, 0, 0
    since_load = 0
    rows = []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, loads_since, loads, waits, serial = m.group(1), 0, 0, 0, 0
            continue
        if name is None:
            continue
        t = line.strip()
        since_load += 1
        if re.match(r"(global|buffer|flat|scratch)_load", t):
            loads += 1
            loads_since += 1
            since_load = 0
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            waits += 1
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if loads_since == 1 and n == 0 and since_load <= 8:
                serial += 1
            loads_since = 0
        elif t.startswith("s_endpgm"):
            rows.append((name, loads, waits, serial))
            name = None
    for name, loads, waits, serial in rows:
        if loads >= 4:
            print("%-28s %-70s loads %3d  vmcnt waits %3d  vmcnt(0) right behind a single load: %3d" %
                  (path.split("/")[-1], name[:70], loads, waits, serial))
