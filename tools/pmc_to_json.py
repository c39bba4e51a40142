#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/pmc_bench.sh (bench.py's own launches) or tools/pmc_train.sh (tools/kbench.py)
->  profiles/<tag>_..._kernel_pmc.json

    python tools/pmc_to_json.py gpurun_out/pmc_bench profiles/r03_bench_kernel_pmc.json B H W S NSCALES

Per kernel: mean counters per launch; HBM-side traffic as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE
from separate passes, KiB; FETCH_SIZE x2 for gfx950's 64-B tally of 128-B requests -- calibrated there for wide
coalesced reads; these kernels read 4-8 bytes per lane, so the raw figure is given too); the issue-side numbers
bench.py prints next to the HBM fraction.  The file records the sha256 of the libmdx_hip.so it was measured on and the
workload shape: bench.py only quotes it for that very build and shape."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

SIMDS, CLK_GHZ = 1024, 2.4


def main():
    src, dst = sys.argv[1], sys.argv[2]
    shape = [int(v) for v in sys.argv[3:8]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "digging-into-self-supervised-monocular-depth-estimation_amd", "libmdx_hip.so")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if name.startswith("mdx::"):
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    sys.path.insert(0, os.path.dirname(so))
    import build as B
    out = {"lib_sha16": hashlib.sha256(open(so, "rb").read()).hexdigest()[:16], "source_sha16": B.source_sha16(), "shape": shape,
           "source": ("tools/pmc_bench.sh: rocprofv3 --pmc over `python bench.py` itself (the timed step's own launches and "
                      "tensors), means per launch" if ("pmc_bench" in src or "pmc_c" in src) else
                      "tools/pmc_train.sh (tools/kbench.py --what train), means per launch"), "kernels": {}}
    regs = {}      # registers / occupancy: tools/kernel_resources.py (the trace's VGPR_Count field is not the ISA's count)
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"counters": m}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["fetch_kib"], e["write_kib"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            e["traffic_bytes"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
            e["traffic_bytes_raw"] = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        if "SQ_INSTS_VALU" in m:
            e["valu_wave_insts"] = m["SQ_INSTS_VALU"]
        if "SQ_ACTIVE_INST_VALU" in m:
            # SQ_ACTIVE_INST_VALU counts, in units of 4 cycles, the time waves spend with a VALU instruction executing: summed
            # over the waves of a SIMD it is the time that SIMD's vector ALU is occupied (a wave64 instruction holds it 4 cycles)
            e["valu_busy_us"] = m["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (CLK_GHZ * 1e3)
        if "SQ_WAVE_CYCLES" in m:
            for key, c in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
                if c in m:
                    e[key] = m[c] / m["SQ_WAVE_CYCLES"]
        if k in dur:
            e["kernel_us_profiled"] = sum(dur[k]) / len(dur[k])
        if k in regs:
            e.update(regs[k])
            tot = regs[k]["vgprs"] + regs[k]["accum_vgprs"]
            e["waves_per_simd"] = min(8, 512 // max(8, (tot + 7) // 8 * 8))
        out["kernels"][k] = e
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
