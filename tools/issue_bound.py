#!/usr/bin/env python3
"""Issue bound of the training kernel's row loop from its instruction mix (no GPU needed).

The kernel is bound by vector-instruction issue, not by HBM.  How close to THAT bound it runs is the ratio of
    measured  : SQ_INSTS_VALU / (SIMDs x launch cycles)        wave-instructions per SIMD-cycle   (profiles/*_pmc.json)
    bound     : 1 / (mean issue cycles of the loop's VALU instructions)
where the mean weights every VALU instruction of the kernel's main loop (the longest backward-branch span of the ISA hipcc
emits for the shipped flags) by the cycles one wave-instruction of its class occupies a SIMD, measured with
tools/int_rate.hip / tools/valu_rate.hip / tools/pk_rate.hip on MI355X at 3 waves per SIMD (profiles/r03_micro_int_rate.txt,
r04_micro_pk_rate.txt):
    2.8   plain VOP1 / VOP2 / VOPC float and integer ops (v_add_f32, v_mul_f32, v_sub, v_max, v_cmp -> vcc, v_mov ...)
    3.1   v_fma_f32 / v_fmac_f32 (VOP3 float, three operands)
    4.7   DPP and SDWA forms, VOP3-encoded integer / bit-field ops, v_div_scale / v_div_fixup / v_div_fmas, e64 compares and selects
    8.5   transcendentals (v_rcp / v_rsq / v_sqrt / v_exp / v_log), v_readlane / v_readfirstlane / v_writelane, 64-bit integer ops
The static mix stands for the dynamic one: the loop has no data-dependent branches that skip vector work for a whole wave
(exec-masked code still issues).

    python tools/issue_bound.py [--kernel SUBSTRING] [--json profiles/r05_issue_bound.json]
"""
import argparse
import collections
import importlib.util
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "digging-into-self-supervised-monocular-depth-estimation_amd")

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
LANE = ("v_readlane", "v_readfirstlane", "v_writelane")
VOP3_INT = ("v_mad_", "v_add3_", "v_bfe_", "v_bfi_", "v_mul_lo_", "v_mul_hi_", "v_div_scale_", "v_div_fixup_", "v_div_fmas_",
            "v_alignb", "v_perm_", "v_lshl_add_u32", "v_add_lshl_", "v_lshl_or_", "v_and_or_", "v_or3_", "v_xad_", "v_med3_",
            "v_min3_", "v_max3_", "v_cvt_pk", "v_dot", "v_sad_", "v_lerp_", "v_cubeid", "v_ldexp", "v_frexp", "v_trig")


def price(m):
    if m.endswith("_dpp") or m.endswith("_sdwa"):
        return "dpp_sdwa", 4.7
    if m.startswith(TRANS) or m.startswith(LANE) or re.search(r"_[uib]64($|_)", m) or m.startswith("v_mad_u64") or m.startswith("v_mad_i64"):
        return "slow", 8.5
    if m.startswith(("v_fma_f32", "v_fmac_f32", "v_pk_fma")):
        return "fma", 3.1
    if m.startswith(VOP3_INT) or m.endswith("_e64"):
        return "vop3", 4.7
    return "plain", 2.8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="photometric_train_kernelILi2ELb1ELb1E")
    ap.add_argument("--source", default="photo_train.hip")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("_mdx_build", os.path.join(PKG, "build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = [build.HIPCC] + [f for f in build.FLAGS if f != "-fPIC"] + build.EXTRA_FLAGS.get(a.source, []) + \
              ["-S", "--cuda-device-only", os.path.join(build.CSRC, a.source), "-o", asm]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and a.kernel in l)
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
    _, lo, hi = best
    ops = [l.strip().split()[0] for l in body[lo:hi + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    valu = [m for m in ops if m.startswith("v_")]
    classes = collections.Counter()
    cycles = collections.Counter()
    for m in valu:
        c, p = price(m)
        classes[c] += 1
        cycles[c] += p
    total = sum(cycles.values())
    mean = total / max(1, len(valu))
    out = {"kernel": body[0].split(":")[0], "source": a.source, "source_sha16": build.source_sha16(),
           "loop_instructions": len(ops), "loop_valu_instructions": len(valu), "class_counts": dict(classes),
           "class_cycles": {k: round(v, 1) for k, v in cycles.items()},
           "mean_issue_cycles_per_valu_instruction": round(mean, 4),
           "issue_bound_inst_per_simd_cycle": round(1.0 / mean, 4),
           "method": "static VALU mix of the kernel's main loop x measured issue cycles per class (tools/issue_bound.py docstring; "
                     "profiles/r03_micro_int_rate.txt, r04_micro_pk_rate.txt)"}
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
        open(a.json, "a").write("\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
