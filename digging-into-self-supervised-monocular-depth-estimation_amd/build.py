"""Builds libmdx_hip.so (the C-ABI of include/mdx.h) for gfx950 with hipcc, in-tree.

    python build.py            # rebuild if sources are newer than the library
    python build.py --force

-ffp-contract=off is part of the numerics contract: fused multiply-adds happen only where the
kernels say __builtin_fmaf (csrc/mdx_device.hpp).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, os.environ.get("MDX_BUILD_NAME", "libmdx_hip.so"))    # MDX_BUILD_NAME + MDX_BUILD_DEFINES: A/B builds
SOURCES = ["photo_fwd.hip", "photo_bwd.hip", "photo_train.hip", "photo_train_finish.hip", "photo_prologue.hip", "photo_abi.hip", "smooth.hip", "loss_total.hip", "adam.hip", "ops.hip", "glue.hip", "glue_nhwc.hip", "disp_head_nhwc.hip", "thinconv_nhwc.hip", "pose_head_nhwc.hip", "norm.hip", "norm_nhwc.hip", "pose.hip", "monitor.hip", "imgproc.hip"]
HEADERS = ["mdx_device.hpp", "mdx_common.hpp", "nhwc_common.hpp", "photo_common.hpp", "photo_train.hpp", "photo_train_math.hpp", "mdx_divtable.inc", os.path.join("..", "..", "include", "mdx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# per-file additions.  photo_train.hip: the SLP vectorizer pairs neighbouring f32 ops into v_pk_* (no faster per element
# on gfx950, profiles/r02_micro_valu_rate.txt) at the price of register pairs, packing moves and un-folded DPP operands
EXTRA_FLAGS = {"photo_train.hip": ["-fno-slp-vectorize"] + os.environ.get("MDX_BUILD_TRAIN_FLAGS", "").split()}    # (A/B builds)
# A/B builds of the kernels: MDX_BUILD_DEFINES="-DMDX_TRAIN_FASTDIV=0 ..." (the shipped library never reads the environment)
FLAGS_EXTRA_ENV = os.environ.get("MDX_BUILD_DEFINES", "").split()
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def source_sha16():
    """Identity of a library build that does not depend on where it was compiled: the sources, headers and flags.  The
    committed counter summaries (profiles/*_kernel_pmc.json) are tied to it; bench.py quotes them only for this build."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS + FLAGS_EXTRA_ENV + [k + ":" + " ".join(v) for k, v in sorted(EXTRA_FLAGS.items())]).encode())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o" if "MDX_BUILD_NAME" not in os.environ else "." + os.environ["MDX_BUILD_NAME"] + ".o"))
        cmd = [HIPCC] + FLAGS + FLAGS_EXTRA_ENV + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
