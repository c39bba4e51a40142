"""Library tuning data shipped with the path (MIOpen find-db for the ResNet convolutions on gfx950)."""
import os

PACKAGE_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install_miopen_db(rank=None):
    """Point MIOpen at the convolution find-db tuned for this path's network shapes on gfx950.

    `miopen_db/` holds MIOpen's own user find-db / perf-db text files, recorded once on an MI355X by running the
    flagship step with find mode on (bench.py --miopen-find, ~14 min): for every convolution of the ResNet-18
    depth / pose networks at 192x640, batch 12, fp32 it names the fastest solver.  MIOpen consults it in immediate
    mode too, so the networks pick the tuned kernels from the first step without any search.  Each rank gets a
    private writable copy (MIOpen locks and appends to the files).  An explicit MIOPEN_USER_DB_PATH wins.
    Must run before the first convolution of the process.  Returns the directory in use (None if disabled).
    """
    import shutil
    import tempfile
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = os.path.join(PACKAGE_DIR, "miopen_db")
    if os.environ.get("MDX_MIOPEN_DB", "1") == "0" or not os.path.isdir(src):
        return None
    if rank is None:
        rank = int(os.environ.get("RANK", "0"))
    names = [n for n in os.listdir(src) if n.endswith(".txt")]
    # MIOpen names its user db files after its own build ("gfx950100.HIP.<version>-<hash>.ufdb.txt") and ignores every
    # other name: tell the caller when the running MIOpen cannot use what is shipped instead of reporting a db in use
    tag = miopen_db_tag()
    if tag is not None and not any(tag in n for n in names):
        import warnings
        warnings.warn("mdx.tuning: the shipped MIOpen find-db (%s) was recorded with another MIOpen build than the one "
                      "running (%s); MIOpen will ignore it -- convolutions take the heuristic solvers" %
                      (sorted(set(n.split(".HIP.")[-1].split(".u")[0] for n in names)), tag))
        return None
    # a directory of this process's own (pid in the name, removed at exit): two jobs of one user with the same RANK --
    # several single-GPU runs, pytest beside bench.py -- must not wipe or rewrite files the other's MIOpen holds open
    dst = tempfile.mkdtemp(prefix="mdx_miopen_db_%d_r%d_" % (os.getuid(), rank))
    for name in names:
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    import atexit
    atexit.register(shutil.rmtree, dst, True)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst


def miopen_db_tag():
    """The "<version>-<hash>" part MIOpen puts into its user-db file names, from the loaded library's version string
    (e.g. "3_5_0_20250912-42-1199-g2584e35062"); None when it cannot be determined (then nothing is checked)."""
    try:
        import re
        import torch
        root = os.path.dirname(os.path.dirname(torch.__file__))
        for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "libMIOpen.so"), "/opt/rocm/lib/libMIOpen.so",
                     os.path.join(root, "torch", "lib", "libMIOpen.so.1")):
            if os.path.exists(cand):
                data = open(cand, "rb").read()
                m = re.search(rb"(\d+_\d+_\d+_\d{8}-[0-9A-Za-z-]+-g[0-9a-f]{6,})", data)
                if m:
                    return m.group(1).decode()
    except Exception:  # noqa: BLE001
        pass
    return None
