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
    dst = os.path.join(tempfile.gettempdir(), "mdx_miopen_db_%d_r%d" % (os.getuid(), rank))
    os.makedirs(dst, exist_ok=True)
    # always a pristine copy: MIOpen rewrites entries in place when a find runs, and a copy left behind by an
    # earlier process then steers later runs by that process's noisy timings
    for name in os.listdir(dst):
        os.remove(os.path.join(dst, name))
    for name in os.listdir(src):
        if name.endswith(".txt"):
            shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
