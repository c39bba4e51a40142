"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["mono_24x40_b2", "border_24x40_b2", "monobugK_24x40_b2", "stereo_16x32_b1",
         "noautomask_16x32_b2", "single_16x32_b2", "stereoonly_16x32_b2", "multi_64x160_b2"]
FULL_CASES = [c for c in CASES if c != "multi_64x160_b2"]


class Case:
    def __init__(self, name):
        self.name = name
        self.z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
        self.B, self.H, self.W, self.S, am, self.n_scales = [int(v) for v in self.z["meta"]]
        self.automask = bool(am)
        self.frame_ids = [f if f == "s" else int(f) for f in self.z["frame_ids"]]
        self.sources_ids = self.frame_ids[1:]

    def color(self, f, s=0):
        """inputs[("color", f, s)]: uint8/255 exactly as the generator formed it."""
        key = "color_u8_%s" % f if s == 0 else "color0_u8_s%d" % s
        return (self.z[key].astype(np.float32) / np.float32(255.0)).astype(np.float32)

    def T(self, f):
        return self.z["T_s" if f == "s" else "T_%s" % f]

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z


def api():
    return dict(np.load(os.path.join(GOLDEN_DIR, "api_ops.npz")))
