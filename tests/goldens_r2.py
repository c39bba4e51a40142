"""Loader for the round-2 / round-3 / round-4 fixtures (tests/golden/r2_*.npz, r3_*.npz, r4_*.npz, written by
tests/golden/make_golden_r2.py, make_golden_r3.py and make_golden_r4.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


FULL_CASES = ["r2_full_192x640_b1", "r3_full_192x640_b1_s13", "r4_full_192x640_b1_stereo"]


class FullCase:
    """The BASELINE image size, compact storage: r2_full_192x640_b1 (scales 0 and 2, 81 % auto-masked),
    r3_full_192x640_b1_s13 (scales 1 and 3, 37 % auto-masked) and r4_full_192x640_b1_stereo (frame_ids [0, -1, 1, "s"]:
    BASELINE configs[4], scales 0 and 3, 47 % auto-masked, every reprojection frame the arg-min of 12-24 % of the pixels)."""

    def __init__(self, name="r2_full_192x640_b1"):
        self.name = name
        self.z = load(name)
        self.B, self.H, self.W, self.S = [int(v) for v in self.z["meta"][:4]]
        self.scales = [int(s) for s in self.z["scales"]]
        self.sources_ids = [-1, 1]
        if "sources" in self.z:                       # round 4: the fixture names its source frames ("s" = the stereo frame)
            self.sources_ids = [f if f == "s" else int(f) for f in (str(v) for v in self.z["sources"])]

    def color(self, f, s=0):
        key = "color_u8_%s" % f if s == 0 else "color0_u8_s%d" % s
        return (self.z[key].astype(np.float32) / np.float32(255.0)).astype(np.float32)

    def disp(self, s):
        return self.z["disp_f16_s%d" % s].astype(np.float32)

    def noise(self, s):
        return self.z["noise_f16_s%d" % s].astype(np.float32)

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z
