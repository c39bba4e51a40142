"""Loader for the round-2 fixtures (tests/golden/r2_*.npz, written by tests/golden/make_golden_r2.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


class FullCase:
    """r2_full_192x640_b1: the BASELINE image size, compact storage."""

    def __init__(self):
        self.z = load("r2_full_192x640_b1")
        self.B, self.H, self.W, self.S = [int(v) for v in self.z["meta"][:4]]
        self.scales = [int(s) for s in self.z["scales"]]
        self.sources_ids = [-1, 1]

    def color(self, f, s=0):
        key = "color_u8_%s" % f if s == 0 else "color0_u8_s%d" % s
        return (self.z[key].astype(np.float32) / np.float32(255.0)).astype(np.float32)

    def disp(self, s):
        return self.z["disp_f16_s%d" % s].astype(np.float32)

    def noise(self, s):
        return self.z["noise_f16_s%d" % s].astype(np.float32)

    def __getitem__(self, k):
        return self.z[k]
