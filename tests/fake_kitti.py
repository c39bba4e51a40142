"""Builds a tiny synthetic KITTI-raw tree (JPEG frames, calibration files, velodyne scans) for loader tests."""
import os

import numpy as np
from PIL import Image

CAM2CAM = """calib_time: 09-Jan-2012 13:57:47
S_rect_02: 1.242000e+03 3.750000e+02
R_rect_00: 1 0 0 0 1 0 0 0 1
P_rect_02: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P_rect_03: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
"""
VELO2CAM = """calib_time: 15-Mar-2012 11:37:16
R: 0 -1 0 0 0 -1 1 0 0
T: 0 -0.08 -0.27
"""


def make(root, n_frames=5, seed=0):
    rng = np.random.RandomState(seed)
    day, drive = "2011_09_26", "2011_09_26/2011_09_26_drive_0001_sync"
    os.makedirs(os.path.join(root, day), exist_ok=True)
    open(os.path.join(root, day, "calib_cam_to_cam.txt"), "w").write(CAM2CAM)
    open(os.path.join(root, day, "calib_velo_to_cam.txt"), "w").write(VELO2CAM)
    for cam in (2, 3):
        d = os.path.join(root, drive, "image_0%d/data" % cam)
        os.makedirs(d, exist_ok=True)
        for i in range(n_frames):
            img = (rng.rand(375 // 5, 1242 // 6, 3) * 255).astype(np.uint8)
            Image.fromarray(img).resize((1242, 375), Image.BILINEAR).save(os.path.join(d, "%010d.jpg" % i))
    v = os.path.join(root, drive, "velodyne_points/data")
    os.makedirs(v, exist_ok=True)
    for i in range(n_frames):
        n = 20000
        pts = np.stack([rng.uniform(3, 70, n), rng.uniform(-20, 20, n), rng.uniform(-2, 1.5, n), np.ones(n)], 1)
        pts.astype(np.float32).tofile(os.path.join(v, "%010d.bin" % i))
    return ["%s %d l" % (drive, i) for i in range(1, n_frames - 1)]
