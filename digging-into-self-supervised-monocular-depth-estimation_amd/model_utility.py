"""Split-file, calibration and velodyne helpers (reference: model_utility.py:18-197), numpy only.

    readlines          model_utility.py:18-22      read_calib / read_cam2cam   model_utility.py:59-104
    read_velodyne_points  model_utility.py:108-115  point2depth                 model_utility.py:128-197
"""
import os

import numpy as np


def readlines(datapath):
    with open(datapath, "r") as f:
        return f.read().splitlines()


def read_calib(path):
    """KITTI calibration file -> {key: float array | string} (reference read_velo2cam, model_utility.py:84-104)."""
    data = {}
    with open(path, "r") as f:
        for line in f:
            if ":" not in line:
                continue
            key, value = line.split(":", 1)
            value = value.strip()
            try:
                data[key] = np.array([float(x) for x in value.split()])
            except ValueError:
                data[key] = value
    return data


read_velo2cam = read_calib


def read_cam2cam(path):
    """4x4 intrinsics of the left / right colour cameras (model_utility.py:59-80)."""
    data = read_calib(path)
    out = []
    for key in ("P_rect_02", "P_rect_03"):
        K = np.eye(4, dtype=np.float32)
        K[:3, :3] = data[key].reshape(3, 4)[:3, :3]
        out.append(K)
    return out[0], out[1]


def read_velodyne_points(filename):
    points = np.fromfile(filename, dtype=np.float32).reshape(-1, 4)
    points[:, 3] = 1.0
    return points


def point2depth(calib_path, point_path, cam=2, vel_depth=False):
    """Projects a velodyne scan into camera `cam` -> sparse depth map [h, w] (model_utility.py:128-197).
    Where several points fall on one pixel the nearest one wins -- with the reference's own index arithmetic, see below."""
    cam2cam = read_calib(os.path.join(calib_path, "calib_cam_to_cam.txt"))
    velo2cam = read_calib(os.path.join(calib_path, "calib_velo_to_cam.txt"))
    velo2cam = np.hstack((velo2cam["R"].reshape(3, 3), velo2cam["T"][..., np.newaxis]))
    velo2cam = np.vstack((velo2cam, np.array([0, 0, 0, 1.0])))
    im_shape = cam2cam["S_rect_02"][::-1].astype(np.int32)
    R_cam2rect = np.eye(4)
    R_cam2rect[:3, :3] = cam2cam["R_rect_00"].reshape(3, 3)
    P_rect = cam2cam["P_rect_0" + str(cam)].reshape(3, 4)
    P_velo2im = P_rect @ R_cam2rect @ velo2cam
    velo = read_velodyne_points(point_path)
    velo = velo[velo[:, 0] >= 0, :]
    pts = (P_velo2im @ velo.T).T
    pts[:, :2] = pts[:, :2] / pts[:, 2][..., np.newaxis]
    if vel_depth:
        pts[:, 2] = velo[:, 0]
    pts[:, 0] = np.round(pts[:, 0]) - 1
    pts[:, 1] = np.round(pts[:, 1]) - 1
    ok = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < im_shape[1]) & (pts[:, 1] < im_shape[0])
    pts = pts[ok]
    depth = np.zeros(tuple(im_shape[:2]))
    xs, ys = pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64)
    depth[ys, xs] = pts[:, 2]        # several points on one pixel: the last one stands (numpy fancy assignment)
    # The reference then walks the points that share a LINEAR INDEX (model_utility.py:187-194) and gives the pixel of
    # the first of them the smallest depth of the group.  Its index is row*(n-1) + col - 1 (sub2ind, :119-124), not
    # row*n + col: besides true duplicates, (row, 0) and (row-1, n-1) share an index -- then the first point's pixel
    # receives the minimum over both pixels' points and the other pixel keeps its last-written value.  Reproduced as is
    # (evaluation numbers are defined by it); vectorised: groups by np.unique, minimum per group, first member's pixel.
    # (the keys are small integers: bincount finds the few points that share an index, and only those are sorted)
    lin = ys * (depth.shape[1] - 1) + xs                             # the reference's index + 1 (>= 0)
    counts = np.bincount(lin, minlength=depth.shape[0] * depth.shape[1] + 1)
    dup = counts[lin] > 1
    if dup.any():
        order = np.flatnonzero(dup)                                  # ascending point order: return_index = first member
        _, first, inverse = np.unique(lin[order], return_index=True, return_inverse=True)
        mins = np.full(len(first), np.inf)
        np.minimum.at(mins, inverse.reshape(-1), pts[order, 2])
        head = order[first]
        depth[ys[head], xs[head]] = mins
    depth[depth < 0] = 0
    return depth


def resize_nearest(depth, out_hw):
    """skimage.transform.resize(depth, out_hw, order=0, preserve_range=True, mode='constant') equivalent."""
    h, w = depth.shape
    H, W = out_hw
    if (h, w) == (H, W):
        return depth
    ys = np.clip(np.round((np.arange(H) + 0.5) * h / H - 0.5).astype(np.int64), 0, h - 1)
    xs = np.clip(np.round((np.arange(W) + 0.5) * w / W - 0.5).astype(np.int64), 0, w - 1)
    return depth[ys][:, xs]
