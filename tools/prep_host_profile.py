#!/usr/bin/env python3
"""cProfile of the host side of mdx.imgproc.image_prep on one synthetic KITTI batch (where do its ~0.25 ms go?)."""
import cProfile
import importlib
import os
import pstats
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
import fake_kitti  # noqa: E402
from mdx import imgproc  # noqa: E402
from model_loader.kitti import KITTIMonoDataset_v2, collate_raw  # noqa: E402
from model_tool.processor import step_reads  # noqa: E402

frames = [0, -1, 1]
with tempfile.TemporaryDirectory() as root:
    names = fake_kitti.make(root, n_frames=14)
    ds = KITTIMonoDataset_v2(root, names, True, frames, 192, 640, "jpg", 4)
    ds.load_depth = False
    ds.gpu_prep = True
    random.seed(1)
    raw = collate_raw([ds[i % len(ds)] for i in range(12)], step_reads)
raw = {k: (v.cuda() if isinstance(k, tuple) and k[0] == "raw" else v) for k, v in raw.items()}
prep = imgproc.image_prep(192, 640, frames, 4, "cuda:0")
for _ in range(5):
    prep(raw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    prep(raw)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
