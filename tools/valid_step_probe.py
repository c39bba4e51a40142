#!/usr/bin/env python3
"""Photometric time of a VALIDATION step (torch.no_grad(), model_train.py:75-79 of the reference) on bench.py's own batch: the
forward-only form of the all-scale kernel + the prologue, timed by the library's HIP events; and the whole no-grad step."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

bench = importlib.import_module("bench")
importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import functional as F  # noqa: E402
from model_train import trainer  # noqa: E402

opt = bench.make_opt(12)
opt.graph = False
tr = trainer(opt)
tr.setting.set_train()
inputs = bench.one_batch(tr.setting, tr.device)
for _ in range(5):                       # a few training steps first: the disparities of a net that has moved off its init
    tr._eager_step(inputs)
tr.setting.set_valid()
with torch.no_grad():
    for _ in range(5):
        tr.batch_process(inputs)
    torch.cuda.synchronize()
    F.TIMING = {"fwd": [], "bwd": [], "train": []}
    t0 = time.perf_counter()
    for _ in range(20):
        tr.batch_process(inputs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    summ = F.timing_summary(F.TIMING)
    F.TIMING = None
print("validation step (no grad): %.2f ms; library-timed kernels (us, launches):" % (1e3 * dt), {k: (round(v[0], 1), v[1]) for k, v in summ.items() if v[1]})
