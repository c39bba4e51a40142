#!/usr/bin/env python3
"""What one training sample costs on the host and what the image preparation costs on the GPU.

    python tools/loader_cost.py [--samples 24] [--batch 12] [--reps 20]

A synthetic KITTI-raw tree (1242x375 JPEGs) is written to a temporary directory; then
  1. KITTIDataset.__getitem__ on one core, Pillow path (decode + 4 Lanczos resizes + jitter + ToTensor per frame) and
     gpu_prep path (decode only) -> ms per sample, samples/s per core;
  2. mdx.imgproc.image_prep on a collated batch of decoded frames -> us per batch (HIP events), per-kernel split.
"""
import argparse
import importlib
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=24)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    a = ap.parse_args()
    import fake_kitti
    from model_loader import KITTIMonoDataset_v2
    from model_loader.kitti import collate_raw
    from model_tool.processor import step_reads
    torch.set_num_threads(1)
    out = {"cores_used": 1, "host_cores": os.cpu_count()}
    with tempfile.TemporaryDirectory() as root:
        names = fake_kitti.make(root, n_frames=a.samples + 2)
        frames = [0, -1, 1]
        for mode in ("pillow", "pillow_uint8", "gpu_prep"):
            ds = KITTIMonoDataset_v2(root, names, True, frames, a.height, a.width, "jpg", 4)
            ds.load_depth = False                      # the velodyne projection is the same in every mode
            ds.uint8 = mode == "pillow_uint8"
            ds.gpu_prep = mode == "gpu_prep"
            random.seed(0)
            ds[0]
            t0 = time.perf_counter()
            for i in range(a.samples):
                ds[i % len(ds)]
            ms = 1e3 * (time.perf_counter() - t0) / a.samples
            out["host_ms_per_sample_" + mode] = round(ms, 2)
            out["host_samples_per_s_per_core_" + mode] = round(1e3 / ms, 1)
        if torch.cuda.is_available():
            from mdx import imgproc
            ds.gpu_prep = True
            random.seed(1)
            batch = collate_raw([ds[i % len(ds)] for i in range(a.batch)], step_reads)
            batch = {k: (v.cuda() if isinstance(k, tuple) and k[0] == "raw" else v) for k, v in batch.items()}
            prep = imgproc.image_prep(a.height, a.width, frames, 4, "cuda:0")
            jittered = int(batch["raw_jitter"][:, 0].sum())
            for _ in range(3):
                prep(batch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(a.reps):
                prep(batch)
            e1.record()
            host_ms = 1e3 * (time.perf_counter() - t0) / a.reps
            e1.synchronize()
            out.update({"gpu_prep_us_per_batch": round(1e3 * e0.elapsed_time(e1) / a.reps, 1), "batch": a.batch,
                        "jittered_samples_in_batch": jittered, "gpu_prep_host_ms_per_batch": round(host_ms, 2)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
