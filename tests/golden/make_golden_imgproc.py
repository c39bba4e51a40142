#!/usr/bin/env python3
"""Writes tests/golden/r2_imgproc.npz: inputs and the outputs of **Pillow itself** (the third-party dependency the
reference's loaders call, kitti_mono.py:288-291, 284-285) for the image-preparation path.  Run in a container that has
Pillow; the version is recorded in the file.

    python tests/golden/make_golden_imgproc.py
"""
import os
import random
import sys

import numpy as np
import PIL
from PIL import Image, ImageEnhance

HERE = os.path.dirname(os.path.abspath(__file__))


def natural(rng, h, w):
    lo = rng.random((max(h // 8, 2), max(w // 8, 2), 3))
    img = np.asarray(Image.fromarray((lo * 255).astype(np.uint8)).resize((w, h), Image.BICUBIC)).astype(np.int32)
    img = img + rng.integers(-12, 13, img.shape)
    img[: h // 4, : w // 4] = rng.integers(0, 2, (h // 4, w // 4, 1)) * 255
    return img.clip(0, 255).astype(np.uint8)


def pil_jitter(img, order, b, c, s, hue_shift):
    """torchvision's PIL ColorJitter arithmetic (adjust_brightness / contrast / saturation / hue)."""
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(b)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(c)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(s)
        else:
            hsv = np.array(img.convert("HSV"), dtype=np.uint8)
            hsv[..., 0] = (hsv[..., 0].astype(np.int16) + hue_shift) % 256
            img = Image.fromarray(hsv, "HSV").convert("RGB")
    return img


def main():
    rng = np.random.default_rng(2026)
    out = {"pillow_version": np.array(PIL.__version__)}
    # a KITTI-proportioned frame at 1/5 size (75x248 -> 40x128 pyramid), a ragged one, an upscale
    cases = {"kitti5": ((75, 248), [(40, 128), (20, 64), (10, 32), (5, 16)]), "ragged": ((37, 53), [(16, 24), (37, 20)]),
             "up": ((12, 20), [(24, 40)])}
    for name, ((h, w), outs) in cases.items():
        img = natural(rng, h, w)
        out[name + "_in"] = img
        for (oh, ow) in outs:
            for flip in (0, 1):
                im = Image.fromarray(img)
                if flip:
                    im = im.transpose(Image.FLIP_LEFT_RIGHT)
                out["%s_%dx%d_f%d" % (name, oh, ow, flip)] = np.asarray(im.resize((ow, oh), Image.LANCZOS))
    # resample coefficients as Pillow applies them, observed through unit impulses: row k of an identity-like image
    img = natural(rng, 40, 128)
    out["jit_in"] = img
    r = random.Random(7)
    for k in range(6):
        order = r.sample(range(4), 4)
        b, c, s = (r.uniform(0.8, 1.2) for _ in range(3))
        hue = int(r.uniform(-0.1, 0.1) * 255)
        out["jit%d_params" % k] = np.array(order + [b, c, s, hue], np.float64)
        out["jit%d_out" % k] = np.asarray(pil_jitter(Image.fromarray(img), order, b, c, s, hue))
    # per-pixel maps on a lattice of the colour cube (the exhaustive check runs against the installed Pillow)
    g = np.arange(0, 256, 7, dtype=np.uint8)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 37, 3)
    cube = np.ascontiguousarray(cube)
    out["cube"] = cube
    out["cube_hsv"] = np.asarray(Image.fromarray(cube).convert("HSV"))
    out["cube_L"] = np.asarray(Image.fromarray(cube).convert("L"))
    out["cube_from_hsv"] = np.asarray(Image.fromarray(cube, "HSV").convert("RGB"))
    path = os.path.join(HERE, "r2_imgproc.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; Pillow", PIL.__version__)


if __name__ == "__main__":
    sys.exit(main())
