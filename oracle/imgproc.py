"""CPU restatement (numpy, integer / byte arithmetic) of the image preparation the reference's KITTI loaders run per
sample -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Reference call sites:
    model_loader/kitti_mono.py:288-291   transforms.Resize((H >> s, W >> s), interpolation=Image.ANTIALIAS), s = 0..3,
                                         each scale resized from the ORIGINAL image (kitti_mono.py:349-350, 361-362)
    model_loader/kitti_mono.py:302-303   image.transpose(Image.FLIP_LEFT_RIGHT)
    model_loader/kitti_mono.py:284-285   ColorJitter (0.8..1.2 brightness / contrast / saturation, +-0.1 hue)
    model_loader/kitti_mono.py:283, 351  transforms.ToTensor(): uint8 HWC -> float32 CHW, x / 255
The arithmetic itself lives in a third-party dependency that is not under /root/reference: **Pillow** (the reference
pins no version; this image has Pillow 12.2.0, whose Image.LANCZOS is the filter the removed ANTIALIAS named) and
torchvision's PIL backend (ImageEnhance.Brightness / Contrast / Color, and the HSV round trip for hue).  The functions
below restate Pillow's published algorithms:
    resample      src/libImaging/Resample.c   precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc,
                                              ImagingResampleVertical_8bpc (two passes, uint8 in between, 22-bit weights)
    to_L          src/libImaging/Convert.c    rgb2l:  (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
    blend         src/libImaging/Blend.c      ImagingBlend, float32 arithmetic, truncation / clipping
    rgb2hsv / hsv2rgb  src/libImaging/Convert.c   (float / double mix, see the functions)
    enhance ops   src/PIL/ImageEnhance.py     Brightness / Contrast / Color degenerates
Pinned: tests/test_imgproc_cpu.py checks every function against the installed Pillow itself -- exhaustively for the
per-pixel maps (all 2^24 RGB and HSV triples, all 2^16 blend pairs at sampled factors), at KITTI's sizes and ragged
sizes for the resampler -- and against committed Pillow-made vectors (tests/golden/r2_imgproc.npz).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size).
    -> ksize, bounds int32 [out,2] (first tap, tap count), kk int32 [out, ksize] (22-bit fixed point)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """one 8bpc pass along `axis` of a uint8 array: clip8((2^21 + sum_k px * kk) >> 22)."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    ksize = kk.shape[1]
    idx = np.minimum(bounds[:, :1] + np.arange(ksize)[None, :], img.shape[0] - 1)      # padded taps carry weight 0
    acc = np.full((len(bounds),) + img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
    for k in range(ksize):
        acc += img[idx[:, k]] * kk[:, k].astype(np.int64).reshape((-1,) + (1,) * (img.ndim - 1))
    out = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resample_lanczos(img, out_h, out_w, flip=False):
    """img uint8 [h, w, 3] -> uint8 [out_h, out_w, 3]: Image.resize((out_w, out_h), Image.LANCZOS) of the (optionally
    left-right flipped) image.  Horizontal pass first, then vertical, each skipped when the size is unchanged
    (Resample.c ImagingResample)."""
    img = np.ascontiguousarray(img[:, ::-1] if flip else img)
    h, w = img.shape[:2]
    if out_w != w:
        _, b, k = resample_coeffs(w, out_w)
        img = _pass(img, b, k, 1)
    if out_h != h:
        _, b, k = resample_coeffs(h, out_h)
        img = _pass(img, b, k, 0)
    return img


def to_L(rgb):
    """Convert.c rgb2l."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(a, b, alpha):
    """Blend.c ImagingBlend(a, b, alpha): a + alpha * (b - a) in float32; truncation inside [0,1], clipping outside."""
    alpha = np.float32(alpha)
    t = a.astype(np.float32) + alpha * (b.astype(np.int32) - a.astype(np.int32)).astype(np.float32)
    if 0.0 <= float(alpha) <= 1.0:
        return t.astype(np.uint8)                      # values are inside [0,255]: C's (UINT8) truncation
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def rgb2hsv(rgb):
    """Convert.c rgb2hsv_row: float32 variables, double-precision expressions where a double literal takes part."""
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    grey = maxc == minc
    cr = np.where(grey, 1, maxc - minc).astype(np.float32)
    mx = np.where(maxc == 0, 1, maxc).astype(np.float32)
    s = cr / mx
    rc = (maxc - r).astype(np.float32) / cr
    gc = (maxc - g).astype(np.float32) / cr
    bc = (maxc - b).astype(np.float32) / cr
    h = np.where(r == maxc, bc - gc,                                                     # float - float
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32))).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    out = np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1)
    return out.astype(np.uint8)


def _c_round(x):
    """C round(): half away from zero (x >= 0 here)."""
    return np.floor(x + 0.5).astype(np.int32)


def hsv2rgb(hsv):
    """Convert.c hsv2rgb."""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float32).astype(np.float64)
    p = np.clip(_c_round(vf * (1.0 - fs)), 0, 255)
    q = np.clip(_c_round(vf * (1.0 - fs * f)), 0, 255)
    t = np.clip(_c_round(vf * (1.0 - fs * (1.0 - f))), 0, 255)
    sel = i % 6
    table = {0: (v, t, p), 1: (q, v, p), 2: (p, v, t), 3: (p, q, v), 4: (t, p, v), 5: (v, p, q)}
    out = np.zeros(hsv.shape[:-1] + (3,), np.int32)
    for k, (rr, gg, bb) in table.items():
        m = sel == k
        out[..., 0] = np.where(m, rr, out[..., 0])
        out[..., 1] = np.where(m, gg, out[..., 1])
        out[..., 2] = np.where(m, bb, out[..., 2])
    grey = s == 0
    for c in range(3):
        out[..., c] = np.where(grey, v, out[..., c])
    return out.astype(np.uint8)


def adjust_brightness(rgb, factor):
    """ImageEnhance.Brightness: blend(black, image, factor)."""
    return blend(np.zeros_like(rgb), rgb, factor)


def contrast_mean(rgb):
    """ImageEnhance.Contrast: int(ImageStat.Stat(image.convert("L")).mean[0] + 0.5) -- exact integer sum / count."""
    L = to_L(rgb)
    return int(float(int(L.astype(np.int64).sum())) / L.size + 0.5)


def adjust_contrast(rgb, factor):
    return blend(np.full_like(rgb, contrast_mean(rgb)), rgb, factor)


def adjust_saturation(rgb, factor):
    """ImageEnhance.Color: blend(image.convert("L").convert("RGB"), image, factor)."""
    return blend(np.repeat(to_L(rgb)[..., None], 3, -1), rgb, factor)


def adjust_hue(rgb, shift):
    """torchvision's PIL hue: H channel of convert("HSV") += uint8(hue_factor * 255) with wrap-around, back to RGB.
    `shift` = int(hue_factor * 255) (the build's loader draws it, model_loader/kitti.py ColorJitter)."""
    hsv = rgb2hsv(rgb)
    hsv[..., 0] = ((hsv[..., 0].astype(np.int32) + int(shift)) % 256).astype(np.uint8)
    return hsv2rgb(hsv)


def color_jitter(rgb, order, brightness, contrast, saturation, hue_shift):
    """the four adjustments in `order` (0 brightness, 1 contrast, 2 saturation, 3 hue), uint8 after each."""
    for op in order:
        if op == 0:
            rgb = adjust_brightness(rgb, brightness)
        elif op == 1:
            rgb = adjust_contrast(rgb, contrast)
        elif op == 2:
            rgb = adjust_saturation(rgb, saturation)
        else:
            rgb = adjust_hue(rgb, hue_shift)
    return rgb


def to_tensor(rgb):
    """ToTensor: uint8 [h,w,3] -> float32 [3,h,w], x / 255 (correctly rounded float32 division)."""
    return (np.ascontiguousarray(rgb.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)).astype(np.float32)
