"""Host side of csrc/imgproc.hip: the KITTI loaders' per-sample image preparation on the GPU.

What the reference does on a CPU worker per sample (model_loader/kitti_mono.py:335-372): for every frame, four Pillow
Lanczos resizes of the decoded 1242x375 image (kitti_mono.py:288-291), ColorJitter on each, ToTensor on each -- 41 ms
per frame next to 3.7 ms of JPEG decoding.  Here the workers only decode; the batch carries the decoded frames
(uint8, interleaved RGB, padded to a common size) and `image_prep` produces exactly the entries a step reads
(processor.step_reads): ("color", f, 0), ("color_aug", f, 0) for every frame and ("color", 0, s), s = 1..3 -- bit-equal
to Pillow's output divided by 255.

    batch keys consumed:  ("raw", f)  uint8 [B, hmax, wmax, 3]      "raw_size" int32 [B, 2] (h, w), CPU
                          "raw_flip"  bool [B], CPU                  "raw_jitter" float64 [B, 9], CPU:
                                                                       (enabled, order[4], brightness, contrast,
                                                                        saturation, hue_shift)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

IMG_JOBS = 32


class ResampleJob(C.Structure):      # mdx_resample_job
    _fields_ = [("src", C.c_void_p), ("xbounds", C.c_void_p), ("xkk", C.c_void_p), ("ybounds", C.c_void_p),
                ("ykk", C.c_void_p), ("inter", C.c_void_p), ("dst_u8", C.c_void_p), ("dst_f32", C.c_void_p),
                ("in_h", C.c_int32), ("in_w", C.c_int32), ("in_stride", C.c_int32), ("flip", C.c_int32),
                ("out_h", C.c_int32), ("out_w", C.c_int32), ("xksize", C.c_int32), ("yksize", C.c_int32)]


class JitterJob(C.Structure):        # mdx_jitter_job
    _fields_ = [("src", C.c_void_p), ("dst_f32", C.c_void_p), ("dst_u8", C.c_void_p), ("lsum", C.c_void_p),
                ("h", C.c_int32), ("w", C.c_int32), ("order", C.c_int32 * 4), ("hue_shift", C.c_int32),
                ("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float)]


def _check_u8(t, what):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise _lib.MdxError("%s must be a contiguous uint8 tensor on the GPU (there is no CPU fallback)" % what)


class plan_cache(object):
    """Lanczos plans (Resample.c precompute_coeffs) per (in_size, out_size): built on the host by the library, kept on
    the device.  KITTI raw has five image sizes, so a run holds ~40 small tables."""

    def __init__(self, device):
        self.device, self.plans = device, {}

    def get(self, in_size, out_size):
        key = (int(in_size), int(out_size))
        if key not in self.plans:
            lib = _lib.lib()
            ksize = lib.mdx_resample_ksize(key[0], key[1])
            if ksize <= 0:
                _lib.check(ksize, "mdx_resample_ksize")
            bounds = np.zeros((key[1], 2), np.int32)
            kk = np.zeros((key[1], ksize), np.int32)
            _lib.check(lib.mdx_resample_plan(key[0], key[1], bounds.ctypes.data_as(C.c_void_p),
                                             kk.ctypes.data_as(C.c_void_p)), "mdx_resample_plan")
            self.plans[key] = (ksize, torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device))
        return self.plans[key]


def resize_lanczos(plans, src, sizes, flips, out_hw, want_u8=False, want_f32=True):
    """Image.resize((out_w, out_h), Image.LANCZOS) of N images in one call.
    src uint8 [N, hmax, wmax, 3] (image n occupies [:h_n, :w_n]); sizes [(h, w)] * N; flips [bool] * N.
    -> (uint8 [N,3,out_h,out_w] or None, float32 [N,3,out_h,out_w] = u8 / 255 or None)."""
    _check_u8(src, "src")
    N, hmax, wmax = src.shape[0], src.shape[1], src.shape[2]
    if src.shape[3] != 3 or len(sizes) != N or len(flips) != N:
        raise _lib.MdxError("resize_lanczos: src [N,h,w,3] with N sizes and N flips expected")
    oh, ow = int(out_hw[0]), int(out_hw[1])
    dev = src.device
    u8 = torch.empty(N, 3, oh, ow, dtype=torch.uint8, device=dev) if want_u8 else None
    f32 = torch.empty(N, 3, oh, ow, dtype=torch.float32, device=dev) if want_f32 else None
    inter = torch.empty(N, 3 * hmax * ow, dtype=torch.uint8, device=dev)
    jobs = (ResampleJob * N)()
    keep = []
    for n in range(N):
        h, w = int(sizes[n][0]), int(sizes[n][1])
        if not (0 < h <= hmax and 0 < w <= wmax):
            raise _lib.MdxError("resize_lanczos: image %d has size %dx%d inside a %dx%d slot" % (n, h, w, hmax, wmax))
        kx, xb, xk = plans.get(w, ow)
        ky, yb, yk = plans.get(h, oh)
        keep += [xb, xk, yb, yk]
        j = jobs[n]
        j.src = src[n].data_ptr()
        j.xbounds, j.xkk, j.ybounds, j.ykk = xb.data_ptr(), xk.data_ptr(), yb.data_ptr(), yk.data_ptr()
        j.inter = inter[n].data_ptr()
        j.dst_u8 = u8[n].data_ptr() if want_u8 else None
        j.dst_f32 = f32[n].data_ptr() if want_f32 else None
        j.in_h, j.in_w, j.in_stride, j.flip = h, w, 3 * wmax, int(bool(flips[n]))
        j.out_h, j.out_w, j.xksize, j.yksize = oh, ow, kx, ky
    _lib.check(_lib.lib().mdx_resample_lanczos_u8(jobs, N, _lib.stream()), "mdx_resample_lanczos_u8")
    inter.record_stream(torch.cuda.current_stream(dev))
    return u8, f32


def color_jitter(src_u8, params, out=None):
    """torchvision's ColorJitter on PIL images, for the images of src_u8 [N,3,h,w] (planar uint8) whose params entry is
    not None: params[n] = (order[4], brightness, contrast, saturation, hue_shift).  Writes float32 u8/255 into
    out[n] ([N,3,h,w] float32; allocated when None -- entries without params are then left unwritten)."""
    _check_u8(src_u8, "src_u8")
    N, _, h, w = src_u8.shape
    dev = src_u8.device
    if out is None:
        out = torch.empty(N, 3, h, w, dtype=torch.float32, device=dev)
    todo = [n for n in range(N) if params[n] is not None]
    if not todo:
        return out
    lsum = torch.empty(len(todo), dtype=torch.int64, device=dev)
    jobs = (JitterJob * len(todo))()
    for i, n in enumerate(todo):
        order, b, c, s, hue = params[n]
        j = jobs[i]
        j.src, j.dst_f32, j.dst_u8, j.lsum = src_u8[n].data_ptr(), out[n].data_ptr(), None, lsum[i].data_ptr()
        j.h, j.w, j.hue_shift = h, w, int(hue)
        for k in range(4):
            j.order[k] = int(order[k]) if k < len(order) else 4
        j.brightness, j.contrast, j.saturation = float(b), float(c), float(s)
    _lib.check(_lib.lib().mdx_color_jitter_u8(jobs, len(todo), _lib.stream()), "mdx_color_jitter_u8")
    lsum.record_stream(torch.cuda.current_stream(dev))
    return out


def color_convert(src_u8, mode):
    """Pillow's convert() on planar uint8 [3, n]: mode "hsv" (RGB->HSV), "rgb" (HSV->RGB), "L" (RGB->L, [n] out)."""
    _check_u8(src_u8, "src_u8")
    code = {"hsv": 0, "rgb": 1, "L": 2}[mode]
    n = src_u8.shape[1]
    dst = torch.empty((n,) if code == 2 else (3, n), dtype=torch.uint8, device=src_u8.device)
    _lib.check(_lib.lib().mdx_color_convert_u8(code, C.c_void_p(src_u8.data_ptr()), C.c_void_p(dst.data_ptr()),
                                               C.c_size_t(n), _lib.stream()), "mdx_color_convert_u8")
    return dst


def jitter_params(row):
    """one row of the batch's "raw_jitter" -> color_jitter's params entry (None when the sample is not jittered)."""
    row = [float(x) for x in row]
    if row[0] == 0.0:
        return None
    return ([int(x) for x in row[1:5]], row[5], row[6], row[7], int(row[8]))


class image_prep(object):
    """Turns the decoded frames of a batch into the entries a training step reads (see the module docstring)."""

    def __init__(self, height, width, frame_ids, scales, device):
        self.h, self.w, self.frame_ids, self.scales, self.device = height, width, list(frame_ids), scales, device
        self.plans = plan_cache(device)

    @staticmethod
    def wanted(batch):
        return any(isinstance(k, tuple) and k[0] == "raw" for k in batch)

    def __call__(self, batch):
        if not self.wanted(batch):
            return batch
        sizes = [tuple(int(v) for v in row) for row in batch["raw_size"].tolist()]
        flips = [bool(v) for v in batch["raw_flip"].tolist()]
        params = [jitter_params(row) for row in batch["raw_jitter"].tolist()]
        any_jitter = any(p is not None for p in params)
        out = {k: v for k, v in batch.items()
               if not (isinstance(k, tuple) and k[0] == "raw") and k not in ("raw_size", "raw_flip", "raw_jitter")}
        for f in self.frame_ids:
            raw = batch[("raw", f)]
            if not raw.is_cuda:
                raw = raw.to(self.device, non_blocking=True)
            u8, f32 = resize_lanczos(self.plans, raw, sizes, flips, (self.h, self.w), want_u8=any_jitter)
            out[("color", f, 0)] = f32
            if any_jitter:
                aug = f32.clone()
                color_jitter(u8, params, aug)
                out[("color_aug", f, 0)] = aug
            else:
                out[("color_aug", f, 0)] = f32       # the reference's identity branch: the same numbers (kitti_mono.py:357-366)
            if f == 0:
                for s in range(1, self.scales):
                    out[("color", 0, s)] = resize_lanczos(self.plans, raw, sizes, flips, (self.h >> s, self.w >> s))[1]
            raw.record_stream(torch.cuda.current_stream(raw.device))
        return out
