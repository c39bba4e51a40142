"""Host side of csrc/imgproc.hip: the KITTI loaders' per-sample image preparation on the GPU.

What the reference does on a CPU worker per sample (model_loader/kitti_mono.py:335-372): for every frame, four Pillow
Lanczos resizes of the decoded 1242x375 image (kitti_mono.py:288-291), ColorJitter on each, ToTensor on each -- 41 ms
per frame next to 3.7 ms of JPEG decoding.  Here the workers only decode; the batch carries the decoded frames
(uint8, interleaved RGB, padded to a common size) and `image_prep` produces exactly the entries a step reads
(processor.step_reads): ("color", f, 0), ("color_aug", f, 0) for every frame and ("color", 0, s), s = 1..3 -- bit-equal
to Pillow's output divided by 255.

    batch keys consumed:  ("depth_idx", 0) int32 [B, n] / ("depth_val", 0) float32 [B, n]: sparse velodyne ground truth
                          (padding: index h*w) + "depth_hw" [B, 2], CPU  ->  ("depth", 0) [B,1,h,w]
                          ("raw", f)  uint8 [B, hmax, wmax, 3]      "raw_size" int32 [B, 2] (h, w), CPU
                          "raw_flip"  bool [B], CPU                  "raw_jitter" float64 [B, 9], CPU:
                                                                       (enabled, order[4], brightness, contrast,
                                                                        saturation, hue_shift)
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib

JITTER_PARTIALS = 128      # MDX_JITTER_PARTIALS


# mdx_resample_job / mdx_jitter_job as numpy records (same layout: pointers first, no padding) -- a batch has ~110 jobs
# and filling ctypes structures field by field cost 0.5 ms of host time per step; whole columns are assigned instead
RESAMPLE_JOB = np.dtype([("src", "<u8"), ("xbounds", "<u8"), ("xkk", "<u8"), ("ybounds", "<u8"), ("ykk", "<u8"),
                         ("inter", "<u8"), ("dst_u8", "<u8"), ("dst_f32", "<u8"),
                         ("in_h", "<i4"), ("in_w", "<i4"), ("in_stride", "<i4"), ("flip", "<i4"),
                         ("out_h", "<i4"), ("out_w", "<i4"), ("xksize", "<i4"), ("yksize", "<i4"),
                         ("xkc", "<u8"), ("xkc_lead", "<i4"), ("xkc_row", "<i4")])
JITTER_JOB = np.dtype([("src", "<u8"), ("dst_f32", "<u8"), ("dst_u8", "<u8"), ("lsum", "<u8"),
                       ("h", "<i4"), ("w", "<i4"), ("order", "<i4", (4,)), ("hue_shift", "<i4"),
                       ("brightness", "<f4"), ("contrast", "<f4"), ("saturation", "<f4")])
assert RESAMPLE_JOB.itemsize == 112 and JITTER_JOB.itemsize == 72


def _check_u8(t, what):
    _lib.ptr(t, torch.uint8)          # GPU, contiguous, uint8, on the CURRENT device (whose stream the library is given)


Plan = collections.namedtuple("Plan", "ksize bounds kk bounds_ptr kk_ptr cols cols_ptr cols_lead cols_row")


class plan_cache(object):
    """Lanczos plans (Resample.c precompute_coeffs) per (in_size, out_size): built on the host by the library, kept on
    the device.  KITTI raw has five image sizes, so a run holds ~40 small tables."""

    def __init__(self, device, cols=True):
        # cols = False withholds the column-major weight table: the horizontal pass then runs in its gather form (tests)
        self.device, self.plans, self.cols = device, {}, cols

    def get(self, in_size, out_size, cols=False):
        """-> Plan.  cols: the plan will serve the HORIZONTAL pass, whose rows form also reads the weights column-major
        (mdx_resample_plan_cols: a second host pass over the plan and up to ~170 KB on the device) -- built on first request
        only, a plan that only ever serves the vertical pass never pays for it."""
        key = (int(in_size), int(out_size))
        plan = self.plans.get(key)
        if plan is None:
            lib = _lib.lib()
            ksize = lib.mdx_resample_ksize(key[0], key[1])
            if ksize <= 0:
                _lib.check(ksize, "mdx_resample_ksize")
            bounds = np.zeros((key[1], 2), np.int32)
            kk = np.zeros((ksize, key[1]), np.int32)          # tap-major (include/mdx.h)
            _lib.check(lib.mdx_resample_plan(key[0], key[1], bounds.ctypes.data_as(C.c_void_p),
                                             kk.ctypes.data_as(C.c_void_p)), "mdx_resample_plan")
            tb, tk = torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device)
            # (device addresses kept beside the tensors: a batch asks for ~150 plans, data_ptr() is not free)
            plan = self.plans[key] = Plan(ksize, tb, tk, tb.data_ptr(), tk.data_ptr(), None, 0, 0, 0)
            # a plan outlives the call and may next be used from another stream (prefetcher / step): finish its upload now
            torch.cuda.current_stream(self.device).synchronize()
        if cols and self.cols and plan.cols is None:
            # the same weights column-major, zero-padded, both directions: what the rows form of the horizontal pass reads
            # with scalar loads (include/mdx.h, mdx_resample_plan_cols)
            lib = _lib.lib()
            lead, row = C.c_int(0), C.c_int(0)
            _lib.check(lib.mdx_resample_plan_cols(key[0], key[1], C.byref(lead), C.byref(row), None), "mdx_resample_plan_cols")
            kc = np.zeros((2, key[1], row.value), np.int32)
            _lib.check(lib.mdx_resample_plan_cols(key[0], key[1], C.byref(lead), C.byref(row),
                                                  kc.ctypes.data_as(C.c_void_p)), "mdx_resample_plan_cols")
            tc = torch.from_numpy(kc).to(self.device)
            plan = self.plans[key] = plan._replace(cols=tc, cols_ptr=tc.data_ptr(), cols_lead=lead.value, cols_row=row.value)
            torch.cuda.current_stream(self.device).synchronize()
        return plan


def resize_lanczos_multi(plans, sources, sizes, flips, outs):
    """Image.resize((out_w, out_h), Image.LANCZOS) for several blocks of images and several output sizes in ONE call.
    sources: list of uint8 [N, hmax, wmax, 3] (image n occupies [:h_n, :w_n]); sizes [(h, w)] * N and flips [bool] * N
    hold for every block (the frames of a sample share them).  outs: list of (block index, (out_h, out_w), want_u8,
    want_f32) -> list of (uint8 [N,3,out_h,out_w] or None, float32 [N,3,out_h,out_w] = u8 / 255 or None); want_u8 may
    be a uint8 [N,3,out_h,out_w] tensor to write into (e.g. a slice of a larger block)."""
    N = len(sizes)
    for src in sources:
        _check_u8(src, "src")
        if src.dim() != 4 or src.shape[0] != N or src.shape[3] != 3 or len(flips) != N:
            raise _lib.MdxError("resize_lanczos: src [N,h,w,3] with N sizes and N flips expected")
    dev = sources[0].device
    O = len(outs)
    jobs = np.zeros((O, N), RESAMPLE_JOB)
    hl, wl = [int(s[0]) for s in sizes], [int(s[1]) for s in sizes]
    uh, uw = sorted(set(hl)), sorted(set(wl))                 # a batch has one to five distinct frame sizes
    hinv, winv = [uh.index(h) for h in hl], [uw.index(w) for w in wl]
    if uh[0] <= 0 or uw[0] <= 0:
        raise _lib.MdxError("resize_lanczos: empty image in %s" % (sizes,))
    scratch = 0
    for (si, (oh, ow), want_u8, want_f32) in outs:
        scratch += N * 3 * sources[si].shape[1] * int(ow)
    inter = torch.empty(scratch, dtype=torch.uint8, device=dev)
    inter_ptr = inter.data_ptr()
    # per output: base address and per-image step of every pointer column; per (output, distinct size): the plan.  The
    # columns of all outputs are then filled with a handful of array operations (a batch is 72 jobs: filled output by
    # output, field by field, this function took as long on the host as its kernels on the GPU)
    results, cols, xpl, ypl = [], [], [], []
    at = 0
    for (si, (oh, ow), want_u8, want_f32) in outs:
        src = sources[si]
        hmax, wmax = src.shape[1], src.shape[2]
        oh, ow = int(oh), int(ow)
        if uh[-1] > hmax or uw[-1] > wmax:
            raise _lib.MdxError("resize_lanczos: an image of %s does not fit its %dx%d slot" % (sizes, hmax, wmax))
        if torch.is_tensor(want_u8):
            u8, want_u8 = want_u8, True
            _check_u8(u8, "want_u8")
            if tuple(u8.shape) != (N, 3, oh, ow):
                raise _lib.MdxError("resize_lanczos: the uint8 output must be [N,3,%d,%d]" % (oh, ow))
        else:
            u8 = torch.empty(N, 3, oh, ow, dtype=torch.uint8, device=dev) if want_u8 else None
        f32 = torch.empty(N, 3, oh, ow, dtype=torch.float32, device=dev) if want_f32 else None
        results.append((u8, f32))
        cols.append((src.data_ptr(), hmax * wmax * 3, inter_ptr + at, 3 * hmax * ow,
                     u8.data_ptr() if want_u8 else 0, 3 * oh * ow if want_u8 else 0,
                     f32.data_ptr() if want_f32 else 0, 12 * oh * ow if want_f32 else 0, 3 * wmax, oh, ow))
        at += N * 3 * hmax * ow
        xpl.append([plans.get(w, ow, cols=True) for w in uw])
        ypl.append([plans.get(h, oh) for h in uh])
    c = np.array(cols, dtype=np.uint64)                                        # [O, 11]
    idx = np.arange(N, dtype=np.uint64)[None, :]
    jobs["src"] = c[:, 0:1] + idx * c[:, 1:2]
    jobs["inter"] = c[:, 2:3] + idx * c[:, 3:4]
    jobs["dst_u8"] = c[:, 4:5] + idx * c[:, 5:6]
    jobs["dst_f32"] = c[:, 6:7] + idx * c[:, 7:8]
    jobs["in_stride"], jobs["out_h"], jobs["out_w"] = c[:, 8:9], c[:, 9:10], c[:, 10:11]
    jobs["in_h"], jobs["in_w"] = np.array(hl, np.int32)[None, :], np.array(wl, np.int32)[None, :]
    jobs["flip"] = np.array([int(bool(f)) for f in flips], np.int32)[None, :]
    xa = np.array([[(p.bounds_ptr, p.kk_ptr, p.ksize, p.cols_ptr, p.cols_lead, p.cols_row) for p in row] for row in xpl],
                  dtype=np.uint64)[:, winv]                                                                         # [O, N, 6]
    ya = np.array([[(p.bounds_ptr, p.kk_ptr, p.ksize) for p in row] for row in ypl], dtype=np.uint64)[:, hinv]
    jobs["xbounds"], jobs["xkk"], jobs["xksize"] = xa[:, :, 0], xa[:, :, 1], xa[:, :, 2]
    jobs["xkc"], jobs["xkc_lead"], jobs["xkc_row"] = xa[:, :, 3], xa[:, :, 4], xa[:, :, 5]
    jobs["ybounds"], jobs["ykk"], jobs["yksize"] = ya[:, :, 0], ya[:, :, 1], ya[:, :, 2]
    jobs = jobs.reshape(-1)
    _lib.check(_lib.lib().mdx_resample_lanczos_u8(jobs.ctypes.data_as(C.c_void_p), len(jobs), _lib.stream()),
               "mdx_resample_lanczos_u8")
    inter.record_stream(torch.cuda.current_stream(dev))
    return results


def resize_lanczos(plans, src, sizes, flips, out_hw, want_u8=False, want_f32=True):
    """one block, one output size: -> (uint8 [N,3,out_h,out_w] or None, float32 = u8 / 255 or None)."""
    return resize_lanczos_multi(plans, [src], sizes, flips, [(0, out_hw, want_u8, want_f32)])[0]


def color_jitter(src_u8, params, out=None):
    """torchvision's ColorJitter on PIL images, for the images of src_u8 [N,3,h,w] (planar uint8) whose params entry is
    not None: params[n] = (order[4], brightness, contrast, saturation, hue_shift).  Writes float32 u8/255 into
    out[n] ([N,3,h,w] float32; allocated when None -- entries without params are then left unwritten)."""
    _check_u8(src_u8, "src_u8")
    N, _, h, w = src_u8.shape
    dev = src_u8.device
    if out is None:
        out = torch.empty(N, 3, h, w, dtype=torch.float32, device=dev)
    _lib.ptr(out, torch.float32)
    todo = [n for n in range(N) if params[n] is not None]
    if not todo:
        return out
    lsum = torch.empty(len(todo), JITTER_PARTIALS, dtype=torch.int64, device=dev)
    jobs = np.zeros(len(todo), JITTER_JOB)
    t = np.array(todo, dtype=np.uint64)
    jobs["src"] = src_u8.data_ptr() + t * np.uint64(3 * h * w)
    jobs["dst_f32"] = out.data_ptr() + t * np.uint64(12 * h * w)
    jobs["lsum"] = lsum.data_ptr() + np.arange(len(todo), dtype=np.uint64) * np.uint64(8 * JITTER_PARTIALS)
    jobs["h"], jobs["w"] = h, w
    sel = [params[n] for n in todo]
    jobs["order"] = [(list(q[0]) + [4, 4, 4, 4])[:4] for q in sel]
    jobs["brightness"], jobs["contrast"], jobs["saturation"] = [q[1] for q in sel], [q[2] for q in sel], [q[3] for q in sel]
    jobs["hue_shift"] = [int(q[4]) for q in sel]
    if not out.is_contiguous() or out.shape != (N, 3, h, w) or out.dtype != torch.float32:
        raise _lib.MdxError("color_jitter: out must be a contiguous float32 [N,3,h,w] tensor")
    _lib.check(_lib.lib().mdx_color_jitter_u8(jobs.ctypes.data_as(C.c_void_p), len(todo), _lib.stream()),
               "mdx_color_jitter_u8")
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


def to_tensor(src_u8):
    """torchvision's ToTensor on a uint8 tensor of any shape: float32(x) / 255 with the IEEE divide (csrc/imgproc.hip
    to_tensor_kernel).  torch's own `x / 255.0` on the GPU multiplies by the rounded reciprocal when the divisor is a
    Python scalar -- one ulp off for 126 of the 256 byte values, i.e. NOT the numbers the reference's loader produces."""
    if not (torch.is_tensor(src_u8) and src_u8.is_cuda and src_u8.dtype == torch.uint8):
        raise _lib.MdxError("to_tensor: a CUDA uint8 tensor is required")
    src = src_u8.contiguous()
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    if src.numel():
        _lib.check(_lib.lib().mdx_to_tensor_u8(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                               C.c_size_t(src.numel()), _lib.stream()), "mdx_to_tensor_u8")
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
        drop = ("raw_size", "raw_flip", "raw_jitter", "depth_hw", ("depth_idx", 0), ("depth_val", 0))
        out = {k: v for k, v in batch.items() if not (isinstance(k, tuple) and k[0] == "raw") and k not in drop}
        if ("depth_idx", 0) in batch:     # sparse velodyne ground truth -> the dense [B,1,h,w] map the loader's contract names
            hw = batch["depth_hw"].tolist()
            if any(row != hw[0] for row in hw):     # the sparse indices of every sample address ONE [gh, gw] grid
                raise _lib.MdxError("image_prep: the ground-truth maps of a batch must share one size, got %s" % (sorted(set(map(tuple, hw))),))
            gh, gw = (int(v) for v in hw[0])
            idx = batch[("depth_idx", 0)].to(self.device, non_blocking=True).long()
            val = batch[("depth_val", 0)].to(self.device, non_blocking=True)
            buf = torch.zeros(idx.shape[0], gh * gw + 1, device=self.device)       # + one slot that takes the padding
            buf.scatter_(1, idx, val)
            out[("depth", 0)] = buf[:, :gh * gw].reshape(-1, 1, gh, gw).contiguous()
        sources = []
        for f in self.frame_ids:
            raw = batch[("raw", f)]
            sources.append(raw if raw.is_cuda else raw.to(self.device, non_blocking=True))
        # every frame at scale 0 (uint8 too when some sample is jittered), the target frame at scales 1..: one call
        B, F = len(sizes), len(sources)
        u8 = torch.empty(F * B, 3, self.h, self.w, dtype=torch.uint8, device=sources[0].device) if any_jitter else None
        outs = [(i, (self.h, self.w), u8[i * B:(i + 1) * B] if any_jitter else False, True) for i in range(F)]
        target = self.frame_ids.index(0)
        outs += [(target, (self.h >> s, self.w >> s), False, True) for s in range(1, self.scales)]
        res = resize_lanczos_multi(self.plans, sources, sizes, flips, outs)
        for i, f in enumerate(self.frame_ids):
            out[("color", f, 0)] = res[i][1]
            # without augmentation the reference hands the same numbers to both entries (kitti_mono.py:357-366)
            out[("color_aug", f, 0)] = res[i][1]
        for s in range(1, self.scales):
            out[("color", 0, s)] = res[len(sources) + s - 1][1]
        if any_jitter:
            # all frames' jitter in one call; samples without a draw run the empty chain (= u8 / 255, the same numbers)
            chain = [p if p is not None else ([4, 4, 4, 4], 1.0, 1.0, 1.0, 0) for p in params] * len(sources)
            aug = color_jitter(u8, chain)
            for i, f in enumerate(self.frame_ids):
                out[("color_aug", f, 0)] = aug[i * B:(i + 1) * B]
        for raw in sources:
            raw.record_stream(torch.cuda.current_stream(raw.device))
        return out
