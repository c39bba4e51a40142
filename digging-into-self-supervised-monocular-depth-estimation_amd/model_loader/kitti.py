"""KITTI raw datasets (reference: model_loader/kitti_mono.py:258-375, kitti_stereo.py:168-306).

One implementation, `KITTIDataset`, produces the dictionary contract the step driver consumes:
    ("color", f, s), ("color_aug", f, s)  [3, H>>s, W>>s] float32 in [0,1]      f in frame_ids, s in 0..3
    ("K", s), ("inv_K", s)                [4,4]
    ("depth", 0)                          [1,375,1242] velodyne ground truth (0 = no return)
    "stereo"                              [4,4]  (only with "s" in frame_ids)
`KITTIMonoDataset_v2` / `KITTIMonoStereoDataset` keep the reference constructor signatures.

Differences, stated: Pillow >= 10 (Image.LANCZOS instead of the removed ANTIALIAS); colour jitter is drawn per
sample (the reference draws it once per dataset through a removed torchvision API); `k_mode="reference_mono"`
reproduces the mono loader's intrinsics exactly (row 1 scaled by WIDTH and floored, kitti_mono.py:326-327 ->
[[371,0,320],[0,1228,320]] at 640x192), `k_mode="scaled"` is the stereo loader's form (kitti_stereo.py:236-246).
"""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils.data import Dataset

from model_utility import point2depth, resize_nearest

SIDE_MAP = {"2": 2, "3": 3, "l": 2, "r": 3}
K_NORM = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def to_tensor(img, uint8=False):
    """ToTensor: [3,H,W] float32 in [0,1] (x / 255).  uint8=True leaves the division to the consumer
    (compute.forward_depth does it on the GPU, same correctly rounded x / 255): a quarter of the bytes go through the
    worker -> shared memory -> pinned memory -> PCIe pipeline."""
    if uint8:   # numpy's transposing copy: 0.16 ms; torch's permute().contiguous() on uint8 takes 48 ms for 640x192
        return torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1)))
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)


class ColorJitter(object):
    """brightness / contrast / saturation in [0.8,1.2], hue in [-0.1,0.1], random order -- one draw per sample."""

    def __init__(self, rng):
        self.b, self.c, self.s = (rng.uniform(0.8, 1.2) for _ in range(3))
        self.h = rng.uniform(-0.1, 0.1)
        self.order = rng.sample(range(4), 4)

    def __call__(self, img):
        for op in self.order:
            if op == 0:
                img = ImageEnhance.Brightness(img).enhance(self.b)
            elif op == 1:
                img = ImageEnhance.Contrast(img).enhance(self.c)
            elif op == 2:
                img = ImageEnhance.Color(img).enhance(self.s)
            else:
                hsv = np.array(img.convert("HSV"), dtype=np.uint8)
                hsv[..., 0] = (hsv[..., 0].astype(np.int16) + int(self.h * 255)) % 256
                img = Image.fromarray(hsv, "HSV").convert("RGB")
        return img


def jitter_row(jitter):
    """a ColorJitter draw as the nine numbers mdx.imgproc.image_prep reads: (enabled, order[4], b, c, s, hue_shift)."""
    if jitter is None:
        return [0.0] * 9
    return [1.0] + [float(v) for v in jitter.order] + [jitter.b, jitter.c, jitter.s, float(int(jitter.h * 255))]


def collate_raw(samples, keep=None):
    """default_collate, with the decoded frames ("raw", f) [h,w,3] of different sizes (KITTI raw has five) padded into one
    [B, hmax, wmax, 3] block per frame; `keep(key)` filters the other entries (processor.step_reads)."""
    from torch.utils.data import default_collate
    raw_keys = [k for k in samples[0] if isinstance(k, tuple) and k[0] == "raw"]
    ragged = [k for k in samples[0] if isinstance(k, tuple) and k[0] in ("depth_idx", "depth_val")]
    rest = default_collate([{k: v for k, v in s.items() if k not in raw_keys and k not in ragged and (keep is None or keep(k))}
                            for s in samples])
    if ragged:   # sparse ground truth: padded to the longest list; padding points one past the last pixel, value 0
        n = max(int(s[("depth_idx", 0)].numel()) for s in samples)
        hw = [int(s["depth_hw"][0]) * int(s["depth_hw"][1]) for s in samples]
        idx = torch.stack([torch.nn.functional.pad(s[("depth_idx", 0)], (0, n - s[("depth_idx", 0)].numel()), value=hw[i])
                           for i, s in enumerate(samples)])
        val = torch.stack([torch.nn.functional.pad(s[("depth_val", 0)], (0, n - s[("depth_val", 0)].numel()))
                           for s in samples])
        rest[("depth_idx", 0)], rest[("depth_val", 0)] = idx, val
    if raw_keys:
        # over EVERY raw key: the frames of a sample share one size today, but a key whose frames differ (another camera) must
        # not be left with uninitialised padding because the first key happened to be uniform
        hmax = max(int(s[k].shape[0]) for s in samples for k in raw_keys)
        wmax = max(int(s[k].shape[1]) for s in samples for k in raw_keys)
        ragged_frames = any(tuple(s[k].shape[:2]) != (hmax, wmax) for s in samples for k in raw_keys)
        for k in raw_keys:
            # 16.8 MB per frame id at batch 12: allocated IN shared memory when this runs in a DataLoader worker (what
            # default_collate does for its stacks) -- a block built in private memory is copied into shared memory when the
            # batch is put on the queue -- and cleared only when frames of different sizes leave padding
            block = _batch_block((len(samples), hmax, wmax, 3), torch.uint8)
            if ragged_frames:
                block.zero_()
            for n, s in enumerate(samples):
                block[n, : s[k].shape[0], : s[k].shape[1]] = s[k]
            rest[k] = block
    return rest


def _batch_block(shape, dtype):
    """An uninitialised tensor for a collated batch entry; in a DataLoader worker its storage lives in shared memory."""
    from torch.utils.data import get_worker_info
    if get_worker_info() is None:
        return torch.empty(shape, dtype=dtype)
    numel = 1
    for v in shape:
        numel *= int(v)
    elem = torch.empty(0, dtype=dtype)
    try:            # what default_collate does for its stacks (private torch API: guarded, the public path below is merely slower)
        storage = elem._typed_storage()._new_shared(numel, device=elem.device)
        return elem.new(storage).resize_(*shape)
    except (AttributeError, TypeError, RuntimeError):
        return torch.zeros(shape, dtype=dtype).share_memory_()


class KITTIDataset(Dataset):
    def __init__(self, datapath, filename, is_training, frame_ids, height=192, width=640, ext=".jpg", scale=4,
                 k_mode="reference_mono", gt_size=(375, 1242), load_depth=True, uint8=False, gpu_prep=False):
        if height % 32 != 0 or width % 32 != 0:
            raise ValueError("(H, W) must be multiples of 32; KITTI sizes are (192, 640) or (320, 1024)")
        self.datapath, self.filename, self.is_training = datapath, list(filename), is_training
        self.frame_ids, self.height, self.width = list(frame_ids), height, width
        self.ext = ext if ext.startswith(".") else "." + ext
        self.scale, self.k_mode, self.gt_size, self.load_depth = scale, k_mode, gt_size, load_depth
        self.uint8 = uint8
        # gpu_prep: the worker only decodes; flip, the Lanczos pyramid, the jitter and ToTensor run on the GPU
        # (mdx.imgproc.image_prep, bit-equal to the Pillow calls below)
        self.gpu_prep = gpu_prep

    def __len__(self):
        return len(self.filename)

    def image_path(self, folder, frame_index, side):
        return os.path.join(self.datapath, folder, "image_0{}/data".format(SIDE_MAP[side]),
                            "{:010d}{}".format(frame_index, self.ext))

    def load_image(self, folder, frame_index, side, do_flip):
        with open(self.image_path(folder, frame_index, side), "rb") as f:
            with Image.open(f) as img:
                image = img.convert("RGB")
        return image.transpose(Image.FLIP_LEFT_RIGHT) if do_flip else image

    def load_point(self, folder, frame_index, side, do_flip):
        calib_path = os.path.join(self.datapath, folder.split("/")[0])
        velo = os.path.join(self.datapath, folder, "velodyne_points/data/{:010d}.bin".format(int(frame_index)))
        depth = resize_nearest(point2depth(calib_path, velo, SIDE_MAP[side]), self.gt_size)
        if do_flip:
            depth = np.fliplr(depth)
        return torch.from_numpy(np.ascontiguousarray(depth[None], dtype=np.float32))

    def intrinsics(self, scale):
        K = K_NORM.copy()
        if self.k_mode == "reference_mono":         # kitti_mono.py:326-327, bug included on purpose
            K[0, :] = K[0, :] * self.width // (2 ** scale)
            K[1, :] = K[1, :] * self.width // (2 ** scale)
        else:                                       # kitti_stereo.py:240-241 (float floor division too)
            K[0, :] = K[0, :] * self.width // (2 ** scale)
            K[1, :] = K[1, :] * self.height // (2 ** scale)
        return torch.from_numpy(K), torch.from_numpy(np.linalg.pinv(K))

    def __getitem__(self, index):
        do_color = self.is_training and random.random() > 0.5
        do_flip = self.is_training and random.random() > 0.5
        line = self.filename[index].split()
        folder, key_frame, side = line[0], int(line[1]), line[2]
        jitter = ColorJitter(random) if do_color else (lambda x: x)
        other = {"r": "l", "l": "r", "2": "3", "3": "2"}[side]
        out = {}
        for frame_id in self.frame_ids:
            if frame_id == "s":
                image = self.load_image(folder, key_frame, other, do_flip and not self.gpu_prep)
            else:
                image = self.load_image(folder, key_frame + frame_id, side, do_flip and not self.gpu_prep)
            if self.gpu_prep:
                out[("raw", frame_id)] = torch.from_numpy(np.array(image, dtype=np.uint8))      # [h, w, 3]
                continue
            for s in range(self.scale):
                small = image.resize((self.width // (2 ** s), self.height // (2 ** s)), Image.LANCZOS)
                out[("color", frame_id, s)] = to_tensor(small, self.uint8)
                out[("color_aug", frame_id, s)] = to_tensor(jitter(small), self.uint8)
        if self.gpu_prep:
            shapes = {tuple(out[("raw", f)].shape) for f in self.frame_ids}
            if len(shapes) != 1:
                raise ValueError("frames of one sample differ in size: %s" % sorted(shapes))
            h, w, _ = shapes.pop()
            out["raw_size"] = torch.tensor([h, w], dtype=torch.int32)
            out["raw_flip"] = torch.tensor(bool(do_flip))
            out["raw_jitter"] = torch.tensor(jitter_row(jitter if do_color else None), dtype=torch.float64)
        if self.load_depth:
            depth = self.load_point(folder, key_frame, side, do_flip)
            if self.gpu_prep:
                # 95 % of a velodyne map is empty: hand over (pixel index, value) pairs -- 0.2 MB instead of 1.9 MB per
                # sample through shared memory, the pinning thread and PCIe; mdx.imgproc.image_prep scatters them
                flat = depth.reshape(-1)
                idx = torch.nonzero(flat).reshape(-1)
                out[("depth_idx", 0)], out[("depth_val", 0)] = idx.to(torch.int32), flat[idx]
                out["depth_hw"] = torch.tensor(depth.shape[-2:], dtype=torch.int32)
            else:
                out[("depth", 0)] = depth
        for s in range(self.scale):
            out[("K", s)], out[("inv_K", s)] = self.intrinsics(s)
        if "s" in self.frame_ids:                    # kitti_stereo.py:249-256
            T = np.eye(4, dtype=np.float32)
            baseline_sign = -1 if do_flip else 1
            side_sign = -1 if SIDE_MAP[side] == 2 else 1
            T[0, 3] = side_sign * baseline_sign * 0.1
            out["stereo"] = torch.from_numpy(T)
        return out


class KITTIMonoDataset_v2(KITTIDataset):
    """reference signature: (datapath, filename, is_training, frame_ids, height, width, ext, scale)."""

    def __init__(self, datapath, filename, is_training, frame_ids, height=192, width=640, ext="jpg", scale=4):
        super().__init__(datapath, filename, is_training, frame_ids, height, width, ext, scale, k_mode="reference_mono")


class KITTIMonoStereoDataset(KITTIDataset):
    def __init__(self, datapath, filename, is_training, frame_ids, height=192, width=640, ext="jpg", scale=4):
        super().__init__(datapath, filename, is_training, frame_ids, height, width, ext, scale, k_mode="scaled")
