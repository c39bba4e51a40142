"""Synthetic stand-in for the KITTI datasets (reference contract: model_loader/kitti_mono.py:258-375,
kitti_stereo.py:168-306): same dictionary keys, shapes and dtypes, no files.  Used by bench.py and the
tests -- the container and the GPU box hold no KITTI data."""
import numpy as np
import torch
from torch.utils.data import Dataset


def make_K(height, width):
    """Normalised KITTI intrinsics scaled to the image (kitti_stereo.py:236-246 form) + pinv."""
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= width
    K[1, :] *= height
    return torch.from_numpy(K), torch.from_numpy(np.linalg.pinv(K).astype(np.float32))


class SyntheticKITTI(Dataset):
    def __init__(self, length, frame_ids, height, width, num_scales=4, seed=0, gt_size=(375, 1242), pool=0, uint8=False,
                 raw=False, is_training=True):
        """pool > 0: only `pool` distinct samples are ever generated (index modulo pool) and they are kept -- the
        generator below costs ~20 ms per sample, far more than decoding a KITTI frame; a throughput measurement of the
        training LOOP must not be a measurement of this stand-in."""
        self.length, self.frame_ids = length, list(frame_ids)
        self.height, self.width, self.num_scales = height, width, num_scales
        self.seed, self.gt_size, self.pool, self._cache = seed, gt_size, pool, {}
        self.uint8 = uint8                    # colours as uint8 (x 255), as model_loader.kitti with uint8=True
        # raw: decoded frames of KITTI's size ([375,1242,3] uint8) + flip / jitter draws, as model_loader.kitti with
        # gpu_prep=True hands them over; mdx.imgproc.image_prep builds the step's entries on the GPU
        self.raw, self.is_training = raw, is_training

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        if self.pool:
            index = index % self.pool
            if index not in self._cache:
                self._cache[index] = self._make(index)
            return dict(self._cache[index])
        return self._make(index)

    def _make(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        inputs = {}
        base = torch.rand(3, self.height // 8, self.width // 8, generator=g)
        base = torch.nn.functional.interpolate(base[None], size=(self.height, self.width), mode="bilinear",
                                               align_corners=False)[0]
        if self.raw:
            import random as _random
            from model_loader.kitti import ColorJitter, jitter_row
            gh, gw = self.gt_size
            big = torch.nn.functional.interpolate(base[None], size=(gh, gw), mode="bilinear", align_corners=False)[0]
            for k, f in enumerate(self.frame_ids):
                img = torch.roll(big, shifts=0 if f == 0 else 6 * k, dims=2) + 0.05 * torch.rand(3, gh, gw, generator=g)
                inputs[("raw", f)] = (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous()
            r = _random.Random(self.seed * 7919 + index)
            do_color = self.is_training and r.random() > 0.5
            do_flip = self.is_training and r.random() > 0.5
            inputs["raw_size"] = torch.tensor([gh, gw], dtype=torch.int32)
            inputs["raw_flip"] = torch.tensor(bool(do_flip))
            inputs["raw_jitter"] = torch.tensor(jitter_row(ColorJitter(r) if do_color else None), dtype=torch.float64)
        for k, f in enumerate(self.frame_ids):
            if self.raw:
                break
            shift = 0 if f == 0 else (3 * k)
            img = torch.roll(base, shifts=shift, dims=2) + 0.05 * torch.rand(3, self.height, self.width, generator=g)
            img = img.clamp(0, 1)
            for s in range(self.num_scales):
                im = img if s == 0 else torch.nn.functional.avg_pool2d(img[None], 2 ** s)[0]
                if self.uint8:
                    im = (im * 255).round().to(torch.uint8)
                inputs[("color", f, s)] = im
                inputs[("color_aug", f, s)] = im
        K, invK = make_K(self.height, self.width)
        for s in range(self.num_scales):
            Ks = K.clone()
            Ks[0, :] /= 2 ** s
            Ks[1, :] /= 2 ** s
            inputs[("K", s)] = Ks
            inputs[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(Ks.numpy()).astype(np.float32))
        if "s" in self.frame_ids:
            T = torch.eye(4)
            T[0, 3] = 0.1
            inputs["stereo"] = T
        gt = torch.zeros(1, *self.gt_size)
        m = torch.rand(1, *self.gt_size, generator=g) < 0.05
        gt[m] = 1 + 79 * torch.rand(int(m.sum()), generator=g)
        if self.raw:           # as model_loader.kitti with gpu_prep: (pixel index, value) pairs
            flat = gt.reshape(-1)
            idx = torch.nonzero(flat).reshape(-1)
            inputs[("depth_idx", 0)], inputs[("depth_val", 0)] = idx.to(torch.int32), flat[idx]
            inputs["depth_hw"] = torch.tensor(self.gt_size, dtype=torch.int32)
        else:
            inputs[("depth", 0)] = gt
        return inputs
