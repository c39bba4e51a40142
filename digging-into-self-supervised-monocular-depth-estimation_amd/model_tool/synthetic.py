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


class scene(object):
    """A small textured world with KNOWN geometry, rendered by ray casting: a bumpy ground plane, two side walls and a bumpy
    back wall around the camera track (a road corridor), each carrying a procedural texture that is a function of the 3-D
    point.  Any camera placed in it sees images that are exactly consistent with one rigid scene -- what the photometric
    loss assumes and what `torch.roll` frames are not.  Units: the stereo baseline is 0.1 (reference kitti_stereo.py:249-256),
    so the camera is ~0.3 above the ground and depths run from ~1 to ~25.  Plain torch on the CPU; nothing from oracle/."""

    def __init__(self, g):
        r = lambda lo, hi: float(lo + (hi - lo) * torch.rand((), generator=g))     # noqa: E731
        self.cam_h = r(0.25, 0.4)                          # camera height above the ground (y points down)
        self.wall_l, self.wall_r = -r(1.2, 2.5), r(1.2, 2.5)
        self.back = r(14.0, 24.0)
        self.tilt = r(-0.04, 0.04)                         # ground slope along z
        self.bump = (r(0.02, 0.05), r(0.6, 1.4), r(0.5, 1.2), r(0, 6.28), r(0, 6.28))   # amplitude, two frequencies, two phases
        # texture: per surface and colour channel, a sum of oriented sinusoids over two surface coordinates
        self.tex = {}
        for name in ("ground", "left", "right", "back"):
            waves = []
            for lam in (0.35, 0.8, 1.7, 4.0):              # wavelengths: several pixels wide at every depth of the corridor
                th = r(0, 3.1416)
                waves.append((2 * 3.14159265 / lam * float(np.cos(th)), 2 * 3.14159265 / lam * float(np.sin(th)),
                              [r(0, 6.28) for _ in range(3)], [r(0.05, 0.14) for _ in range(3)]))
            self.tex[name] = ([r(0.35, 0.65) for _ in range(3)], waves)

    def _texture(self, name, u, v):
        base, waves = self.tex[name]
        out = []
        for c in range(3):
            t = torch.full_like(u, base[c])
            for (ku, kv, ph, amp) in waves:
                t = t + amp[c] * torch.sin(ku * u + kv * v + ph[c])
            out.append(t)
        return torch.stack(out)

    def render(self, K, T, height, width, want_color=True):
        """Image [3,H,W] and depth [H,W] (z in the camera's own frame) of the camera whose points are X_cam = T X_world
        (T [4,4], the reference's convention for a target -> source transform with the TARGET camera as the world), with
        intrinsics K [4,4]; pixel (row i, column j) looks along K^-1 (j, i, 1) -- the reference's pixel grid
        (model_layer/warp.py:193-246) and grid_sample's align_corners=True convention."""
        K = K.double()
        T = T.double()
        R, t = T[:3, :3], T[:3, 3]
        cen = -(R.T @ t)                                   # camera centre in the world
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float64), torch.arange(width, dtype=torch.float64), indexing="ij")
        pix = torch.stack((xs, ys, torch.ones_like(xs)), 0).reshape(3, -1)
        d_cam = torch.linalg.inv(K[:3, :3]) @ pix          # z component 1: the ray parameter is the camera-frame depth
        d = R.T @ d_cam
        amp, f1, f2, p1, p2 = self.bump
        best = torch.full((height * width,), float("inf"), dtype=torch.float64)
        color = torch.zeros(3, height * width, dtype=torch.float64)

        def hit(name, tpar, ok, u, v):
            nonlocal best, color
            ok = ok & (tpar > 1e-3) & (tpar < best)
            if ok.any():
                best = torch.where(ok, tpar, best)
                if want_color:
                    color = torch.where(ok[None], self._texture(name, u, v), color)

        # ground y = cam_h + tilt * z + bumps(x, z): a few fixed-point steps from the plane's own intersection
        tg = (self.cam_h - cen[1]) / d[1].clamp(min=1e-9)
        for _ in range(4):
            X = cen[:, None] + tg * d
            tg = (self.cam_h + self.tilt * X[2] + amp * torch.sin(f1 * X[0] + p1) * torch.sin(f2 * X[2] + p2) - cen[1]) / d[1].clamp(min=1e-9)
        X = cen[:, None] + tg * d
        hit("ground", tg, d[1] > 1e-6, X[0], X[2])
        for name, xw in (("left", self.wall_l), ("right", self.wall_r)):
            tw = (xw - cen[0]) / torch.where(d[0].abs() < 1e-9, torch.full_like(d[0], 1e-9), d[0])
            X = cen[:, None] + tw * d
            hit(name, tw, torch.ones_like(tw, dtype=torch.bool), X[2], X[1])
        tb = (self.back - cen[2]) / d[2].clamp(min=1e-9)
        for _ in range(4):
            X = cen[:, None] + tb * d
            tb = (self.back + 6 * amp * torch.sin(f2 * X[0] + p2) * torch.sin(f1 * X[1] + p1) - cen[2]) / d[2].clamp(min=1e-9)
        X = cen[:, None] + tb * d
        hit("back", tb, d[2] > 1e-6, X[0], X[1])
        depth = torch.where(torch.isfinite(best), best, torch.full_like(best, 100.0))
        return color.reshape(3, height, width).clamp(0, 1).float(), depth.reshape(height, width).float()

    @staticmethod
    def motion(g, frame_id):
        """Target -> source transform of a temporal neighbour: mostly along the optical axis (a car), a little sideways, a
        small yaw / pitch; frame -1 lies behind the target, +1 ahead.  "s": the stereo partner (x shifted by the baseline)."""
        T = torch.eye(4, dtype=torch.float64)
        if frame_id == "s":
            T[0, 3] = 0.1
            return T
        r = lambda lo, hi: float(lo + (hi - lo) * torch.rand((), generator=g))     # noqa: E731
        step = r(0.12, 0.22) * (1 if frame_id > 0 else -1)
        yaw, pitch = r(-0.02, 0.02), r(-0.006, 0.006)
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
        Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=torch.float64)
        T[:3, :3] = Ry @ Rx
        # the source camera sits `step` ahead of the target along z (and a little aside): X_src = R (X - c)
        c = torch.tensor([r(-0.02, 0.02), r(-0.005, 0.005), step], dtype=torch.float64)
        T[:3, 3] = -(T[:3, :3] @ c)
        return T


class SyntheticKITTI(Dataset):
    def __init__(self, length, frame_ids, height, width, num_scales=4, seed=0, gt_size=(375, 1242), pool=0, uint8=False,
                 raw=False, is_training=True, geometry=False):
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
        # geometry: frames rendered from ONE rigid textured scene at known poses, ground truth = that scene's depth
        # (class scene above) -- for tests of what the step LEARNS; the default frames (a shifted copy of the target, random
        # ground truth) cost less to make and serve the throughput measurements
        self.geometry = geometry
        if geometry and raw:
            raise ValueError("SyntheticKITTI: geometry=True renders the step's entries directly (raw=False)")

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
        world = scene(g) if self.geometry else None
        K0 = make_K(self.height, self.width)[0]
        for k, f in enumerate(self.frame_ids):
            if self.raw:
                break
            if self.geometry:
                T = torch.eye(4, dtype=torch.float64) if f == 0 else scene.motion(g, f)
                img, _ = world.render(K0, T, self.height, self.width)
                img = (img + 0.01 * torch.randn(3, self.height, self.width, generator=g)).clamp(0, 1)
                inputs[("pose_gt", f)] = T.float()           # (not read by the step: what a test may compare poses with)
            else:
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
        if self.geometry:       # the scene's depth seen through the same camera at the ground truth's size, 5 % of the pixels
            Kg = make_K(*self.gt_size)[0]
            dense = world.render(Kg, torch.eye(4, dtype=torch.float64), *self.gt_size, want_color=False)[1]
            gt[m] = dense[None][m]
            inputs[("depth_dense", 0)] = torch.nn.functional.interpolate(dense[None, None], size=(self.height, self.width),
                                                                         mode="nearest")[0]
        else:
            gt[m] = 1 + 79 * torch.rand(int(m.sum()), generator=g)
        if self.raw:           # as model_loader.kitti with gpu_prep: (pixel index, value) pairs
            flat = gt.reshape(-1)
            idx = torch.nonzero(flat).reshape(-1)
            inputs[("depth_idx", 0)], inputs[("depth_val", 0)] = idx.to(torch.int32), flat[idx]
            inputs["depth_hw"] = torch.tensor(self.gt_size, dtype=torch.int32)
        else:
            inputs[("depth", 0)] = gt
        return inputs
