"""`compute`: the step driver (reference: model_tool/processor.py:16-218), same method names.

    forward_depth   processor.py:33-55      forward_pose   processor.py:58-136
    image2warping   processor.py:139-163    compute_loss   processor.py:166-218

With opt.fused (default) image2warping only forms the projection matrices.  compute_loss then runs, when gradients
are wanted (training), ONE gfx950 kernel for all scales that evaluates the photometric term AND its gradient
(opt.fused_train, default: mdx.functional.photometric_train, csrc/photo_train.hip); without gradients
(validation) or with opt.fused_train = False one fused forward kernel per scale (+ one backward kernel per scale).
The identity losses, which do not depend on the scale, are evaluated once per step.  With opt.fused = False the
reference's op-by-op sequence runs on the fine-grained kernels and every reference output key is populated
(("warp_color", f, s), ...).  Results are identical between the modes.
"""
import os

import torch

from model_layer import *  # noqa: F401,F403  (the reference does the same star-import, processor.py:11)
from model_loss import *   # noqa: F401,F403
from mdx import functional as F


def _opt(opt, name, default):
    return getattr(opt, name, default)


def step_reads(key):
    """The entries of the data layer's dictionary a step consumes.  The reference collates and copies
    every key to the device (processor.py:34-35); the colour_aug pyramids, the source frames' pyramids and the
    intrinsics of scales 1-3 are never read: they are neither uploaded (forward_depth) nor, with
    opt.collate_step_keys, stacked into the batch by the DataLoader workers."""
    if not isinstance(key, tuple):
        return True                                         # "stereo"
    if key[0] == "color":
        return key[2] == 0 or key[1] == 0                   # warp sources / identity / target; target pyramid (smoothness)
    if key[0] == "color_aug":
        return key[2] == 0                                  # network inputs
    if key[0] in ("K", "inv_K"):
        return key[1] == 0
    return True                                             # ("depth", 0), injected test entries


HOST_KEYS = ("raw_size", "raw_flip", "raw_jitter", "depth_hw")     # read on the host by mdx.imgproc.image_prep: never uploaded


def device_key(key):
    return step_reads(key) and key not in HOST_KEYS


def _pose_head_output(axisangle, translation):
    """The [M, F, 1, 6] tensor the pose head returned as (x[..., :3], x[..., 3:]) (pose_decoder.py:53-56), or None when the two
    are not those views of one dense tensor."""
    base = axisangle._base
    if base is None or base is not translation._base or not base.is_contiguous() or axisangle.dim() != 4:
        return None
    M, Fr = axisangle.shape[0], axisangle.shape[1]
    want = (Fr * 6, 6, 6, 1)
    if (base.numel() != M * Fr * 6 or tuple(axisangle.shape) != (M, Fr, 1, 3) or tuple(translation.shape) != (M, Fr, 1, 3)
            or axisangle.stride() != want or translation.stride() != want
            or axisangle.storage_offset() != base.storage_offset() or translation.storage_offset() != base.storage_offset() + 3):
        return None
    return base.view(M, Fr, 1, 6)


class compute(object):
    def __init__(self, opt, device):
        self.opt = opt
        self.device = device
        self.num_pose_frames = len(opt.frame_ids) if opt.pose_frames == "all" else 2
        self.fused = _opt(opt, "fused", True)
        self.fused_train = _opt(opt, "fused_train", True)
        if not str(device).startswith("cuda"):
            # the reference's device pick on a machine without a GPU (model_train.py:28; BASELINE configs[0]): the fused entries are
            # GPU kernels, so the CPU run takes the reference-shaped op-by-op path, whose ops dispatch CPU tensors to the package's
            # plain-PyTorch restatement (mdx/composite.py)
            self.fused = False
        # "device": N(0,1) drawn on the GPU; "cpu": the reference's torch.randn on the host + H2D copy
        # (processor.py:195) -- same stream of numbers as the reference for a given torch seed.
        self.noise_mode = _opt(opt, "noise", "device")
        self.amp = _opt(opt, "amp", "none")
        # both frame pairs through the separate pose network in one batch (same numbers, see below)
        self.batch_pose_pairs = _opt(opt, "batch_pose_pairs", True)
        self._prep = None
        self._rng = None          # device {seed, offset} of the in-kernel noise (created at the first step)
        # False: round 2's form (identity kernel + torch.randn + the training kernel re-deriving the target statistics
        # per scale) -- kept for A/B measurements (bench.py --no-prologue)
        self.prologue = _opt(opt, "prologue", True)
        # False: the step's small ops as the reference spells them -- the scalar tail of the loss as ~5 torch ops per scale, the
        # pose head's output sliced per frame into param2matrix + K @ T (A/B and parity tests; same numbers)
        self.fused_tail = self.fused and _opt(opt, "fused_tail", True)

    # -- networks ---------------------------------------------------------------------------------
    def _autocast(self):
        enabled = self.amp == "bf16" and str(self.device).startswith("cuda")
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=enabled)

    def _step_reads(self, key):
        return device_key(key)

    def prepare(self, inputs):
        """Batches that carry decoded frames (("raw", f), model_loader.kitti with gpu_prep): flip, Lanczos pyramid, colour
        jitter and ToTensor on the GPU, on the current stream -> the entries the step reads.  Other batches pass."""
        from mdx.imgproc import image_prep
        if not image_prep.wanted(inputs):
            return inputs
        if self._prep is None:
            self._prep = image_prep(self.opt.height, self.opt.width, self.opt.frame_ids, len(self.opt.scales), self.device)
        # IN PLACE, as the reference's forward_depth treats its dictionary (processor.py:34-35): the caller's batch is the
        # one control.metric reads ("depth", 0) from afterwards
        prepared = self._prep(inputs)
        inputs.clear()
        inputs.update(prepared)
        return inputs

    def stage_inputs(self, inputs):
        """What forward_depth does to the batch before the networks run (decoded frames -> step entries, upload, uint8 -> float32):
        idempotent, so a caller that wants the pose network to start beside the depth network calls it first."""
        dev = torch.device(self.device)
        inputs = self.prepare(inputs)
        for key in inputs:
            if torch.is_tensor(inputs[key]) and inputs[key].device != dev and self._step_reads(key):
                inputs[key] = inputs[key].to(self.device, non_blocking=True)
            # uint8 colours (opt.uint8_loader): ToTensor's x / 255 happens here, on the device, with the IEEE divide
            # (mdx.imgproc.to_tensor; torch's `x / 255.0` on the GPU multiplies by the reciprocal: other numbers)
            if (torch.is_tensor(inputs[key]) and inputs[key].dtype == torch.uint8 and isinstance(key, tuple)
                    and key[0] in ("color", "color_aug") and inputs[key].device == dev):
                if inputs[key].is_cuda:
                    from mdx.imgproc import to_tensor
                    inputs[key] = to_tensor(inputs[key])
                else:
                    inputs[key] = torch.true_divide(inputs[key], 255.0)      # ATen's CPU kernel divides
        return inputs

    def forward_depth(self, inputs, outputs, setting):
        inputs = self.stage_inputs(inputs)
        with self._autocast():
            if self.opt.pose_type == "shared":
                all_frames = torch.cat([inputs[("color_aug", f, 0)] for f in self.opt.frame_ids])
                feats = setting.model["encoder"](all_frames)
                feats = [torch.split(f, self.opt.batch) for f in feats]
                outputs["features"] = feats
                for index, frame_id in enumerate(self.opt.frame_ids):
                    outputs.update({frame_id: [f[index] for f in feats]})
                outputs.update(setting.model["decoder"](outputs[0]))
            else:  # "separate" and "posecnn": one image through the depth network
                outputs["features"] = setting.model["encoder"](inputs[("color_aug", 0, 0)])
                outputs.update(setting.model["decoder"](outputs["features"]))
        return inputs, outputs

    def forward_pose(self, inputs, outputs, setting):
        opt = self.opt
        with self._autocast():
            if self.num_pose_frames == 2 and opt.pose_type == "separate" and self.batch_pose_pairs:
                self._forward_pose_pairs_batched(inputs, outputs, setting)
            elif self.num_pose_frames == 2:
                for frame_id in opt.frame_ids[1:]:
                    if frame_id == "s":
                        continue
                    first, second = (frame_id, 0) if frame_id < 0 else (0, frame_id)
                    if opt.pose_type == "shared":
                        pose_inputs = [outputs[first], outputs[second]]
                    else:
                        pose_inputs = torch.cat([inputs[("color_aug", first, 0)], inputs[("color_aug", second, 0)]], 1)
                        if opt.pose_type == "separate":
                            pose_inputs = [setting.model["pose_encoder"](pose_inputs)]
                    axisangle, translation = setting.model["pose_decoder"](pose_inputs)
                    outputs[("R", frame_id, 0)] = axisangle
                    outputs[("T", frame_id, 0)] = translation
                    outputs[("c2c", frame_id, 0)] = param2matrix(
                        axisangle=axisangle[:, 0].float(), translation=translation[:, 0].float(),
                        invert=(frame_id < 0))
            else:
                frames = [f for f in opt.frame_ids if f != "s"]
                if opt.pose_type == "shared":
                    axisangle, translation = setting.model["pose_decoder"]([outputs[f] for f in frames])
                else:
                    all_frames = torch.cat([inputs[("color_aug", f, 0)] for f in frames], 1)
                    if opt.pose_type == "separate":
                        all_frames = [setting.model["pose_encoder"](all_frames)]
                    axisangle, translation = setting.model["pose_decoder"](all_frames)
                for index, frame_id in enumerate(opt.frame_ids[1:]):
                    if frame_id != "s":
                        outputs[("R", frame_id, 0)] = axisangle
                        outputs[("T", frame_id, 0)] = translation
                        outputs[("c2c", frame_id, 0)] = param2matrix(
                            axisangle=axisangle[:, index].float(), translation=translation[:, index].float())
        return inputs, outputs

    def _forward_pose_pairs_batched(self, inputs, outputs, setting):
        """The frame pairs of processor.py:61-83 (one pose-network call per source frame) as ONE pass of the
        convolutions over the concatenated pairs.  The pose encoder's batch norms normalise each pair with its own
        statistics and update the running statistics pair by pair (BatchNorm2d.batch_groups), so every number is
        the one the per-pair loop produces; the parameter gradients of the two uses are summed inside the
        weight-gradient kernels instead of by ~200 small accumulation kernels."""
        from model_layer.depth_encoder import BatchNorm2d
        opt = self.opt
        frames = [f for f in opt.frame_ids[1:] if f != "s"]
        pairs = []
        for frame_id in frames:
            first, second = (frame_id, 0) if frame_id < 0 else (0, frame_id)
            pairs.append([inputs[("color_aug", first, 0)], inputs[("color_aug", second, 0)]])
        n = pairs[0][0].shape[0]
        # the pairs as they are: the package's encoder normalises and lays them out channels-last in one pass (F.FrameStack);
        # any other module gets the concatenation
        stack = F.FrameStack(pairs)
        encoder = setting.model["pose_encoder"]
        takes_stack = self.fused_tail and hasattr(getattr(encoder, "module", encoder), "_normalised_input") and stack.ok()
        with BatchNorm2d.batch_groups(len(pairs), encoder):
            feats = encoder(stack if takes_stack else stack.tensor())
        axisangle, translation = setting.model["pose_decoder"]([feats])
        for k, frame_id in enumerate(frames):
            outputs[("R", frame_id, 0)] = axisangle[k * n:(k + 1) * n]
            outputs[("T", frame_id, 0)] = translation[k * n:(k + 1) * n]
        raw = _pose_head_output(axisangle, translation)
        K = inputs.get(("K", 0))
        if (self.fused_tail and raw is not None and raw.is_cuda
                and frames == list(opt.frame_ids[1:]) and torch.is_tensor(K) and K.is_cuda and not K.requires_grad):
            # every source frame's matrix AND projection from the pose head's output in one launch (one in backward): the
            # reference's row slice + [:, 0] + param2matrix + K @ T per frame is ~10 launches forward and ~30 backward
            T, P = F.pose_projection(raw.float(), K, [k * n for k in range(len(frames))], [0] * len(frames),
                                     [int(f < 0) for f in frames])
            for k, frame_id in enumerate(frames):
                outputs[("c2c", frame_id, 0)] = T[k]
            outputs[("P", "pose_head")] = P
            return
        for k, frame_id in enumerate(frames):
            aa, tr = outputs[("R", frame_id, 0)], outputs[("T", frame_id, 0)]
            outputs[("c2c", frame_id, 0)] = param2matrix(axisangle=aa[:, 0].float(), translation=tr[:, 0].float(),
                                                         invert=(frame_id < 0))

    # -- geometry ---------------------------------------------------------------------------------
    def _transformation(self, inputs, outputs, frame_id, depth_fn):
        opt = self.opt
        if frame_id == "s":
            return inputs["stereo"]
        if opt.pose_type in ["shared", "separate"]:
            return outputs[("c2c", frame_id, 0)]
        # posecnn (processor.py:153-157): translation scaled by the mean inverse depth
        axisangle, translation = outputs[("R", frame_id, 0)], outputs[("T", frame_id, 0)]
        depth = depth_fn()
        mean_inv_depth = (1 / depth).mean(3, True).mean(2, True)
        return param2matrix(axisangle[:, 0].float(), translation[:, 0].float() * mean_inv_depth[:, 0], (frame_id < 0))

    def _depth(self, outputs, scale):
        disp = interpolate(outputs[("disp", scale)].float(), self.opt.height, self.opt.width, "bilinear", False)
        return disparity2depth(disp, self.opt.min_depth, self.opt.max_depth)[1]

    def image2warping(self, inputs, outputs, setting):
        opt = self.opt
        K = inputs[("K", 0)]
        if self.fused and ("P", "pose_head") in outputs:      # forward_pose formed the projections with the matrices
            P = outputs.pop(("P", "pose_head"))
            for s in opt.scales:
                outputs[("P", s)] = P
            return inputs, outputs
        for scale in opt.scales:
            depth = None
            if not self.fused:
                depth = self._depth(outputs, scale)
                outputs[("depth", 0, scale)] = depth
            Ps = []
            for frame_id in opt.frame_ids[1:]:
                T = self._transformation(inputs, outputs, frame_id, lambda: self._depth(outputs, scale))
                if self.fused:
                    Ps.append(F.compose_projection(K, T))
                else:
                    cam = setting.inv_projection[0](depth, inputs[("inv_K", 0)])
                    grid = setting.for_projection[0](cam, K, T)
                    outputs[("warp_color", frame_id, scale)] = grid_sample(
                        inputs[("color", frame_id, 0)], grid, "border", True)
            if self.fused:
                outputs[("P", scale)] = torch.stack(Ps)
                if opt.pose_type != "posecnn":   # P does not depend on the scale: share it
                    for s in opt.scales:
                        outputs[("P", s)] = outputs[("P", scale)]
                    break
        return inputs, outputs

    # -- loss -------------------------------------------------------------------------------------
    def draws_in_kernel(self):
        """Does the step draw its auto-mask noise inside the prologue kernel (device {seed, offset} state)?"""
        return (self.fused and self.fused_train and self.prologue and bool(self.opt.use_automasking)
                and self.noise_mode != "cpu" and len(self.opt.scales) <= 4)

    def noise_rng(self, device=None):
        """The device-resident {seed, offset} of the in-kernel generator (created at first use; the offset counts the
        steps taken: the finishing kernel of every step advances it).  Part of the run's state: control.save stores the
        offset, control.resume puts it back (set_noise_offset), graphed_step's warm-up restores it."""
        if self._rng is None:
            import os
            self._rng = F.noise_state(device or self.device, stream=int(_opt(self.opt, "noise_stream", os.environ.get("RANK", "0"))))
            if getattr(self, "_rng_offset0", None) is not None:
                self._rng.tensor[1] = int(self._rng_offset0)
        return self._rng

    def noise_offset(self):
        """Steps the in-kernel generator has served (None: it has not been used)."""
        return None if self._rng is None else int(self._rng.tensor[1])

    def set_noise_offset(self, value):
        self._rng_offset0 = None if value is None else int(value)
        if self._rng is not None and value is not None:
            self._rng.tensor[1] = int(value)

    def _noise(self, shape):
        if self.noise_mode == "cpu":
            return torch.randn(shape).to(self.device)
        return torch.randn(shape, device=self.device)

    def loss_prologue(self, inputs, outputs):
        """The launches of compute_loss that need the batch and the disparities only -- the prologue kernel (identity losses, noise,
        the target's window statistics) and the smoothness passes -- issued NOW and kept for compute_loss.  A caller that runs the
        pose network beside the depth network calls this on the depth network's stream before it joins the pose stream: 66 us of
        kernels leave the stretch of the step where nothing else can run.  Does nothing unless the step takes the default path
        (one-launch training kernel, fused tail, in-kernel noise)."""
        opt = self.opt
        target = inputs[("color", 0, 0)]
        if not (self.fused and self.fused_train and self.fused_tail and self.prologue and len(opt.scales) <= 4 and target.is_cuda
                and torch.is_grad_enabled() and all(("disp", s) in outputs for s in opt.scales)):
            return
        automask = bool(opt.use_automasking)
        if automask and (self.noise_mode == "cpu" or any(("noise", s) in inputs for s in opt.scales)):
            return
        sources = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
        if automask:
            self.noise_rng(target.device)
        pre = F.photometric_prologue(target, sources, len(opt.scales), noises=None, rng=self._rng, automask=automask)
        disps = [outputs[("disp", s)].float() for s in opt.scales]
        smooth = F.smooth_launch(disps, [inputs[("color", 0, s)] for s in opt.scales], need_grad=any(d.requires_grad for d in disps))
        outputs[("loss_prologue",)] = dict(pre=pre, smooth=smooth)

    def compute_loss(self, inputs, outputs, setting):
        opt = self.opt
        target = inputs[("color", 0, 0)]
        sources = [inputs[("color", f, 0)] for f in opt.frame_ids[1:]]
        S = len(sources)
        B, _, H, W = target.shape
        automask = bool(opt.use_automasking)
        total_loss = 0
        ident = None
        one_launch = self.fused and self.fused_train and len(opt.scales) <= 4
        if self.fused and automask and not one_launch:
            ident = F.identity_loss(target, sources)          # once per step (scale-independent)
        # training: every scale's photometric term and its gradient in one launch (posecnn: one projection per scale);
        # validation / torch.no_grad(): the same launch in its forward-only form.  What the scales share -- identity
        # losses + noise + their minimum, the target's window statistics -- comes from ONE prologue launch
        # (csrc/photo_prologue.hip); the noise is drawn inside it (the reference's host-side torch.randn with --noise cpu,
        # or tensors injected by the parity tests, are handed to it instead)
        train = None
        smooth_all = None
        if one_launch:
            nsc = len(opt.scales)
            noises = None
            if automask:
                if all(("noise", s) in inputs for s in opt.scales):     # injected (parity tests)
                    noises = [inputs[("noise", s)] for s in opt.scales]
                elif self.noise_mode == "cpu":
                    noises = list(self._noise((nsc, B, S, H, W)).unbind(0))
                elif self.prologue:
                    self.noise_rng(target.device)
            pre = None
            early = outputs.pop(("loss_prologue",), None)          # loss_prologue() ran ahead of the pose stream's join
            if early is not None:
                pre = early["pre"]
            elif self.prologue:
                pre = F.photometric_prologue(target, sources, nsc, noises=noises, rng=self._rng, automask=automask)
            elif automask:
                ident = F.identity_loss(target, sources)
                if noises is None:
                    noises = list(torch.randn((nsc, B, S, H, W), device=self.device).unbind(0))
            # The smoothness launches go BETWEEN the prologue and the training kernel (they depend on neither).  Launched right
            # behind the prologue -- which writes 80 MB in 44 us -- the training kernel runs 17 % longer (223 us against 190 us,
            # same code, same data: profiles/r04_load_latency.txt); two short launches in between and it does not.
            if self.fused_tail and target.is_cuda:
                # smoothness, photometric term and the scalar tail of processor.py:208-217 as one autograd node: one launch
                # finishes the scalar, one launch in backward writes every scale's disparity gradient (mdx/functional.py: _TrainLoss)
                res = F.train_loss([outputs[("disp", s)].float() for s in opt.scales],
                                   (outputs[("P", opt.scales[0])] if opt.pose_type != "posecnn"
                                    else [outputs[("P", s)] for s in opt.scales]),
                                   target, sources, inputs[("inv_K", 0)], [inputs[("color", 0, s)] for s in opt.scales],
                                   opt.scales, opt.disp_smoothness, ident, noises if pre is None else None,
                                   automask=automask, min_depth=opt.min_depth, max_depth=opt.max_depth,
                                   need_depth=(opt.scales[0] == 0), pre=pre, smooth=early["smooth"] if early is not None else None)
                if res["depth"] is not None:
                    outputs[("depth", 0, 0)] = res["depth"]
                for k, scale in enumerate(opt.scales):
                    outputs[("automask", scale)] = res["idx"][k]
                outputs["loss"] = res["loss"]
                return outputs
            if self.fused and target.is_cuda:
                smooth_all = F.smooth_loss_multi([outputs[("disp", s)].float() for s in opt.scales],
                                                 [inputs[("color", 0, s)] for s in opt.scales])
            train = F.photometric_train([outputs[("disp", s)].float() for s in opt.scales],
                                        (outputs[("P", opt.scales[0])] if opt.pose_type != "posecnn"
                                         else [outputs[("P", s)] for s in opt.scales]),
                                        target, sources, inputs[("inv_K", 0)], ident, noises if pre is None else None,
                                        automask=automask, min_depth=opt.min_depth, max_depth=opt.max_depth,
                                        need_depth=(opt.scales[0] == 0), pre=pre)
            if train["depth"] is not None:
                outputs[("depth", 0, 0)] = train["depth"]
        # fused mode: the smoothness term of every scale with each of its passes launched once (2 launches, not 16)
        if smooth_all is None and self.fused and len(opt.scales) <= 4 and target.is_cuda:
            smooth_all = F.smooth_loss_multi([outputs[("disp", s)].float() for s in opt.scales],
                                             [inputs[("color", 0, s)] for s in opt.scales])
        for k, scale in enumerate(opt.scales):
            disp = outputs[("disp", scale)].float()
            color = inputs[("color", 0, scale)]
            if train is not None:
                mean_min = train["sums"][k] / float(B * H * W)
                outputs[("automask", scale)] = train["idx"][k]
            elif self.fused:
                noise = self._noise((B, S, H, W)) if automask else None
                if ("noise", scale) in inputs:                 # injected (parity tests)
                    noise = inputs[("noise", scale)]
                res = F.photometric_scale(disp, outputs[("P", scale)], target, sources, inputs[("inv_K", 0)],
                                          ident, noise, automask=automask, min_depth=opt.min_depth,
                                          max_depth=opt.max_depth, need_depth=(scale == 0))
                mean_min = res["sum"][0] / float(B * H * W)
                outputs[("automask", scale)] = res["idx"]
                if res["depth"] is not None:
                    outputs[("depth", 0, scale)] = res["depth"]
            else:
                reprojection_loss = torch.cat([setting.loss["reprojection"](
                    outputs[("warp_color", f, scale)], target) for f in opt.frame_ids[1:]], 1)
                if automask:
                    identity_loss = torch.cat([setting.loss["reprojection"](s, target) for s in sources], 1)
                    noise = inputs[("noise", scale)] if ("noise", scale) in inputs else self._noise(identity_loss.shape)
                    identity_loss = identity_loss + 0.00001 * noise
                    combined_loss = torch.cat((identity_loss, reprojection_loss), dim=1)
                else:
                    combined_loss = reprojection_loss
                if combined_loss.shape[1] == 1:
                    to_optimise = combined_loss
                else:
                    to_optimise, idxs = torch.min(combined_loss, dim=1)
                    outputs[("automask", scale)] = idxs
                mean_min = to_optimise.mean()
            smooth_loss = smooth_all[k] if smooth_all is not None else setting.loss["edge_aware"](disp=disp, color=color)
            scale_loss = mean_min + opt.disp_smoothness * smooth_loss / (2 ** scale)
            total_loss = total_loss + scale_loss
        total_loss = total_loss / len(opt.scales)
        outputs["loss"] = total_loss
        return outputs
