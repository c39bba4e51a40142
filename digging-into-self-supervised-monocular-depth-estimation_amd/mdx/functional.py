"""Autograd functions over the libmdx_hip.so C-ABI.

Fused path (what the training step runs):
    compose_projection(K, T)                   -> P = (K@T)[:, :3]      (warp.py:260)
    identity_loss(target, sources)             -> [B,S,H,W]             (processor.py:187-191)
    photometric_scale(disp, P, target, ...)    -> sum(to_optimise), idx (processor.py:141-162,172-204)
    smooth_loss(disp, color)                   -> scalar                (model_loss.py:107-116)
Fine-grained ops mirror model_layer/warp.py and model_loss/model_loss.py one to one.
All tensors must be CUDA/HIP float32; there is no CPU path.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 8) // 8 + 1, dtype=torch.float64, device=device)


# ------------------------------------------------------------------------------------------------
# fused path
# ------------------------------------------------------------------------------------------------
class _ComposeProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, T):
        K, T = _f32c(K), _f32c(T)
        B = K.shape[0]
        P = torch.empty(B, 3, 4, device=K.device, dtype=torch.float32)
        check(lib().mdx_compose_projection(ptr(K), ptr(T), B, ptr(P), stream()), "mdx_compose_projection")
        ctx.save_for_backward(K, T)
        return P

    @staticmethod
    def backward(ctx, gP):
        K, T = ctx.saved_tensors
        gK = gT = None
        if ctx.needs_input_grad[1]:
            gT = torch.matmul(K[:, :3, :].transpose(1, 2), gP)
        if ctx.needs_input_grad[0]:
            gK = torch.zeros_like(K)
            gK[:, :3, :] = torch.matmul(gP, T.transpose(1, 2))
        return gK, gT


def compose_projection(K, T):
    """(K @ T)[:, :3, :] with ATen-CPU's rounding order.  K, T: [B,4,4] -> [B,3,4]."""
    return _ComposeProjection.apply(K, T)


def identity_loss(target, sources, desc=None):
    """ReprojectionLoss(color_f, target) for every source frame -> [B,S,H,W] (no grad: inputs only)."""
    target = _f32c(target)
    sources = [_f32c(s) for s in sources]
    B, _, H, W = target.shape
    S = len(sources)
    d = desc or _lib.make_desc(B, H, W, H, W, S, True, 0.1, 100.0)
    src = _lib.make_sources(sources)
    out = torch.empty(B, S, H, W, device=target.device, dtype=torch.float32)
    check(lib().mdx_identity_loss(C.byref(d), ptr(target), C.byref(src), ptr(out), stream()), "mdx_identity_loss")
    return out


# Measurement aid: when a dict {"fwd": [], "bwd": []}, every fused per-scale launch goes through the *_timed entry
# points and appends its (start, stop) hipEvent pair, recorded right around the fused kernel on the launch stream;
# timing_summary() turns them into mean microseconds.  Same kernels, same order, same results as the plain path.
TIMING = None


def _timing_hook(kind):
    if TIMING is None:
        return None
    t = _lib.Timing(lib().mdx_event_create(), lib().mdx_event_create())
    TIMING.setdefault(kind, []).append(t)
    return t


def timing_summary(timing):
    """{kind: (mean_us, n)} of the recorded pairs; destroys the events."""
    out = {}
    for kind, pairs in timing.items():
        us = []
        for t in pairs:
            v = C.c_float()
            check(lib().mdx_event_elapsed_us(C.c_void_p(t.start), C.c_void_p(t.stop), C.byref(v)), "mdx_event_elapsed_us")
            us.append(v.value)
            lib().mdx_event_destroy(C.c_void_p(t.start))
            lib().mdx_event_destroy(C.c_void_p(t.stop))
        out[kind] = (sum(us) / max(len(us), 1), len(us))
    return out


class _PhotometricScale(torch.autograd.Function):
    """One scale: returns (sum over pixels of to_optimise [1], idx uint8 [B,H,W], extras...)."""

    @staticmethod
    def forward(ctx, disp, P, target, invK, ident, noise, cfg, *sources):
        disp, P, target, invK = _f32c(disp), _f32c(P), _f32c(target), _f32c(invK)
        sources = [_f32c(s) for s in sources]
        B, _, h, w = disp.shape
        _, _, H, W = target.shape
        S = len(sources)
        automask = cfg["automask"]
        d = _lib.make_desc(B, H, W, h, w, S, automask, cfg["min_depth"], cfg["max_depth"])
        src = _lib.make_sources(sources)
        dev = disp.device
        if automask:
            ident, noise = _f32c(ident), _f32c(noise)
        idx = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        loss_sum = torch.empty(1, device=dev, dtype=torch.float32)
        to_opt = torch.empty(B, H, W, device=dev, dtype=torch.float32) if cfg.get("need_to_opt") else None
        depth = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32) if cfg.get("need_depth") else None
        # What training keeps for the backward kernel (cfg):
        #   save_coef (default)  the SSIM coefficient maps of each pixel's arg-min frame (53 MB per scale at B=12):
        #                        the backward is then the 3x3 gather + geometry chain only, and re-samples the warped
        #                        colour of a pixel from the corners it gathers anyway (nothing else is stored);
        #   save_warp only       the warped colours (35 MB): the backward rebuilds the window statistics from them;
        #   neither              the backward re-warps the 2-pixel halo.
        keep = disp.requires_grad or P.requires_grad
        keep_coef = keep and cfg.get("save_coef", True)
        keep_warp = keep and cfg.get("save_warp", True) and not keep_coef
        warp = torch.empty(S, B, 3, H, W, device=dev, dtype=torch.float32) \
            if (cfg.get("need_warp") or keep_warp) else None
        reproj = torch.empty(B, S, H, W, device=dev, dtype=torch.float32) if cfg.get("need_reproj") else None
        coef = torch.empty(B, 9, H, W, device=dev, dtype=torch.float32) if keep_coef else None
        nws = lib().mdx_photometric_workspace_bytes(C.byref(d))
        ws = _ws(nws, dev)
        hook = _timing_hook("fwd") if keep_coef else None
        check(lib().mdx_photometric_fwd_timed(
            C.byref(d), ptr(disp), ptr(target), C.byref(src), ptr(invK), ptr(P),
            ptr(ident, optional=True) if automask else None, ptr(noise, optional=True) if automask else None,
            ptr(idx, torch.uint8), ptr(loss_sum), ptr(to_opt, optional=True), ptr(depth, optional=True),
            ptr(warp, optional=True), ptr(reproj, optional=True), ptr(coef, optional=True), ptr(ws, torch.float64),
            C.c_size_t(nws), stream(), C.byref(hook) if hook is not None else None), "mdx_photometric_fwd")
        ctx.save_for_backward(disp, P, target, invK, idx, *sources)
        ctx.warp = warp if keep_warp else None
        ctx.coef = coef
        ctx.cfg = dict(cfg)
        ctx.mark_non_differentiable(*[t for t in (idx, to_opt, depth, warp, reproj) if t is not None])
        return (loss_sum, idx, to_opt, depth, warp, reproj)

    @staticmethod
    def backward(ctx, g_sum, *_unused):
        disp, P, target, invK, idx, *sources = ctx.saved_tensors
        cfg = ctx.cfg
        B, _, h, w = disp.shape
        _, _, H, W = target.shape
        S = len(sources)
        d = _lib.make_desc(B, H, W, h, w, S, cfg["automask"], cfg["min_depth"], cfg["max_depth"])
        src = _lib.make_sources(sources)
        dev = disp.device
        gdisp = torch.empty_like(disp)
        gP = torch.empty(S, B, 3, 4, device=dev, dtype=torch.float32)
        g_dev = _f32c(g_sum.reshape(1))
        nws = lib().mdx_photometric_workspace_bytes(C.byref(d))
        ws = _ws(nws, dev)
        hook = _timing_hook("bwd") if ctx.coef is not None else None
        check(lib().mdx_photometric_bwd_timed(
            C.byref(d), ptr(disp), ptr(target), C.byref(src), ptr(invK), ptr(P), ptr(idx, torch.uint8),
            ptr(ctx.warp, optional=True), ptr(ctx.coef, optional=True), C.c_float(1.0), ptr(g_dev), ptr(gdisp), ptr(gP),
            ptr(ws, torch.float64),
            C.c_size_t(nws), stream(), C.byref(hook) if hook is not None else None), "mdx_photometric_bwd")
        return (gdisp, gP, None, None, None, None, None) + (None,) * S


def photometric_scale(disp, P, target, sources, invK, ident=None, noise=None, automask=True,
                      min_depth=0.1, max_depth=100.0, need_to_opt=False, need_depth=False,
                      need_warp=False, need_reproj=False, save_warp=True, save_coef=True):
    """Fused warp + SSIM/L1 + min for one scale.

    disp [B,1,h,w] (grad), P [S,B,3,4] (grad), target [B,3,H,W], sources: list of S [B,3,H,W],
    invK [B,4,4], ident/noise [B,S,H,W] (automask).  Returns dict with
    'sum' ([1], differentiable; divide by B*H*W for to_optimise.mean()), 'idx' (uint8 [B,H,W]) and the
    optional 'to_opt', 'depth', 'warp' ([S,B,3,H,W]), 'reproj' ([B,S,H,W]).
    """
    cfg = dict(automask=bool(automask), min_depth=float(min_depth), max_depth=float(max_depth),
               need_to_opt=need_to_opt, need_depth=need_depth, need_warp=need_warp, need_reproj=need_reproj,
               save_warp=save_warp, save_coef=save_coef)
    out = _PhotometricScale.apply(disp, P, target, invK, ident, noise, cfg, *sources)
    return dict(zip(("sum", "idx", "to_opt", "depth", "warp", "reproj"), out))


class noise_state(object):
    """Device-resident {seed, offset} of the kernel-side N(0,1) generator (csrc/photo_prologue.hip: Philox4x32-10 keyed by
    the seed, counter = (pixel, draw, offset)).  The offset is advanced ON THE DEVICE by the step's finishing kernel, so
    a captured step draws new numbers at every replay and nothing on the host takes part.  seed: torch.initial_seed()
    (what torch.manual_seed set) + `stream` (the data-parallel rank: ranks draw different noise)."""

    def __init__(self, device, seed=None, stream=0):
        seed = (torch.initial_seed() if seed is None else int(seed)) + 0x9E3779B97F4A7C15 * int(stream)
        self.tensor = torch.tensor([seed % (1 << 63), 0], dtype=torch.int64, device=device)

    def ptr(self):
        return C.c_void_p(self.tensor.data_ptr())


def photometric_prologue(target, sources, nscales, noises=None, rng=None, automask=True, need_ident=False, advance=False):
    """What the scales of a step share, once per step (include/mdx.h: mdx_photometric_prologue).  noises: list of
    nscales [B,S,H,W] tensors (injected: parity) or None -> drawn in the kernel from `rng` (a noise_state).
    -> dict: 'tstat' [B,H,W,6], 'bidfi' (list of [B,H,W,2]; automask), 'ident' ([B,S,H,W] or None), 'rng'."""
    target = _f32c(target)
    sources = [_f32c(x) for x in sources]
    B, _, H, W = target.shape
    S = len(sources)
    dev = target.device
    d = _lib.make_train_desc(B, H, W, S, [(H, W)] * nscales, automask, 0.1, 100.0)
    tstat = torch.empty(B, H, W, 6, device=dev, dtype=torch.float32)
    bidfi = [torch.empty(B, H, W, 2, device=dev, dtype=torch.float32) for _ in range(nscales)] if automask else None
    ident = torch.empty(B, S, H, W, device=dev, dtype=torch.float32) if (need_ident and automask) else None
    if automask and noises is None and rng is None:
        raise _lib.MdxError("photometric_prologue: either injected noises or a noise_state")
    if noises is not None:
        noises = [_f32c(x) for x in noises]
    check(lib().mdx_photometric_prologue(
        C.byref(d), ptr(target), C.byref(_lib.make_sources(sources)) if automask else None,
        _lib.ptr_array(noises) if (automask and noises is not None) else None,
        rng.ptr() if (rng is not None and noises is None) else None, int(bool(advance)),
        ptr(ident, optional=True), ptr(tstat), _lib.ptr_array(bidfi) if automask else None, stream()),
        "mdx_photometric_prologue")
    return dict(tstat=tstat, bidfi=bidfi, ident=ident, rng=(rng if noises is None else None))


def _train_launch(cfg, target, invK, ident, noises, sources, Pl, disps):
    """The launch behind _PhotometricTrain and _TrainLoss: every scale's photometric term (and, with cfg['grads'], its
    unit-upstream gradients gdisp / gP) in one launch.  Returns a dict of the buffers it filled."""
    nsc = len(disps)
    disps = [_f32c(x) for x in disps]
    target, invK = _f32c(target), _f32c(invK)
    sources = [_f32c(x) for x in sources]
    per_scale_P = len(Pl) > 1
    Ps = [_f32c(x) for x in Pl] if per_scale_P else [_f32c(Pl[0])] * nsc
    B, _, H, W = target.shape
    S = len(sources)
    automask = cfg["automask"]
    dev = target.device
    d = _lib.make_train_desc(B, H, W, S, [tuple(x.shape[2:]) for x in disps], automask, cfg["min_depth"],
                             cfg["max_depth"], cfg.get("rows_per_chunk", 0))
    src = _lib.make_sources(sources)
    pre = cfg.get("pre")
    if automask and pre is None:
        ident = _f32c(ident)
        noises = [_f32c(x) for x in noises]
    idx = [torch.empty(B, H, W, device=dev, dtype=torch.uint8) for _ in range(nsc)]
    sums = torch.empty(nsc, device=dev, dtype=torch.float32)
    grads = cfg.get("grads", True)         # False: every scale's forward alone (validation / torch.no_grad())
    gdisp = [torch.empty_like(x) for x in disps] if grads else None
    gP = torch.empty(nsc, S, B, 3, 4, device=dev, dtype=torch.float32) if grads else None
    depth0 = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32) if cfg.get("need_depth") else None
    to_opt = [torch.empty(B, H, W, device=dev, dtype=torch.float32) for _ in range(nsc)] \
        if cfg.get("need_to_opt") else None
    nws = lib().mdx_photometric_train_workspace_bytes(C.byref(d))
    ws = torch.empty(nws // 16 + 1, 2, dtype=torch.float64, device=dev)
    hook = _timing_hook("train" if grads else "eval")
    tail = (_lib.ptr_array(idx, torch.uint8), ptr(sums), _lib.ptr_array(gdisp) if grads else None,
            ptr(gP) if grads else None, ptr(depth0, optional=True),
            _lib.ptr_array(to_opt) if to_opt is not None else None, ptr(ws, torch.float64), C.c_size_t(nws), stream(),
            C.byref(hook) if hook is not None else None)
    if pre is not None:
        check(lib().mdx_photometric_train_pre(
            C.byref(d), _lib.ptr_array(disps), ptr(target), C.byref(src), ptr(invK), _lib.ptr_array(Ps),
            ptr(pre["tstat"]), _lib.ptr_array(pre["bidfi"]) if automask else None,
            pre["rng"].ptr() if pre.get("rng") is not None else None, *tail), "mdx_photometric_train_pre")
    else:
        check(lib().mdx_photometric_train(
            C.byref(d), _lib.ptr_array(disps), ptr(target), C.byref(src), ptr(invK), _lib.ptr_array(Ps),
            ptr(ident) if automask else None, _lib.ptr_array(noises) if automask else None, *tail),
            "mdx_photometric_train")
    return dict(sums=sums, idx=idx, gdisp=gdisp, gP=gP, depth0=depth0, to_opt=to_opt, per_scale_P=per_scale_P, pixels=B * H * W)


class _PhotometricTrain(torch.autograd.Function):
    """All scales, forward and gradient, in one launch (csrc/photo_train.hip).  Returns (sums [nscales], idx_0..,
    depth0 or None, to_opt_0.. or None); keeps only the unit-upstream gradients for backward."""

    @staticmethod
    def forward(ctx, target, invK, ident, cfg, noises, sources, nP, *tensors):
        # tensors = nP projection tensors (1: shared by the scales; nscales: one per scale) followed by the disparities
        r = _train_launch(cfg, target, invK, ident, noises, sources, tensors[:nP], tensors[nP:])
        idx, depth0, to_opt = r["idx"], r["depth0"], r["to_opt"]
        if r["gP"] is not None:
            ctx.save_for_backward(r["gP"], *r["gdisp"])
        ctx.per_scale_P = r["per_scale_P"]
        extras = idx + [depth0] + (to_opt if to_opt is not None else [])
        ctx.mark_non_differentiable(*[t for t in extras if t is not None])
        return (r["sums"], *idx, depth0, *(to_opt if to_opt is not None else []))

    @staticmethod
    def backward(ctx, g_sums, *_unused):
        gP, *gdisp = ctx.saved_tensors
        g = g_sums.float()
        gPs = gP * g.view(-1, 1, 1, 1, 1)
        gP_out = tuple(gPs[s] for s in range(gP.shape[0])) if ctx.per_scale_P else (gPs.sum(0),)
        return (None, None, None, None, None, None, None) + gP_out + tuple(gd * g[s] for s, gd in enumerate(gdisp))


def photometric_train(disps, P, target, sources, invK, ident=None, noises=None, automask=True, min_depth=0.1,
                      max_depth=100.0, need_depth=False, need_to_opt=False, rows_per_chunk=0, pre=None):
    """The training step's photometric term for every scale at once: forward and gradient in one launch.  Under
    torch.no_grad() (or when nothing requires a gradient) the forward-only form of the same kernel runs: every scale's
    loss sum, indices and depth in one launch, nothing computed or kept for a backward.

    disps: list of [B,1,h_s,w_s] (grad); P [S,B,3,4] (grad) shared by the scales, or a list with one P per scale
    (posecnn: the translation is scaled by the scale's mean inverse depth, processor.py:153-157);
    noises: list of [B,S,H,W] (automask).  pre: the dict photometric_prologue() returned for this step -- the kernel then
    loads the target's window statistics and each pixel's best identity channel instead of re-deriving them per scale
    (ident / noises are not needed; same results bit for bit).  Returns dict: 'sums' [nscales] (differentiable: sum over pixels of
    to_optimise per scale), 'idx' (list of uint8 [B,H,W]), 'depth' (scale 0, optional), 'to_opt' (optional list)."""
    Pl = list(P) if isinstance(P, (list, tuple)) else [P]
    if len(Pl) not in (1, len(disps)):
        raise _lib.MdxError("photometric_train: %d projections for %d scales (one, or one per scale)" % (len(Pl), len(disps)))
    grads = torch.is_grad_enabled() and any(t.requires_grad for t in list(disps) + Pl)
    cfg = dict(automask=bool(automask), min_depth=float(min_depth), max_depth=float(max_depth),
               need_depth=bool(need_depth), need_to_opt=bool(need_to_opt), rows_per_chunk=int(rows_per_chunk),
               grads=grads, pre=pre)
    n = len(disps)
    out = _PhotometricTrain.apply(target, invK, ident, cfg, list(noises) if noises is not None else None,
                                  list(sources), len(Pl), *Pl, *disps)
    res = dict(sums=out[0], idx=list(out[1:1 + n]), depth=out[1 + n])
    res["to_opt"] = list(out[2 + n:2 + 2 * n]) if need_to_opt else None
    return res


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, color, normalize=True):
        disp, color = _f32c(disp), _f32c(color)
        B, _, h, w = disp.shape
        dev = disp.device
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        g = torch.empty_like(disp) if need else None
        nws = lib().mdx_smooth_workspace_bytes(B, h, w)
        ws = _ws(nws, dev)
        check(lib().mdx_smooth_loss(B, h, w, ptr(disp), ptr(color), int(normalize), ptr(loss), ptr(g, optional=True),
                                    ptr(ws, torch.float64), C.c_size_t(nws), stream()), "mdx_smooth_loss")
        if need:
            ctx.save_for_backward(g)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        return g * gout, None, None


def smooth_loss(disp, color, normalize=True):
    """SmoothLoss()(disp, color) -> scalar (model_loss.py:107-116); normalize=False is the bare
    EdgeAwareSmooth (model_loss.py:77-88)."""
    return _SmoothLoss.apply(disp, color, normalize)


def _smooth_multi_launch(normalize, colors, disps, need):
    """The two launches behind _SmoothLossMulti and _TrainLoss -> (loss [nscales], unit-upstream gradients or None)."""
    disps = [_f32c(d) for d in disps]
    colors = [_f32c(c) for c in colors]
    n = len(disps)
    B = disps[0].shape[0]
    dev = disps[0].device
    hs = (C.c_int32 * n)(*[int(d.shape[2]) for d in disps])
    ws_ = (C.c_int32 * n)(*[int(d.shape[3]) for d in disps])
    loss = torch.empty(n, device=dev, dtype=torch.float32)
    gs = [torch.empty_like(d) for d in disps] if need else None
    nws = lib().mdx_smooth_multi_workspace_bytes(n, B, hs, ws_)
    ws = _ws(nws, dev)
    check(lib().mdx_smooth_loss_multi(n, B, hs, ws_, _lib.ptr_array(disps), _lib.ptr_array(colors), int(normalize),
                                      ptr(loss), _lib.ptr_array(gs) if need else None, ptr(ws, torch.float64),
                                      C.c_size_t(nws), stream()), "mdx_smooth_loss_multi")
    return loss, gs


class _SmoothLossMulti(torch.autograd.Function):
    @staticmethod
    def forward(ctx, normalize, colors, *disps):
        need = any(ctx.needs_input_grad[2:])
        loss, gs = _smooth_multi_launch(normalize, colors, disps, need)
        if need:
            ctx.save_for_backward(*gs)
        return loss

    @staticmethod
    def backward(ctx, gout):
        gs = ctx.saved_tensors
        return (None, None) + tuple(g * gout[k] for k, g in enumerate(gs))


def smooth_loss_multi(disps, colors, normalize=True):
    """SmoothLoss()(disp_s, color_s) for every scale of a step -> tensor [nscales]; each of the four passes is ONE
    launch for all scales (the per-scale op takes 16 launches per step)."""
    return _SmoothLossMulti.apply(bool(normalize), list(colors), *disps)


class _TrainLoss(torch.autograd.Function):
    """The whole loss of a training step (processor.py:163-217) as ONE autograd node: the smoothness launches, the training
    kernel, one single-thread launch that finishes the scalar (csrc/loss_total.hip) -- and a backward of ONE launch that turns
    the kernels' unit-upstream gradients into every scale's disparity gradient and the projection gradient."""

    @staticmethod
    def forward(ctx, target, invK, ident, cfg, noises, sources, colors, nP, *tensors):
        Pl, disps = tensors[:nP], tensors[nP:]
        nsc = len(disps)
        grads = cfg.get("grads", True)
        # the smoothness launches go BETWEEN the prologue and the training kernel (processor.compute_loss explains why) -- or
        # were issued ahead of this node (smooth_launch)
        if cfg.get("smooth") is not None:
            smooth, gs = cfg["smooth"]
            if grads and gs is None:
                raise _lib.MdxError("train_loss: the smoothness passes were launched without their gradients")
        else:
            smooth, gs = _smooth_multi_launch(True, colors, disps, grads)
        r = _train_launch(cfg, target, invK, ident, noises, sources, Pl, disps)
        total = torch.empty((), device=target.device, dtype=torch.float32)
        scales = (C.c_int32 * nsc)(*[int(v) for v in cfg["scales"]])
        check(lib().mdx_loss_total_fwd(nsc, ptr(r["sums"]), ptr(smooth), scales, C.c_int64(r["pixels"]),
                                       C.c_double(cfg["disp_smoothness"]), ptr(total), stream()), "mdx_loss_total_fwd")
        if grads:
            ctx.save_for_backward(r["gP"], *r["gdisp"], *gs)
        ctx.meta = (nsc, r["per_scale_P"], [int(v) for v in cfg["scales"]], r["pixels"], float(cfg["disp_smoothness"]),
                    [t.dtype for t in disps], [t.dtype for t in Pl])
        extras = [r["sums"], smooth] + r["idx"] + [r["depth0"]]
        ctx.mark_non_differentiable(*[t for t in extras if t is not None])
        return (total, r["sums"], smooth, *r["idx"], r["depth0"])

    @staticmethod
    def backward(ctx, g_total, *_unused):
        nsc, per_scale_P, scale_ids, pixels, lam, ddt, pdt = ctx.meta
        gP, *rest = ctx.saved_tensors
        gd, gs = rest[:nsc], rest[nsc:]
        g = _f32c(g_total).reshape(1)
        out = [torch.empty_like(x) for x in gd]
        nP = gP[0].numel()
        need_P = any(ctx.needs_input_grad[8:8 + len(pdt)])
        gP_out = (torch.empty_like(gP) if per_scale_P else torch.empty_like(gP[0])) if need_P else None
        scales = (C.c_int32 * nsc)(*scale_ids)
        counts = (C.c_int64 * nsc)(*[x.numel() for x in gd])
        check(lib().mdx_loss_total_bwd(nsc, ptr(g), scales, C.c_int64(pixels), C.c_double(lam), _lib.ptr_array(gd),
                                       _lib.ptr_array(gs), counts, _lib.ptr_array(out), ptr(gP) if need_P else None, nP,
                                       int(per_scale_P), ptr(gP_out, optional=True), stream()), "mdx_loss_total_bwd")
        if gP_out is None:
            gPs = (None,) * len(pdt)
        elif per_scale_P:
            gPs = tuple(gP_out[s].to(pdt[s]) for s in range(nsc))
        else:
            gPs = (gP_out.to(pdt[0]),)
        return (None,) * 8 + gPs + tuple(o.to(dt) for o, dt in zip(out, ddt))


def smooth_launch(disps, colors, need_grad=True):
    """The smoothness passes of a step issued on their own (no autograd node): (loss [nscales], unit-upstream gradients or None),
    to be handed to train_loss(smooth=...) -- for a caller that wants them on the stream before the training kernel's other
    inputs are ready."""
    return _smooth_multi_launch(True, list(colors), [d.detach() for d in disps], bool(need_grad))


def train_loss(disps, P, target, sources, invK, colors, scales, disp_smoothness, ident=None, noises=None, automask=True,
               min_depth=0.1, max_depth=100.0, need_depth=False, pre=None, smooth=None):
    """outputs["loss"] of a training step (processor.py:163-217) from the disparities and the projections: smoothness
    (smooth_loss_multi), photometric term (photometric_train) and the scalar tail, one autograd node.  colors: the target at
    every scale; scales: opt.scales.  Returns dict: 'loss' (scalar, differentiable), 'sums', 'smooth' [nscales], 'idx', 'depth'.
    The numbers are those of smooth_loss_multi + photometric_train + the reference's scalar ops, bit for bit (the projection
    gradient: the scales are summed in index order)."""
    Pl = list(P) if isinstance(P, (list, tuple)) else [P]
    if len(Pl) not in (1, len(disps)):
        raise _lib.MdxError("train_loss: %d projections for %d scales (one, or one per scale)" % (len(Pl), len(disps)))
    if len(colors) != len(disps) or len(scales) != len(disps):
        raise _lib.MdxError("train_loss: %d disparities, %d colour maps, %d scales" % (len(disps), len(colors), len(scales)))
    grads = torch.is_grad_enabled() and any(t.requires_grad for t in list(disps) + Pl)
    cfg = dict(automask=bool(automask), min_depth=float(min_depth), max_depth=float(max_depth), need_depth=bool(need_depth),
               need_to_opt=False, rows_per_chunk=0, grads=grads, pre=pre, scales=[int(v) for v in scales],
               disp_smoothness=float(disp_smoothness), smooth=smooth)
    n = len(disps)
    out = _TrainLoss.apply(target, invK, ident, cfg, list(noises) if noises is not None else None, list(sources),
                           list(colors), len(Pl), *Pl, *disps)
    return dict(loss=out[0], sums=out[1], smooth=out[2], idx=list(out[3:3 + n]), depth=out[3 + n])


# ------------------------------------------------------------------------------------------------
# fine-grained ops (reference API granularity)
# ------------------------------------------------------------------------------------------------
class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        x = _f32c(x)
        B, Cc, h, w = x.shape
        out = torch.empty(B, Cc, H, W, device=x.device, dtype=torch.float32)
        check(lib().mdx_interpolate_bilinear_fwd(ptr(x), B * Cc, h, w, ptr(out), H, W, stream()),
              "mdx_interpolate_bilinear_fwd")
        ctx.shape = (B, Cc, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cc, h, w, H, W = ctx.shape
        g = _f32c(g)
        gin = torch.empty(B, Cc, h, w, device=g.device, dtype=torch.float32)
        check(lib().mdx_interpolate_bilinear_bwd(ptr(g), B * Cc, H, W, ptr(gin), h, w, stream()),
              "mdx_interpolate_bilinear_bwd")
        return gin, None, None


def interpolate_bilinear(x, H, W):
    return _Interpolate.apply(x, int(H), int(W))


class _Disp2Depth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        disp = _f32c(disp)
        sd, depth = torch.empty_like(disp), torch.empty_like(disp)
        check(lib().mdx_disparity2depth_fwd(ptr(disp), C.c_size_t(disp.numel()), C.c_double(min_depth),
                                            C.c_double(max_depth), ptr(sd), ptr(depth), stream()),
              "mdx_disparity2depth_fwd")
        ctx.save_for_backward(disp)
        ctx.mm = (min_depth, max_depth)
        return sd, depth

    @staticmethod
    def backward(ctx, gsd, gdepth):
        (disp,) = ctx.saved_tensors
        gsd = _f32c(gsd) if gsd is not None else None
        gdepth = _f32c(gdepth) if gdepth is not None else None
        if gsd is None and gdepth is None:
            return None, None, None
        g = torch.empty_like(disp)
        check(lib().mdx_disparity2depth_bwd(ptr(disp), ptr(gsd, optional=True), ptr(gdepth, optional=True),
                                            C.c_size_t(disp.numel()), C.c_double(ctx.mm[0]),
                                            C.c_double(ctx.mm[1]), ptr(g), stream()), "mdx_disparity2depth_bwd")
        return g, None, None


def disparity2depth(disp, min_depth, max_depth):
    return _Disp2Depth.apply(disp, float(min_depth), float(max_depth))


class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, invK):
        depth, invK = _f32c(depth), _f32c(invK)
        B, _, H, W = depth.shape
        cam = torch.empty(B, 4, H * W, device=depth.device, dtype=torch.float32)
        check(lib().mdx_backproject_fwd(ptr(depth), ptr(invK), B, H, W, ptr(cam), stream()), "mdx_backproject_fwd")
        ctx.save_for_backward(invK)
        ctx.shape = (B, H, W)
        return cam

    @staticmethod
    def backward(ctx, gcam):
        (invK,) = ctx.saved_tensors
        B, H, W = ctx.shape
        gcam = _f32c(gcam)
        gdepth = torch.empty(B, 1, H, W, device=gcam.device, dtype=torch.float32)
        check(lib().mdx_backproject_bwd(ptr(gcam), ptr(invK), B, H, W, ptr(gdepth), stream()), "mdx_backproject_bwd")
        return gdepth, None


def backproject(depth, invK):
    return _Backproject.apply(depth, invK)


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam, P, H, W, eps):
        cam, P = _f32c(cam), _f32c(P)
        B = cam.shape[0]
        grid = torch.empty(B, H, W, 2, device=cam.device, dtype=torch.float32)
        check(lib().mdx_project_fwd(ptr(cam), ptr(P), B, H, W, C.c_float(eps), ptr(grid), stream()), "mdx_project_fwd")
        ctx.save_for_backward(cam, P)
        ctx.dims = (B, H, W, eps)
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        cam, P = ctx.saved_tensors
        B, H, W, eps = ctx.dims
        ggrid = _f32c(ggrid)
        gcam = torch.empty_like(cam)
        gP = torch.empty_like(P)
        nws = lib().mdx_project_workspace_bytes(B, H, W)
        ws = _ws(nws, cam.device)
        check(lib().mdx_project_bwd(ptr(cam), ptr(P), ptr(ggrid), B, H, W, C.c_float(eps), ptr(gcam), ptr(gP),
                                    ptr(ws, torch.float64), C.c_size_t(nws), stream()), "mdx_project_bwd")
        return gcam, gP, None, None, None


def project(cam, P, H, W, eps=1e-7):
    return _Project.apply(cam, P, int(H), int(W), float(eps))


class _GridSampleBorder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, grid):
        img, grid = _f32c(img), _f32c(grid)
        B, Cc, Hi, Wi = img.shape
        _, Ho, Wo, _ = grid.shape
        out = torch.empty(B, Cc, Ho, Wo, device=img.device, dtype=torch.float32)
        check(lib().mdx_grid_sample_border_fwd(ptr(img), ptr(grid), B, Cc, Hi, Wi, Ho, Wo, ptr(out), stream()),
              "mdx_grid_sample_border_fwd")
        ctx.save_for_backward(img, grid)
        return out

    @staticmethod
    def backward(ctx, gout):
        img, grid = ctx.saved_tensors
        B, Cc, Hi, Wi = img.shape
        _, Ho, Wo, _ = grid.shape
        gout = _f32c(gout)
        ggrid = torch.empty_like(grid)
        gimg = torch.empty_like(img) if ctx.needs_input_grad[0] else None
        check(lib().mdx_grid_sample_border_bwd(ptr(img), ptr(grid), ptr(gout), B, Cc, Hi, Wi, Ho, Wo, ptr(ggrid),
                                               ptr(gimg, optional=True), stream()), "mdx_grid_sample_border_bwd")
        return gimg, ggrid


def grid_sample_border(img, grid):
    return _GridSampleBorder.apply(img, grid)


class _ReprojectionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred, target = _f32c(pred), _f32c(target)
        B, _, H, W = pred.shape
        out = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
        check(lib().mdx_reprojection_loss_fwd(ptr(pred), ptr(target), B, H, W, ptr(out), stream()),
              "mdx_reprojection_loss_fwd")
        ctx.save_for_backward(pred, target)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        B, _, H, W = pred.shape
        gout = _f32c(gout)
        gp = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        gt = torch.empty_like(pred) if ctx.needs_input_grad[1] else None
        if gp is None and gt is None:
            return None, None
        check(lib().mdx_reprojection_loss_bwd(ptr(pred), ptr(target), ptr(gout), B, H, W, ptr(gp, optional=True),
                                              ptr(gt, optional=True), stream()), "mdx_reprojection_loss_bwd")
        return gp, gt


def reprojection_loss(pred, target):
    return _ReprojectionLoss.apply(pred, target)


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = _f32c(x), _f32c(y)
        B, Cc, H, W = x.shape
        out = torch.empty_like(x)
        check(lib().mdx_ssim_fwd(ptr(x), ptr(y), B * Cc, H, W, ptr(out), stream()), "mdx_ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, y = ctx.saved_tensors
        B, Cc, H, W = x.shape
        gout = _f32c(gout)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        if gx is None and gy is None:
            return None, None
        check(lib().mdx_ssim_bwd(ptr(x), ptr(y), ptr(gout), B * Cc, H, W, ptr(gx, optional=True), ptr(gy, optional=True),
                                 stream()), "mdx_ssim_bwd")
        return gx, gy


def ssim(x, y):
    """SSIM map clamp((1 - SSIM) / 2, 0, 1) of model_loss.py:28-41, differentiable in both images."""
    return _Ssim.apply(x, y)


def min_automask(ident, noise, reproj, automask=True, need_combined=False):
    """identity+noise / concat / torch.min(dim=1) -> (to_opt [B,H,W], idx uint8, combined or None)."""
    reproj = _f32c(reproj)
    B, S, H, W = reproj.shape
    dev = reproj.device
    if automask:
        ident, noise = _f32c(ident), _f32c(noise)
    Cc = 2 * S if automask else S
    comb = torch.empty(B, Cc, H, W, device=dev, dtype=torch.float32) if need_combined else None
    to_opt = torch.empty(B, H, W, device=dev, dtype=torch.float32)
    idx = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
    check(lib().mdx_min_automask_fwd(ptr(ident, optional=True) if automask else None,
                                     ptr(noise, optional=True) if automask else None, ptr(reproj), B, S, H, W,
                                     int(automask), ptr(comb, optional=True), ptr(to_opt), ptr(idx, torch.uint8),
                                     stream()), "mdx_min_automask_fwd")
    return to_opt, idx, comb


# ---- network glue around the convolutions (csrc/glue.hip, csrc/glue_nhwc.hip) -------------------------------------
# Every op below has a planar (NCHW memory) and a channels-last ([B][H][W][C] memory) form; the layout of the first
# map decides, the other maps are brought to it, outputs and gradients come back in it.
_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}
_CL = torch.channels_last


def _glue_dtype(t, what):
    if not t.is_cuda:
        raise _lib.MdxError("%s needs CUDA tensors (the HIP path has no CPU fallback), got %s" % (what, t.device))
    if t.dtype not in _DTYPE_CODE:
        raise _lib.MdxError("%s supports float32 / bfloat16, got %s" % (what, t.dtype))
    return _DTYPE_CODE[t.dtype]


from .layout import is_channels_last  # noqa: E402,F401  (re-exported: the layout test every op below dispatches on)


def _nhwc_ok(dtype, *channels):
    """csrc/nhwc_common.hpp: a thread owns one 16-byte channel vector."""
    n = 4 if dtype == torch.float32 else 8
    return all(c % n == 0 for c in channels)


def _as(t, cl):
    return t.contiguous(memory_format=_CL) if cl else t.contiguous()


class _DecoderGlue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, skip, bias, elu, upsample, out_dtype):
        in_code = _glue_dtype(raw, "decoder_glue")
        B, C1, h, w = raw.shape
        u = 2 if upsample else 1
        C2 = 0
        if skip is not None:
            if skip.dtype != raw.dtype or skip.shape[0] != B or tuple(skip.shape[2:]) != (u * h, u * w):
                raise _lib.MdxError("decoder_glue: skip %s %s does not match raw %s %s (x%d)"
                                    % (tuple(skip.shape), skip.dtype, tuple(raw.shape), raw.dtype, u))
            C2 = skip.shape[1]
        cl = is_channels_last(raw) and _nhwc_ok(raw.dtype, C1, C2)
        raw = _as(raw, cl)
        skip = _as(skip, cl) if skip is not None else None
        if bias is not None:
            bias = _f32c(bias)
            if bias.shape != (C1,):
                raise _lib.MdxError("decoder_glue: bias %s for %d channels" % (tuple(bias.shape), C1))
        out = torch.empty(B, C1 + C2, u * h + 2, u * w + 2, device=raw.device, dtype=out_dtype,
                          memory_format=_CL if cl else torch.contiguous_format)
        fn = lib().mdx_decoder_glue_nhwc_fwd if cl else lib().mdx_decoder_glue_fwd
        check(fn(ptr(raw, raw.dtype, cl=cl), ptr(skip, raw.dtype, cl=cl) if skip is not None else None,
                 ptr(bias) if bias is not None else None, ptr(out, out_dtype, cl=cl), B, C1, C2, h, w,
                 int(upsample), int(elu), in_code, _DTYPE_CODE[out_dtype], stream()), "mdx_decoder_glue_fwd")
        ctx.save_for_backward(raw, bias)
        ctx.meta = (C2, bool(elu), bool(upsample), out_dtype, cl)
        return out

    @staticmethod
    def backward(ctx, gout):
        raw, bias = ctx.saved_tensors
        C2, elu, upsample, out_dtype, cl = ctx.meta
        B, C1, h, w = raw.shape
        u = 2 if upsample else 1
        gout = _as(gout.to(out_dtype), cl)
        fmt = _CL if cl else torch.contiguous_format
        graw = torch.empty_like(raw)
        gskip = torch.empty(B, C2, u * h, u * w, device=raw.device, dtype=raw.dtype, memory_format=fmt) if C2 else None
        dbias = ws = None
        nws = 0
        code = _DTYPE_CODE[raw.dtype]
        if bias is not None:
            dbias = torch.empty(C1, device=raw.device, dtype=torch.float32)
            nws = (lib().mdx_decoder_glue_nhwc_workspace_bytes(B, C1, h, w, code) if cl
                   else lib().mdx_decoder_glue_workspace_bytes(B, C1, h, w))
            ws = torch.empty(nws // 4 + 1, device=raw.device, dtype=torch.float32)
        fn = lib().mdx_decoder_glue_nhwc_bwd if cl else lib().mdx_decoder_glue_bwd
        check(fn(ptr(gout, out_dtype, cl=cl), ptr(raw, raw.dtype, cl=cl), ptr(bias) if bias is not None else None,
                 ptr(graw, raw.dtype, cl=cl), ptr(gskip, raw.dtype, cl=cl) if C2 else None,
                 ptr(dbias) if bias is not None else None, B, C1, C2, h, w, int(upsample), int(elu), code,
                 _DTYPE_CODE[out_dtype], ptr(ws) if ws is not None else None, C.c_size_t(nws), stream()),
              "mdx_decoder_glue_bwd")
        return graw, gskip, dbias, None, None, None


def decoder_glue(raw, skip=None, elu=True, upsample=True, out_dtype=None, bias=None):
    """ReflectionPad2d(1)(cat(nearest_x2(ELU(raw + bias)), skip)) in one pass (reference: depth_decoder.py:44-47,96-106).

    raw [B,C1,h,w] (the convolution output BEFORE its ELU), skip [B,C2,u*h,u*w] or None -> [B,C1+C2,u*h+2,u*w+2].
    bias [C1]: the bias of the convolution that produced raw when that convolution was run without it (its add and
    its gradient reduction then happen inside these kernels instead of in two more passes over the map).
    elu / upsample switch the two stages off (plain pad: elu=False, upsample=False).  float32 or bfloat16; a
    channels-last `raw` gives a channels-last result (csrc/glue_nhwc.hip)."""
    out_dtype = out_dtype or raw.dtype
    if skip is not None and skip.dtype != raw.dtype:
        skip = skip.to(raw.dtype)
    return _DecoderGlue.apply(raw, skip, bias, bool(elu), bool(upsample), out_dtype)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, relu):
        code = _glue_dtype(x, "bias_act")
        B, Cc, H, W = x.shape
        x = _as(x, True)
        b32 = bias if bias.dtype == torch.float32 else bias.float()
        y = torch.empty_like(x)
        check(lib().mdx_bias_act_nhwc_fwd(ptr(x, x.dtype, cl=True), ptr(b32), ptr(y, x.dtype, cl=True), B, Cc, H, W, int(relu), code,
                                          stream()), "mdx_bias_act_nhwc_fwd")
        ctx.save_for_backward(y)
        ctx.meta = (bool(relu), bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        relu, bdt = ctx.meta
        B, Cc, H, W = y.shape
        code = _DTYPE_CODE[y.dtype]
        dy = _as(dy.to(y.dtype), True)
        dx = torch.empty_like(y)
        db = torch.empty(Cc, device=y.device, dtype=torch.float32)
        nws = lib().mdx_bias_act_nhwc_workspace_bytes(B, Cc, H, W, code)
        ws = torch.empty(nws // 4 + 1, device=y.device, dtype=torch.float32)
        check(lib().mdx_bias_act_nhwc_bwd(ptr(dy, y.dtype, cl=True), ptr(y, y.dtype, cl=True), ptr(dx, y.dtype, cl=True), ptr(db), B, Cc,
                                          H, W, int(relu), code, ptr(ws), C.c_size_t(nws), stream()), "mdx_bias_act_nhwc_bwd")
        return dx, db.to(bdt), None


def bias_act_ok(x):
    """Can bias_act take this map?  (channels-last GPU map, float32 / bfloat16, whole 16-byte channel vectors.)"""
    return (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPE_CODE and is_channels_last(x) and _nhwc_ok(x.dtype, x.shape[1]))


def bias_act(x, bias, relu=True):
    """act(x + bias[None, :, None, None]) on a channels-last map (csrc/pose_head_nhwc.hip): what follows each of the pose decoder's
    convolutions (pose_decoder.py:24-50) when the convolution runs without its bias -- one launch; backward one launch for dx and
    the bias gradient's block partials + a finishing pass."""
    if not bias_act_ok(x):
        raise _lib.MdxError("bias_act: needs a channels-last GPU map with whole 16-byte channel vectors, got %s %s" % (tuple(x.shape), x.dtype))
    return _BiasAct.apply(x, bias, bool(relu))


class _MeanBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, scale):
        code = _glue_dtype(x, "mean_bias")
        B, Cc, H, W = x.shape
        x = _as(x, True)
        b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        out = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
        check(lib().mdx_mean_bias_nhwc_fwd(ptr(x, x.dtype, cl=True), ptr(b32) if b32 is not None else None, ptr(out), B, Cc, H, W,
                                           C.c_float(scale), code, stream()), "mdx_mean_bias_nhwc_fwd")
        ctx.meta = (tuple(x.shape), x.dtype, float(scale), None if bias is None else bias.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        (B, Cc, H, W), dt, scale, bdt = ctx.meta
        g = _f32c(g)
        dx = torch.empty((B, Cc, H, W), device=g.device, dtype=dt, memory_format=_CL)
        db = torch.empty(Cc, device=g.device, dtype=torch.float32) if bdt is not None else None
        check(lib().mdx_mean_bias_nhwc_bwd(ptr(g), ptr(dx, dt, cl=True), ptr(db) if db is not None else None, B, Cc, H, W,
                                           C.c_float(scale), _DTYPE_CODE[dt], stream()), "mdx_mean_bias_nhwc_bwd")
        return dx, (db.to(bdt) if db is not None else None), None


def mean_bias(x, bias=None, scale=1.0):
    """scale * (x.mean((2, 3)) + bias) -> [B, C] float32 for a channels-last GPU map: the pose head's spatial mean and 0.01
    (pose_decoder.py:51-53) behind a convolution that ran without its bias.  One launch each way."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPE_CODE and is_channels_last(x)):
        raise _lib.MdxError("mean_bias: needs a channels-last GPU map, got %s %s" % (tuple(x.shape), x.dtype))
    return _MeanBias.apply(x, bias, float(scale))


class FrameStack(object):
    """torch.cat([torch.cat(block, 1) for block in blocks], 0) of planar [n,3,H,W] frames that nobody has materialised yet: what
    the step hands the pose encoder for its frame pairs (processor.py:61-75), so that encoder_input can write the normalised,
    channels-last network input in one pass.  tensor() is the concatenation itself (for a consumer that cannot take the stack)."""

    def __init__(self, blocks):
        self.blocks = [list(b) for b in blocks]

    def tensor(self):
        rows = [torch.cat(b, 1) if len(b) > 1 else b[0] for b in self.blocks]
        return torch.cat(rows, 0) if len(rows) > 1 else rows[0]

    def ok(self):
        fr = [t for b in self.blocks for t in b]
        first = fr[0]
        return (1 <= len(self.blocks) <= 2 and len({len(b) for b in self.blocks}) == 1 and 1 <= len(self.blocks[0]) <= 2
                and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape == first.shape and t.shape[1] == 3
                        and t.is_contiguous() and not t.requires_grad for t in fr))


def encoder_input(stack, mean=0.45, std=0.225, dtype=torch.float32):
    """(cat(frames) - mean) / std as a channels-last map of `dtype` in one launch (depth_encoder.py:89 behind processor.py:61-75's
    concatenations).  The division is ATen-GPU's: a multiplication by the float32 reciprocal."""
    if not stack.ok():
        raise _lib.MdxError("encoder_input: needs 1-2 blocks of 1-2 planar float32 [n,3,H,W] GPU frames that require no gradient")
    fr = [t for b in stack.blocks for t in b]
    n, _, H, W = fr[0].shape
    blocks, groups = len(stack.blocks), len(stack.blocks[0])
    out = torch.empty((blocks * n, 3 * groups, H, W), device=fr[0].device, dtype=dtype, memory_format=_CL)
    inv = float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(std, dtype=torch.float32))
    check(lib().mdx_encoder_input_nhwc(_lib.ptr_array(fr), blocks, groups, n, H, W, C.c_float(mean), C.c_float(inv),
                                       ptr(out, dtype, cl=True), _DTYPE_CODE[dtype], stream()), "mdx_encoder_input_nhwc")
    return out


class _ThinConv3x3(torch.autograd.Function):
    """conv2d(x, w) (3x3, no padding, no bias) whose WEIGHT gradient is csrc/thinconv_nhwc.hip's MFMA kernel; forward and data
    gradient stay MIOpen's."""
    ARGS = ([1, 1], [0, 0], [1, 1], False, [0, 0], 1)

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return torch.ops.aten.convolution(x, weight, None, *_ThinConv3x3.ARGS)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, weight, None, *_ThinConv3x3.ARGS, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            B, Cin, Hp, Wp = x.shape
            gy = _as(gy, True)
            gw = torch.empty_like(weight)
            nws = lib().mdx_thin_conv3x3_wgrad_workspace_bytes(B, Cin, 16, Hp - 2, Wp - 2)
            ws = torch.empty(nws // 4 + 1, device=x.device, dtype=torch.float32)
            check(lib().mdx_thin_conv3x3_wgrad(ptr(x, cl=True), ptr(gy, cl=True), ptr(gw, cl="any"), C.c_int64(gw.stride(0)),
                                               C.c_int64(gw.stride(1)), C.c_int64(gw.stride(2)), C.c_int64(gw.stride(3)), B, Cin, 16,
                                               Hp - 2, Wp - 2, ptr(ws), C.c_size_t(nws), stream()), "mdx_thin_conv3x3_wgrad")
        return gx, gw


def thin_conv_ok(x, weight):
    """A 3x3 convolution thin_conv3x3 takes: float32 channels-last GPU map [B,Cin,h+2,w+2] with Cin 16 or 32, weight [16,Cin,3,3]
    (dense, either memory format), w a multiple of 4, no autocast."""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and is_channels_last(x)
            and tuple(weight.shape) == (16, x.shape[1], 3, 3) and x.shape[1] in (16, 32) and x.shape[3] > 2 and (x.shape[3] - 2) % 4 == 0
            and x.shape[2] > 2 and (weight.is_contiguous() or weight.is_contiguous(memory_format=_CL)) and not torch.is_autocast_enabled())


def thin_conv3x3(x, weight):
    """conv2d(x, weight) for the decoder's thin convolutions (depth_decoder.py:96-106: 16 output channels on the two biggest maps)
    with the weight gradient from one MFMA launch (csrc/thinconv_nhwc.hip: 101 us against MIOpen's 190 on the 16-channel map of
    scale 0); forward and data gradient are MIOpen's.  x: the reflection-padded channels-last input."""
    if not thin_conv_ok(x, weight):
        raise _lib.MdxError("thin_conv3x3: needs a float32 channels-last map with 16 or 32 channels and a [16,Cin,3,3] weight, got %s %s"
                            % (tuple(x.shape), tuple(weight.shape)))
    return _ThinConv3x3.apply(x, weight)


class _DispHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        code = _glue_dtype(x, "disp_head")
        B, Cc, Hp, Wp = x.shape
        x = _as(x, True)
        w32 = weight if weight.dtype == torch.float32 else weight.float()
        if not (w32.is_contiguous() or w32.is_contiguous(memory_format=torch.channels_last)):
            w32 = w32.contiguous()
        b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        disp = torch.empty(B, 1, Hp - 2, Wp - 2, device=x.device, dtype=torch.float32)
        st = (C.c_int64(w32.stride(1)), C.c_int64(w32.stride(2)), C.c_int64(w32.stride(3)))
        check(lib().mdx_disp_head_nhwc_fwd(ptr(x, x.dtype, cl=True), ptr(w32, cl="any"), *st, ptr(b32) if b32 is not None else None,
                                           ptr(disp), B, Cc, Hp - 2, Wp - 2, code, stream()), "mdx_disp_head_nhwc_fwd")
        ctx.save_for_backward(x, w32, disp)
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype)
        return disp

    @staticmethod
    def backward(ctx, g):
        x, w32, disp = ctx.saved_tensors
        wdt, bdt = ctx.meta
        B, Cc, Hp, Wp = x.shape
        code = _DTYPE_CODE[x.dtype]
        g = _f32c(g)
        gx = torch.empty_like(x)
        gw = torch.empty_like(w32)                     # the weight's own strides
        gb = torch.empty(1, device=x.device, dtype=torch.float32) if bdt is not None else None
        nws = lib().mdx_disp_head_nhwc_workspace_bytes(B, Cc, Hp - 2, Wp - 2, code)
        ws = torch.empty(nws // 4 + 1, device=x.device, dtype=torch.float32)
        st = (C.c_int64(w32.stride(1)), C.c_int64(w32.stride(2)), C.c_int64(w32.stride(3)))
        check(lib().mdx_disp_head_nhwc_bwd(ptr(x, x.dtype, cl=True), ptr(w32, cl="any"), *st, ptr(g), ptr(disp),
                                           ptr(gx, x.dtype, cl=True), ptr(gw, cl="any"), ptr(gb) if gb is not None else None, B, Cc,
                                           Hp - 2, Wp - 2, code, ptr(ws), C.c_size_t(nws), stream()), "mdx_disp_head_nhwc_bwd")
        return gx, gw.to(wdt), (gb.to(bdt) if gb is not None else None)


def disp_head_ok(x, weight):
    """Can disp_head take this padded map and this head's weight?  (channels-last GPU map, one output channel, 3x3, a channel count
    that is a power of two from 4 to 256.)"""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in _DTYPE_CODE and is_channels_last(x)):
        return False
    n = 4                                  # channels per thread, either dtype (csrc/disp_head_nhwc.hip: DH_N)
    lp = x.shape[1] // n
    return (tuple(weight.shape) == (1, x.shape[1], 3, 3) and x.shape[1] % n == 0 and 1 <= lp <= 64 and lp & (lp - 1) == 0
            and x.shape[2] > 2 and x.shape[3] > 2)


def disp_head(x, weight, bias=None):
    """sigmoid(conv2d(x, weight, bias)) for a disparity head (depth_decoder.py:73-74,108-110): x [B,C,h+2,w+2] the reflection-padded,
    channels-last input, weight [1,C,3,3], bias [1] -> disp [B,1,h,w] float32 (csrc/disp_head_nhwc.hip: one launch forward, one +
    a finishing pass backward; float32 accumulation whatever x's dtype)."""
    if not disp_head_ok(x, weight):
        raise _lib.MdxError("disp_head: needs a channels-last GPU map [B,C,h+2,w+2] with C a power of two from 4 to 256 "
                            "and a [1,C,3,3] weight; got x %s %s, weight %s" % (tuple(x.shape), x.dtype, tuple(weight.shape)))
    return _DispHead.apply(x, weight, bias)


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fork):
        code = _glue_dtype(x, "maxpool3s2")
        B, Cc, H, W = x.shape
        cl = is_channels_last(x) and _nhwc_ok(x.dtype, Cc)
        x = _as(x, cl)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        fmt = _CL if cl else torch.contiguous_format
        out = torch.empty(B, Cc, Ho, Wo, device=x.device, dtype=x.dtype, memory_format=fmt)
        arg = torch.empty(B, Cc, Ho, Wo, device=x.device, dtype=torch.uint8, memory_format=fmt)
        if cl:
            check(lib().mdx_maxpool3s2_nhwc_fwd(ptr(x, x.dtype, cl=True), ptr(out, x.dtype, cl=True),
                                                ptr(arg, torch.uint8, cl=True), B, Cc, H, W, code, stream()),
                  "mdx_maxpool3s2_nhwc_fwd")
        else:
            check(lib().mdx_maxpool3s2_fwd(ptr(x, x.dtype), ptr(out, x.dtype), ptr(arg, torch.uint8), B * Cc, H, W, code,
                                           stream()), "mdx_maxpool3s2_fwd")
        ctx.save_for_backward(arg)
        ctx.meta = (H, W, x.dtype, cl)
        ctx.set_materialize_grads(False)
        return (out, out.view_as(out)) if fork else out

    @staticmethod
    def backward(ctx, gout, gout2=None):
        (arg,) = ctx.saved_tensors
        H, W, dtype, cl = ctx.meta
        B, Cc = arg.shape[:2]
        gout, gout2 = _two_grads(gout, gout2, dtype, cl, fused_add=cl)
        if gout is None:
            return None, None
        gin = torch.empty(B, Cc, H, W, device=gout.device, dtype=dtype, memory_format=_CL if cl else torch.contiguous_format)
        if cl:
            check(lib().mdx_maxpool3s2_nhwc_bwd(ptr(gout, dtype, cl=True), ptr(gout2, dtype, cl=True) if gout2 is not None else None,
                                                ptr(arg, torch.uint8, cl=True), ptr(gin, dtype, cl=True), B, Cc, H, W,
                                                _DTYPE_CODE[dtype], stream()), "mdx_maxpool3s2_nhwc_bwd")
        else:
            check(lib().mdx_maxpool3s2_bwd(ptr(gout, dtype), ptr(arg, torch.uint8), ptr(gin, dtype), B * Cc, H, W,
                                           _DTYPE_CODE[dtype], stream()), "mdx_maxpool3s2_bwd")
        return gin, None


def _two_grads(g1, g2, dtype, cl, fused_add):
    """The one or two upstream gradients of a forked output, in the op's dtype and layout.  fused_add: the kernel adds
    the second one on the way in (channels-last forms); otherwise they are added here."""
    gs = [_as(g.to(dtype), cl) for g in (g1, g2) if g is not None]
    if not gs:
        return None, None
    if len(gs) == 2 and not fused_add:
        gs = [gs[0] + gs[1]]
    return gs[0], (gs[1] if len(gs) == 2 else None)


def maxpool3s2(x, fork=False):
    """MaxPool2d(kernel_size=3, stride=2, padding=1) (ResNet stem); backward is a gather, not ATen's atomics.
    fork: return the result twice (two tensors on one storage, not to be modified in place) for its two consumers --
    the backward then receives their gradients separately and adds them inside its kernel."""
    return _MaxPool3s2.apply(x, bool(fork))


# ---- training-mode BatchNorm2d fused with residual add and ReLU (csrc/norm.hip, csrc/norm_nhwc.hip) ---------------
class _BnAct(torch.autograd.Function):
    """groups > 1: the batch is `groups` consecutive sub-batches, each normalised with its own statistics and the
    running statistics updated once per sub-batch, in order -- exactly what `groups` separate calls of the module
    would do (the pose network sees both frame pairs of a step in one batch this way)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, eps, momentum, relu, groups, fork):
        code = _glue_dtype(x, "bn_act")
        B, Cc, H, W = x.shape
        if B % groups:
            raise _lib.MdxError("bn_act: batch %d is not divisible into %d groups" % (B, groups))
        cl = is_channels_last(x) and _nhwc_ok(x.dtype, Cc)
        x = _as(x, cl)
        if res is not None:
            if res.shape != x.shape or res.dtype != x.dtype:
                raise _lib.MdxError("bn_act: residual %s %s does not match x %s %s"
                                    % (tuple(res.shape), res.dtype, tuple(x.shape), x.dtype))
            res = _as(res, cl)
        weight, bias = _f32c(weight), _f32c(bias)
        y = torch.empty_like(x)
        save_mean = torch.empty(groups, Cc, device=x.device, dtype=torch.float32)
        save_invstd = torch.empty(groups, Cc, device=x.device, dtype=torch.float32)
        Bg = B // groups
        nws = lib().mdx_bn_nhwc_workspace_bytes(Bg, Cc, H, W, groups, code) if cl else lib().mdx_bn_workspace_bytes(Bg, Cc, H, W)
        ws = torch.empty(nws // 4 + 1, device=x.device, dtype=torch.float32)
        fn = lib().mdx_bn_act_nhwc_fwd if cl else lib().mdx_bn_act_fwd
        check(fn(ptr(x, x.dtype, cl=cl), ptr(res, x.dtype, cl=cl) if res is not None else None, ptr(weight), ptr(bias),
                 ptr(running_mean) if running_mean is not None else None,
                 ptr(running_var) if running_var is not None else None, ptr(y, x.dtype, cl=cl), ptr(save_mean),
                 ptr(save_invstd), Bg, Cc, H, W, groups, C.c_float(eps), C.c_float(momentum), int(relu), code, ptr(ws),
                 C.c_size_t(nws), stream()), "mdx_bn_act_fwd")
        # channels-last float32, ReLU, no residual: backward re-derives the ReLU mask from x (y is not read: one map less per pass;
        # 760.3 -> 765.4 images/s.  bfloat16 maps: 1498 -> 1492 -- the re-derivation costs more than half a map of traffic)
        mask_from_x = cl and bool(relu) and res is None and x.dtype == torch.float32
        ctx.save_for_backward(x, bias if mask_from_x else y, weight, save_mean, save_invstd)
        ctx.meta = (bool(relu), res is not None, groups, cl, mask_from_x)
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        ctx.set_materialize_grads(False)
        return (y, y.view_as(y)) if fork else y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        x, y, weight, save_mean, save_invstd = ctx.saved_tensors
        relu, has_res, groups, cl, mask_from_x = ctx.meta
        bias = None
        if mask_from_x:
            y, bias = None, y
        none = (None,) * 11
        dy, dy2 = _two_grads(dy, dy2, x.dtype, cl, fused_add=cl)
        if dy is None:
            return none
        B, Cc, H, W = x.shape
        Bg = B // groups
        code = _DTYPE_CODE[x.dtype]
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dgamma = torch.empty(Cc, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(Cc, device=x.device, dtype=torch.float32)
        if cl:
            nws = lib().mdx_bn_nhwc_workspace_bytes(Bg, Cc, H, W, groups, code)
            ws = torch.empty(nws // 4 + 1, device=x.device, dtype=torch.float32)
            check(lib().mdx_bn_act_nhwc_bwd(
                ptr(dy, x.dtype, cl=True), ptr(dy2, x.dtype, cl=True) if dy2 is not None else None,
                ptr(y, x.dtype, cl=True) if y is not None else None, ptr(x, x.dtype, cl=True), ptr(weight),
                ptr(bias) if bias is not None else None, ptr(save_mean), ptr(save_invstd), ptr(dx, x.dtype, cl=True),
                ptr(dres, x.dtype, cl=True) if has_res else None, ptr(dgamma), ptr(dbeta), Bg, Cc, H, W, groups, int(relu),
                code, ptr(ws), C.c_size_t(nws), stream()), "mdx_bn_act_nhwc_bwd")
        else:
            nws = lib().mdx_bn_workspace_bytes(Bg, Cc, H, W)
            ws = torch.empty(nws // 4 + 1, device=x.device, dtype=torch.float32)
            check(lib().mdx_bn_act_bwd(
                ptr(dy, x.dtype), ptr(y, x.dtype), ptr(x, x.dtype), ptr(weight), ptr(save_mean), ptr(save_invstd),
                ptr(dx, x.dtype), ptr(dres, x.dtype) if has_res else None, ptr(dgamma), ptr(dbeta), Bg, Cc, H, W, groups,
                int(relu), code, ptr(ws), C.c_size_t(nws), stream()), "mdx_bn_act_bwd")
        return (dx, dres, dgamma, dbeta) + (None,) * 7


def bn_act(x, weight, bias, running_mean, running_var, eps=1e-5, momentum=0.1, residual=None, relu=True, groups=1,
           fork=False):
    """Training-mode batch norm over (B,H,W) of x [B,C,H,W] with batch statistics, then `+ residual`, then ReLU
    (csrc/norm.hip; csrc/norm_nhwc.hip when x is channels-last).  Updates running_mean / running_var in place like
    torch.nn.functional.batch_norm(training=True).  float32 or bfloat16 activations, float32 parameters.
    groups: number of consecutive sub-batches normalised independently (see _BnAct).
    fork: return the result twice (two tensors on one storage, not to be modified in place), one per consumer -- a ResNet
    block's output feeds the next block's first convolution and its identity path; the backward then receives the two
    gradients separately and adds them inside its kernel instead of autograd doing so in a pass of its own."""
    return _BnAct.apply(x, residual, weight, bias, running_mean, running_var, float(eps), float(momentum), bool(relu),
                        int(groups), bool(fork))


# ---- param2matrix (csrc/pose.hip) ----------------------------------------------------------------------------------
class _Param2Matrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        aa, tr = _f32c(axisangle).reshape(-1, 3), _f32c(translation).reshape(-1, 3)
        N = aa.shape[0]
        if tr.shape[0] != N:
            raise _lib.MdxError("param2matrix: %d rotations, %d translations" % (N, tr.shape[0]))
        M = torch.empty(N, 4, 4, device=aa.device, dtype=torch.float32)
        check(lib().mdx_param2matrix_fwd(ptr(aa), ptr(tr), N, int(invert), ptr(M), stream()), "mdx_param2matrix_fwd")
        ctx.save_for_backward(aa, tr)
        ctx.meta = (bool(invert), axisangle.shape, translation.shape)
        return M

    @staticmethod
    def backward(ctx, gM):
        aa, tr = ctx.saved_tensors
        invert, sa, st = ctx.meta
        gM = _f32c(gM)
        gaa, gtr = torch.empty_like(aa), torch.empty_like(tr)
        check(lib().mdx_param2matrix_bwd(ptr(aa), ptr(tr), ptr(gM), aa.shape[0], int(invert), ptr(gaa), ptr(gtr),
                                         stream()), "mdx_param2matrix_bwd")
        return gaa.reshape(sa), gtr.reshape(st), None


def param2matrix(axisangle, translation, invert=False):
    """(axis-angle [N,1,3], translation [N,1,3]) -> camera-to-camera matrix [N,4,4] (reference warp.py:126-153) in
    one launch; backward by forward-mode differentiation inside the kernel."""
    return _Param2Matrix.apply(axisangle, translation, bool(invert))


class _PoseProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, K, B, row0, frame, invert):
        raw, K = _f32c(raw), _f32c(K)
        M, F = raw.shape[0], raw.shape[1]
        S = len(row0)
        T = torch.empty(S, B, 4, 4, device=raw.device, dtype=torch.float32)
        P = torch.empty(S, B, 3, 4, device=raw.device, dtype=torch.float32)
        sel = tuple((C.c_int32 * S)(*[int(v) for v in a]) for a in (row0, frame, invert))
        check(lib().mdx_pose_projection_fwd(ptr(raw), M, F, ptr(K), B, S, *sel, ptr(T), ptr(P), stream()),
              "mdx_pose_projection_fwd")
        ctx.save_for_backward(raw, K)
        ctx.meta = (B, [int(v) for v in row0], [int(v) for v in frame], [int(v) for v in invert])
        return T, P

    @staticmethod
    def backward(ctx, gT, gP):
        raw, K = ctx.saved_tensors
        B, row0, frame, invert = ctx.meta
        M, F = raw.shape[0], raw.shape[1]
        S = len(row0)
        graw = torch.empty_like(raw)
        sel = tuple((C.c_int32 * S)(*a) for a in (row0, frame, invert))
        check(lib().mdx_pose_projection_bwd(ptr(raw), M, F, ptr(K), B, S, *sel,
                                            ptr(_f32c(gP)) if gP is not None else None,
                                            ptr(_f32c(gT)) if gT is not None else None, ptr(graw), stream()),
              "mdx_pose_projection_bwd")
        return graw, None, None, None, None, None


def pose_projection(raw, K, row0, frame, invert):
    """The pose head's output -> (T [S,B,4,4], P [S,B,3,4]) for every source frame in one launch (and one backward):
    raw [M,F,1,6] or [M,F,6] (axis-angle | translation, pose_decoder.py:51-53); source s takes rows row0[s]..row0[s]+B-1, entry
    frame[s], inverted if invert[s] (processor.py:61-83); P = (K @ T)[:, :3] (processor.py:143-160).  B = K.shape[0].
    The numbers are those of the slices + param2matrix + compose_projection."""
    if K.requires_grad:
        raise _lib.MdxError("pose_projection: K must not require a gradient (use param2matrix + compose_projection)")
    B = K.shape[0]
    return _PoseProjection.apply(raw.reshape(raw.shape[0], raw.shape[1], 6), K, B, list(row0), list(frame), list(invert))


# ---- train-time depth monitor (csrc/monitor.hip) -------------------------------------------------------------------
def depth_monitor(pred, gt, window, min_depth=1e-3, max_depth=80.0):
    """compute_depth_metric's seven numbers (reference model_metric.py:70-105) + the number of valid pixels, as a
    float32 tensor [8] on the device: six small launches, no compaction, no sort, no synchronisation.
    pred [B,1,h,w], gt [B,1,gh,gw]; window = (r0, r1, c0, c1) of the ground-truth image."""
    pred, gt = _f32c(pred), _f32c(gt)
    B, _, h, w = pred.shape
    _, _, gh, gw = gt.shape
    r0, r1, c0, c1 = (int(v) for v in window)
    out = torch.empty(8, device=pred.device, dtype=torch.float32)
    nws = lib().mdx_depth_monitor_workspace_bytes(B, r0, r1, c0, c1)
    ws = _ws(nws, pred.device)
    check(lib().mdx_depth_monitor(ptr(pred), B, h, w, ptr(gt), gh, gw, r0, r1, c0, c1, C.c_float(min_depth),
                                  C.c_float(max_depth), ptr(out), ptr(ws, torch.float64), C.c_size_t(nws), stream()),
          "mdx_depth_monitor")
    return out
