#!/usr/bin/env python3
"""Kernel micro-benchmark: launches only the hand-written photometric kernels (B=12, 192x640, S=2, the
BASELINE configs[1] shape) so that rocprofv3 --kernel-trace / --pmc sees nothing else.

    python tools/kbench.py [--reps 20] [--B 12] [--S 2] [--what fwd,bwd,ident,smooth,train] [--rows 0]

`train` = mdx_photometric_train: all scales, forward and gradient, one launch (csrc/photo_train.hip); its fused
kernel is timed by the HIP events the *_timed hook records around it, the whole call (with the finishing pass and the
upsample transposes) by events around the call.
"""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import _lib  # noqa: E402
from mdx import functional as F  # noqa: E402
from model_tool.synthetic import make_K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--B", type=int, default=12)
    ap.add_argument("--H", type=int, default=192)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--S", type=int, default=2)
    ap.add_argument("--what", type=str, default="fwd,bwd,ident,smooth,train")
    ap.add_argument("--rows", type=int, default=0, help="rows per chunk of the training kernel (0 = library default)")
    ap.add_argument("--nscales", type=int, default=4)
    ap.add_argument("--save_warp", type=int, default=2,
                    help="0: backward re-warps; 1: forward stores the warp; 2: SSIM coefficient maps only (training form); 3: both")
    a = ap.parse_args()
    dev = "cuda:0"
    lib = _lib.lib()
    B, H, W, S = a.B, a.H, a.W, a.S
    g = torch.Generator().manual_seed(0)
    base = torch.rand(B, 3, H // 4, W // 4, generator=g)
    base = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=False)
    tgt = (base + 0.05 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1).to(dev)
    srcs = [(torch.roll(base, 3 * (k + 1), 3) + 0.05 * torch.rand(B, 3, H, W, generator=g)).clamp(0, 1).to(dev)
            for k in range(S)]
    K, invK = make_K(H, W)
    K, invK = K.to(dev).repeat(B, 1, 1), invK.to(dev).repeat(B, 1, 1)
    Ts = []
    for k in range(S):
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, :3, 3] = 0.03 * torch.randn(B, 3, generator=g)
        Ts.append(T.to(dev))
    P = torch.stack([F.compose_projection(K, T) for T in Ts])
    ident = F.identity_loss(tgt, srcs)
    noise = torch.randn(B, S, H, W, generator=g).to(dev)
    idx = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
    src = _lib.make_sources(srcs)
    what = a.what.split(",")
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    all_disps = []
    for s in range(4):
        h, w = H >> s, W >> s
        # network-like disparity: smooth field (low-res noise, bilinearly upsampled) through a sigmoid
        lo = torch.randn(B, 1, max(H // 32, 2), max(W // 32, 2), generator=g)
        disp = torch.sigmoid(torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear",
                                                             align_corners=False)).contiguous().to(dev)
        all_disps.append(disp)
        warp = torch.empty(S, B, 3, H, W, device=dev)
        coef = torch.empty(B, 9, H, W, device=dev)
        color_s = torch.nn.functional.avg_pool2d(tgt, 2 ** s) if s else tgt
        d = _lib.make_desc(B, H, W, h, w, S, True, 0.1, 100.0)
        nws = lib.mdx_photometric_workspace_bytes(C.byref(d))
        ws = torch.empty(nws // 8 + 1, dtype=torch.float64, device=dev)
        gdisp, gP = torch.empty_like(disp), torch.empty(S, B, 3, 4, device=dev)
        loss = torch.empty(1, device=dev)
        nsw = lib.mdx_smooth_workspace_bytes(B, h, w)
        sws = torch.empty(nsw // 8 + 1, dtype=torch.float64, device=dev)

        def fwd():
            _lib.check(lib.mdx_photometric_fwd(
                C.byref(d), _lib.ptr(disp), _lib.ptr(tgt), C.byref(src), _lib.ptr(invK), _lib.ptr(P),
                _lib.ptr(ident), _lib.ptr(noise), _lib.ptr(idx, torch.uint8), None, None, None,
                _lib.ptr(warp) if a.save_warp in (1, 3) else None, None, _lib.ptr(coef) if a.save_warp >= 2 else None,
                _lib.ptr(ws, torch.float64), C.c_size_t(nws), _lib.stream()), "fwd")

        def bwd():
            _lib.check(lib.mdx_photometric_bwd(
                C.byref(d), _lib.ptr(disp), _lib.ptr(tgt), C.byref(src), _lib.ptr(invK), _lib.ptr(P),
                _lib.ptr(idx, torch.uint8), _lib.ptr(warp) if a.save_warp in (1, 3) else None,
                _lib.ptr(coef) if a.save_warp >= 2 else None, C.c_float(1e-6), None, _lib.ptr(gdisp), _lib.ptr(gP),
                _lib.ptr(ws, torch.float64), C.c_size_t(nws), _lib.stream()), "bwd")

        def ident_fn():
            F.identity_loss(tgt, srcs)

        def smooth():
            _lib.check(lib.mdx_smooth_loss(B, h, w, _lib.ptr(disp), _lib.ptr(color_s), 1, _lib.ptr(loss),
                                           _lib.ptr(gdisp), _lib.ptr(sws, torch.float64), C.c_size_t(nsw),
                                           _lib.stream()), "smooth")
        fwd()
        fns = {"fwd": fwd, "bwd": bwd, "ident": ident_fn, "smooth": smooth}
        out = []
        for name in what:
            if name in ("train", "pre") or (name == "ident" and s):
                continue
            fn = fns[name]
            for _ in range(3):
                fn()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            e1.synchronize()
            out.append("%s %.1f us" % (name, 1e3 * e0.elapsed_time(e1) / a.reps))
        print("scale %d: %s  (masked %.1f%%)" % (s, ", ".join(out), 100.0 * float((idx < S).float().mean())), flush=True)

    if "pre" in what:
        # the step's prologue (what the scales share: csrc/photo_prologue.hip) and the training kernel fed by it
        nsc = a.nscales
        disps = [x.clone().requires_grad_(True) for x in all_disps[:nsc]]
        noises = [torch.randn(B, S, H, W, generator=g).to(dev) for _ in range(nsc)]
        st = F.noise_state(dev, seed=1)

        def timeit(fn):
            for _ in range(3):
                fn()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            e1.synchronize()
            return 1e3 * e0.elapsed_time(e1) / a.reps
        t_drawn = timeit(lambda: F.photometric_prologue(tgt, srcs, nsc, rng=st))
        t_inj = timeit(lambda: F.photometric_prologue(tgt, srcs, nsc, noises=noises))
        t_randn = timeit(lambda: torch.randn((nsc, B, S, H, W), device=dev))
        pre = F.photometric_prologue(tgt, srcs, nsc, noises=noises)
        # the kernels through the C-ABI directly (as the `train` path below): the Python wrapper's per-call work would let
        # the queue run dry between launches and inflate event timings
        td = _lib.make_train_desc(B, H, W, S, [tuple(x.shape[2:]) for x in disps], True, 0.1, 100.0, a.rows)
        idxs = [torch.empty(B, H, W, dtype=torch.uint8, device=dev) for _ in range(nsc)]
        sums = torch.empty(nsc, device=dev)
        gdisps = [torch.empty_like(x) for x in disps]
        gPs = torch.empty(nsc, S, B, 3, 4, device=dev)
        nws = lib.mdx_photometric_train_workspace_bytes(C.byref(td))
        ws = torch.empty(nws // 16 + 1, 2, dtype=torch.float64, device=dev)
        pd, pP = _lib.ptr_array([x.detach() for x in disps]), _lib.ptr_array([P] * nsc)
        pi, pg, pb = _lib.ptr_array(idxs, torch.uint8), _lib.ptr_array(gdisps), _lib.ptr_array(pre["bidfi"])

        def train_pre(grads, hook=None):
            _lib.check(lib.mdx_photometric_train_pre(
                C.byref(td), pd, _lib.ptr(tgt), C.byref(src), _lib.ptr(invK), pP, _lib.ptr(pre["tstat"]), pb, None, pi,
                _lib.ptr(sums), pg if grads else None, _lib.ptr(gPs) if grads else None, None, None,
                _lib.ptr(ws, torch.float64), C.c_size_t(nws), _lib.stream(), C.byref(hook) if hook is not None else None),
                "train_pre")

        def kernel_us(grads):
            hooks = [_lib.Timing(lib.mdx_event_create(), lib.mdx_event_create()) for _ in range(a.reps)]
            for hk in hooks:
                train_pre(grads, hk)
            torch.cuda.synchronize()
            us = []
            for hk in hooks:
                v = C.c_float()
                _lib.check(lib.mdx_event_elapsed_us(C.c_void_p(hk.start), C.c_void_p(hk.stop), C.byref(v)), "elapsed")
                us.append(v.value)
            return sum(us) / len(us)
        t_call = timeit(lambda: train_pre(True))
        t_eval = timeit(lambda: train_pre(False))
        tsum, esum = {"train": (kernel_us(True), a.reps)}, {"eval": (kernel_us(False), a.reps)}
        print("prologue: drawn noise %.1f us, injected noise %.1f us (torch.randn of the %d x [B,S,H,W] maps alone: %.1f us)"
              % (t_drawn, t_inj, nsc, t_randn))
        print("train_pre (%d scales): whole call %.1f us, fused kernel %.1f us; forward-only form: call %.1f us, kernel %.1f us"
              % (nsc, t_call, tsum["train"][0], t_eval, esum["eval"][0]), flush=True)

    if "train" in what:
        nsc = a.nscales
        disps = all_disps[:nsc]
        noises = [torch.randn(B, S, H, W, generator=g).to(dev) for _ in range(nsc)]
        td = _lib.make_train_desc(B, H, W, S, [tuple(x.shape[2:]) for x in disps], True, 0.1, 100.0, a.rows)
        idxs = [torch.empty(B, H, W, dtype=torch.uint8, device=dev) for _ in range(nsc)]
        sums = torch.empty(nsc, device=dev)
        gdisps = [torch.empty_like(x) for x in disps]
        gPs = torch.empty(nsc, S, B, 3, 4, device=dev)
        nws = lib.mdx_photometric_train_workspace_bytes(C.byref(td))
        ws = torch.empty(nws // 16 + 1, 2, dtype=torch.float64, device=dev)
        pd, pP, pn = _lib.ptr_array(disps), _lib.ptr_array([P] * nsc), _lib.ptr_array(noises)
        pi, pg = _lib.ptr_array(idxs, torch.uint8), _lib.ptr_array(gdisps)

        def train(hook=None):
            _lib.check(lib.mdx_photometric_train(
                C.byref(td), pd, _lib.ptr(tgt), C.byref(src), _lib.ptr(invK), pP, _lib.ptr(ident), pn, pi,
                _lib.ptr(sums), pg, _lib.ptr(gPs), None, None, _lib.ptr(ws, torch.float64), C.c_size_t(nws),
                _lib.stream(), C.byref(hook) if hook is not None else None), "train")
        for _ in range(3):
            train()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(a.reps):
            train()
        e1.record()
        e1.synchronize()
        hooks = [_lib.Timing(lib.mdx_event_create(), lib.mdx_event_create()) for _ in range(a.reps)]
        for hk in hooks:
            train(hk)
        torch.cuda.synchronize()
        us = []
        for hk in hooks:
            v = C.c_float()
            _lib.check(lib.mdx_event_elapsed_us(C.c_void_p(hk.start), C.c_void_p(hk.stop), C.byref(v)), "elapsed")
            us.append(v.value)
        if os.environ.get("MDX_TRAIN_STAMPS"):   # diagnostic library build: per-item phase clocks at the workspace's end
            items = int(os.environ["MDX_TRAIN_STAMPS"])
            torch.cuda.synchronize()
            st = ws.view(torch.int64).reshape(-1)[-(items * 8 + 2):-2].reshape(items, 8).cpu().double()
            # the tensor is a little larger than the workspace: find the block by its step-count column
            raw = ws.view(torch.int64).reshape(-1).cpu()
            nb = nws // 8
            st = raw[nb - items * 8:nb].reshape(items, 8).double()
            hw = (raw[nb - items * 8:nb].reshape(items, 8)[:, 5] >> 32)
            steps = (raw[nb - items * 8:nb].reshape(items, 8)[:, 5] & 0xffffffff).double()
            xcc = raw[nb - items * 8:nb].reshape(items, 8)[:, 7] & 0xf
            # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[14:13]
            simd_key = (xcc << 16) | (hw & 0x7f30)
            import collections
            per = collections.Counter(simd_key.tolist())
            busy = collections.defaultdict(float)
            for k, life in zip(simd_key.tolist(), st[:, 4].tolist()):
                busy[k] += life
            cnt = sorted(per.values())
            bz = sorted(busy.values())
            print("   placement: %d distinct SIMDs used; items per SIMD min %d median %d max %d; sum of item lifetimes per SIMD "
                  "(cycles) min %.0f median %.0f max %.0f" % (len(per), cnt[0], cnt[len(cnt) // 2], cnt[-1], bz[0],
                                                              bz[len(bz) // 2], bz[-1]))
            end = st[:, 6] + st[:, 4]
            print("   kernel span by stamps: %.0f ticks of the 100 MHz clock? first start -> last end" % (end.max() - st[:, 6].min()))
            print("stamps: items %d, steps/item %.1f; cycles per step: issue %.0f  sample %.0f  ssim %.0f  grad %.0f ; "
                  "item lifetime %.0f cycles (%.1f us at 100 MHz ticks?)" % (
                      items, steps.mean(), (st[:, 0] / steps).mean(), (st[:, 1] / steps).mean(),
                      (st[:, 2] / steps).mean(), (st[:, 3] / steps).mean(), st[:, 4].mean(), st[:, 4].mean() / 100.0))
            if os.environ.get("MDX_STAMPS_DUMP"):
                import numpy as np
                np.save(os.environ["MDX_STAMPS_DUMP"], raw[nb - items * 8:nb].reshape(items, 8).numpy())
            t0 = st[:, 6] - st[:, 6].min()
            print("   start offsets (ticks): median %.0f max %.0f ; lifetime min %.0f max %.0f" % (
                t0.median(), t0.max(), st[:, 4].min(), st[:, 4].max()))
        masked = [100.0 * float((i < S).float().mean()) for i in idxs]
        print("train (%d scales, rows/chunk %d): whole call %.1f us, fused kernel %.1f us (min %.1f)  masked %s"
              % (nsc, a.rows, 1e3 * e0.elapsed_time(e1) / a.reps, sum(us) / len(us), min(us),
                 " ".join("%.1f%%" % m for m in masked)), flush=True)


if __name__ == "__main__":
    main()
