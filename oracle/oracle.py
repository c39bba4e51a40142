"""ctypes binding of the CPU oracle (oracle/mdx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package.  Arrays are numpy float32, contiguous, NCHW.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MDX_ORACLE_SO") or os.path.join(_HERE, "libmdx_oracle.so")   # MDX_ORACLE_SO: sanitizer build
_lib = None

_f = C.POINTER(C.c_float)
_u8 = C.POINTER(C.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "mdx_oracle.c")
    if os.environ.get("MDX_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_photometric_fwd.restype = C.c_double
        _lib.orc_min_automask.restype = C.c_double
        _lib.orc_smooth_loss.restype = C.c_double
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f)


def _pu8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def upsample_bilinear(x, H, W):
    x = f32(x)
    B, Cc, h, w = x.shape
    out = np.empty((B, Cc, H, W), np.float32)
    lib().orc_upsample_bilinear(_p(x), B * Cc, h, w, _p(out), H, W)
    return out


def upsample_bilinear_bwd(gout, h, w):
    gout = f32(gout)
    B, Cc, H, W = gout.shape
    gin = np.empty((B, Cc, h, w), np.float32)
    lib().orc_upsample_bilinear_bwd(_p(gout), B * Cc, H, W, _p(gin), h, w)
    return gin


def disparity2depth(disp, min_depth, max_depth):
    disp = f32(disp)
    sd, dep = np.empty_like(disp), np.empty_like(disp)
    lib().orc_disparity2depth(_p(disp), C.c_size_t(disp.size), C.c_double(min_depth),
                              C.c_double(max_depth), _p(sd), _p(dep))
    return sd, dep


def compose_projection(K, T):
    K, T = f32(K), f32(T)
    B = K.shape[0]
    P = np.empty((B, 3, 4), np.float32)
    lib().orc_compose_projection(_p(K), _p(T), B, _p(P))
    return P


def compose_projection_bwd(K, gP):
    K, gP = f32(K), f32(gP)
    B = K.shape[0]
    gT = np.empty((B, 4, 4), np.float32)
    lib().orc_compose_projection_bwd(_p(K), _p(gP), B, _p(gT))
    return gT


def backproject(depth, invK):
    depth, invK = f32(depth), f32(invK)
    B, _, H, W = depth.shape
    cam = np.empty((B, 4, H * W), np.float32)
    lib().orc_backproject(_p(depth), _p(invK), B, H, W, _p(cam))
    return cam


def project(cam, P, H, W):
    cam, P = f32(cam), f32(P)
    B = cam.shape[0]
    grid = np.empty((B, H, W, 2), np.float32)
    lib().orc_project(_p(cam), _p(P), B, H, W, _p(grid))
    return grid


def grid_sample(img, grid):
    img, grid = f32(img), f32(grid)
    B, Cc, Hi, Wi = img.shape
    _, Ho, Wo, _ = grid.shape
    out = np.empty((B, Cc, Ho, Wo), np.float32)
    lib().orc_grid_sample(_p(img), _p(grid), B, Cc, Hi, Wi, Ho, Wo, _p(out))
    return out


def grid_sample_bwd(img, grid, gout, need_img=False):
    img, grid, gout = f32(img), f32(grid), f32(gout)
    B, Cc, Hi, Wi = img.shape
    _, Ho, Wo, _ = grid.shape
    gg = np.empty_like(grid)
    lib().orc_grid_sample_bwd_grid(_p(img), _p(grid), _p(gout), B, Cc, Hi, Wi, Ho, Wo, _p(gg))
    if not need_img:
        return gg
    gi = np.empty_like(img)
    lib().orc_grid_sample_bwd_img(_p(grid), _p(gout), B, Cc, Hi, Wi, Ho, Wo, _p(gi))
    return gg, gi


def ssim(x, y):
    x, y = f32(x), f32(y)
    B, Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().orc_ssim(_p(x), _p(y), B * Cc, H, W, _p(out))
    return out


def reprojection_loss(pred, target):
    pred, target = f32(pred), f32(target)
    B, _, H, W = pred.shape
    out = np.empty((B, 1, H, W), np.float32)
    lib().orc_reprojection_loss(_p(pred), _p(target), B, H, W, _p(out))
    return out


def reprojection_loss_bwd(pred, target, gout, need_target=False):
    pred, target, gout = f32(pred), f32(target), f32(gout)
    B, _, H, W = pred.shape
    gp = np.empty_like(pred)
    gt = np.empty_like(pred) if need_target else None
    lib().orc_reprojection_loss_bwd(_p(pred), _p(target), _p(gout), B, H, W, _p(gp), _p(gt))
    return (gp, gt) if need_target else gp


def min_automask(ident, noise, reproj, automask=True):
    reproj = f32(reproj)
    B, S, H, W = reproj.shape
    ident = f32(ident) if ident is not None else None
    noise = f32(noise) if noise is not None else None
    Cc = 2 * S if automask else S
    comb = np.empty((B, Cc, H, W), np.float32)
    to_opt = np.empty((B, H, W), np.float32)
    idx = np.empty((B, H, W), np.uint8)
    tot = lib().orc_min_automask(_p(ident), _p(noise), _p(reproj), B, S, H, W, int(automask),
                                 _p(comb), _p(to_opt), _pu8(idx))
    return comb, to_opt, idx, tot


def smooth_loss(disp, color, need_grad=False):
    disp, color = f32(disp), f32(color)
    B, _, h, w = disp.shape
    g = np.empty_like(disp) if need_grad else None
    v = lib().orc_smooth_loss(_p(disp), _p(color), B, h, w, _p(g))
    return (v, g) if need_grad else v


def _srcs(sources):
    sources = [f32(s) for s in sources]
    arr = (_f * len(sources))(*[_p(s) for s in sources])
    return sources, arr


def photometric_fwd(disp, target, sources, invK, P, noise, min_depth=0.1, max_depth=100.0,
                    automask=True, full=False):
    """One scale of processor.py:139-217.  P: [S,B,3,4].  Returns dict."""
    disp, target, invK, P = f32(disp), f32(target), f32(invK), f32(P)
    B, _, h, w = disp.shape
    _, _, H, W = target.shape
    sources, sarr = _srcs(sources)
    S = len(sources)
    noise = f32(noise) if noise is not None else None
    Cc = 2 * S if automask else S
    out = {"to_opt": np.empty((B, H, W), np.float32), "idx": np.zeros((B, H, W), np.uint8),
           "depth": np.empty((B, 1, H, W), np.float32)}
    if full:
        out.update(grid=np.empty((S, B, H, W, 2), np.float32), warp=np.empty((S, B, 3, H, W), np.float32),
                   reproj=np.empty((B, S, H, W), np.float32), ident=np.empty((B, S, H, W), np.float32),
                   combined=np.empty((B, Cc, H, W), np.float32))
    out["sum"] = lib().orc_photometric_fwd(
        B, H, W, h, w, S, C.c_double(min_depth), C.c_double(max_depth), int(automask), _p(disp),
        _p(target), sarr, _p(invK), _p(P), _p(noise), _p(out["depth"]), _p(out.get("grid")),
        _p(out.get("warp")), _p(out.get("reproj")), _p(out.get("ident")), _p(out.get("combined")),
        _p(out["to_opt"]), _pu8(out["idx"]))
    return out


_lib64 = None


def photometric_bwd_f64(disp, target, sources, invK, P, idx, g_min, min_depth=0.1, max_depth=100.0, automask=True):
    """orc_photometric_bwd evaluated in float64 (oracle/Makefile target f64: the same C source with float = double) on
    the float32 inputs and the given arg-min indices: the exact gradient to ~1e-15, against which the tests measure the
    rounding error of the float32 evaluations (the oracle's and the GPU's) entry by entry.  -> (gdisp, gP) float64."""
    global _lib64
    if _lib64 is None:
        so = os.path.join(_HERE, "libmdx_oracle_f64.so")
        src = os.path.join(_HERE, "mdx_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "f64"])
        _lib64 = C.CDLL(so)
    d = C.POINTER(C.c_double)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)     # noqa: E731
    disp, target, invK, P = f64(disp), f64(target), f64(invK), f64(P)
    srcs = [f64(s) for s in sources]
    B, _, h, w = disp.shape
    _, _, H, W = target.shape
    S = len(srcs)
    sarr = (d * S)(*[s.ctypes.data_as(d) for s in srcs])
    gdisp, gP = np.empty_like(disp), np.empty((S, B, 3, 4), np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.uint8)
    _lib64.orc_photometric_bwd(B, H, W, h, w, S, C.c_double(min_depth), C.c_double(max_depth), int(automask),
                               disp.ctypes.data_as(d), target.ctypes.data_as(d), sarr, invK.ctypes.data_as(d),
                               P.ctypes.data_as(d), _pu8(idx), C.c_double(g_min), gdisp.ctypes.data_as(d),
                               gP.ctypes.data_as(d))
    return gdisp, gP


def photometric_bwd(disp, target, sources, invK, P, idx, g_min, min_depth=0.1, max_depth=100.0,
                    automask=True):
    disp, target, invK, P = f32(disp), f32(target), f32(invK), f32(P)
    B, _, h, w = disp.shape
    _, _, H, W = target.shape
    sources, sarr = _srcs(sources)
    S = len(sources)
    gdisp = np.empty_like(disp)
    gP = np.empty((S, B, 3, 4), np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.uint8)
    lib().orc_photometric_bwd(B, H, W, h, w, S, C.c_double(min_depth), C.c_double(max_depth),
                              int(automask), _p(disp), _p(target), sarr, _p(invK), _p(P), _pu8(idx),
                              C.c_double(g_min), _p(gdisp), _p(gP))
    return gdisp, gP
