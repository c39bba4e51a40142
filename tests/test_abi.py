"""CPU: the C-ABI library loads and exports every symbol include/mdx.h declares; host-side argument
validation (no kernel is launched, no GPU needed)."""
import ctypes as C
import importlib
import os
import re

import pytest

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import _lib  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "mdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = header_functions()
    assert len(names) >= 25
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.mdx_version() == 510
    assert lib.mdx_status_string(0) == b"MDX_OK"
    assert lib.mdx_status_string(-3) == b"MDX_ERR_WORKSPACE"


def test_desc_init_constants_and_flags():
    d = _lib.make_desc(12, 192, 640, 96, 320, 2, True, 0.1, 100.0)
    import numpy as np
    assert np.float32(d.disp_a) == np.float32(0.01) and np.float32(d.disp_b) == np.float32(9.99)
    assert d.flags & 1 and not d.flags & 2          # automask, generic upsample kernel (H+W > 128)
    assert d.flags & 4 and d.flags & 8              # 639 and 191 are verified constant divisors
    small = _lib.make_desc(2, 24, 40, 12, 20, 2, False, 0.1, 100.0)
    assert not small.flags & 1 and small.flags & 2  # ATen's small-output kernel (H+W <= 128)
    ev = _lib.make_desc(1, 192, 640, 192, 640, 1, False, 1e-3, 80)
    assert np.float32(ev.disp_a) == np.float32(0.0125) and np.float32(ev.disp_b) == np.float32(999.9875)


@pytest.mark.parametrize("args", [(0, 192, 640, 192, 640, 2), (1, 2, 640, 2, 640, 2), (1, 192, 640, 192, 640, 0),
                                  (1, 192, 640, 192, 640, 5), (1, 192, 640, 384, 640, 2)])
def test_desc_init_rejects_bad_shapes(args):
    d = _lib.Desc()
    rc = _lib.lib().mdx_desc_init(C.byref(d), *args, 1, C.c_double(0.1), C.c_double(100.0))
    assert rc == -1   # MDX_ERR_BAD_SHAPE


def test_null_pointers_and_workspace_are_reported_not_crashed():
    lib = _lib.lib()
    d = _lib.make_desc(1, 32, 64, 32, 64, 2, True, 0.1, 100.0)
    assert lib.mdx_photometric_workspace_bytes(C.byref(d)) > 0
    assert lib.mdx_smooth_workspace_bytes(1, 32, 64) > 0
    assert lib.mdx_project_workspace_bytes(1, 32, 64) > 0
    assert lib.mdx_photometric_fwd(C.byref(d), None, None, None, None, None, None, None, None, None, None, None,
                                   None, None, None, C.c_size_t(0), None) == -2
    assert lib.mdx_compose_projection(None, None, 1, None, None) == -2
    assert lib.mdx_smooth_loss(1, 32, 64, None, None, 1, None, None, None, C.c_size_t(0), None) == -2
    with pytest.raises(_lib.MdxError):
        _lib.check(-4, "unit test")


def test_round3_entry_points_report_bad_arguments():
    """mdx_photometric_prologue / mdx_photometric_train_pre / mdx_ssim_bwd / mdx_to_tensor_u8: argument errors come back
    as status codes (host-side checks only: nothing is launched, no GPU needed)."""
    lib = _lib.lib()
    td = _lib.make_train_desc(1, 32, 64, 2, [(32, 64), (16, 32)], True, 0.1, 100.0)
    assert lib.mdx_photometric_train_workspace_bytes(C.byref(td)) > 0
    assert lib.mdx_photometric_prologue(None, None, None, None, None, 0, None, None, None, None) == -2
    assert lib.mdx_photometric_prologue(C.byref(td), None, None, None, None, 0, None, None, None, None) == -2
    bad = _lib.TrainDesc()
    C.memmove(C.byref(bad), C.byref(td), C.sizeof(td))
    bad.nscales = 9
    fake = C.c_void_p(4096)          # a non-null, 16-byte aligned address that is never dereferenced on these paths
    assert lib.mdx_photometric_prologue(C.byref(bad), fake, None, None, None, 0, None, fake, None, None) == -1
    # auto-mask on: sources / outputs per scale / a noise source are required
    assert lib.mdx_photometric_prologue(C.byref(td), fake, None, None, None, 0, None, fake, None, None) == -2
    assert lib.mdx_photometric_train_pre(C.byref(td), None, None, None, None, None, None, None, None, None, None, None,
                                         None, None, None, None, C.c_size_t(0), None, None) == -2
    assert lib.mdx_ssim_bwd(None, None, None, 3, 8, 8, None, None, None) == -2
    assert lib.mdx_ssim_bwd(fake, fake, fake, 3, 2, 8, fake, None, None) == -1
    assert lib.mdx_to_tensor_u8(None, None, C.c_size_t(16), None) == -2
    assert lib.mdx_to_tensor_u8(fake, fake, C.c_size_t(0), None) == -1
    assert lib.mdx_to_tensor_u8(fake, C.c_void_p(4100), C.c_size_t(16), None) == -6      # destination not 16-byte aligned


def test_round4_workspace_contracts():
    """Host-side size functions of the entries whose insides changed in round 4 (no GPU needed).  mdx_photometric_train: at
    S >= 3 the workspace also holds the items' (u, v) rings -- 3 rows x S planes x 64 lanes x 8 bytes per work item -- and a
    caller that hands over less is told so (MDX_ERR_WORKSPACE) before anything is launched.  mdx_smooth_loss(_multi): four
    doubles per block of the main pass and image."""
    lib = _lib.lib()
    hw = [(192, 640), (96, 320), (48, 160), (24, 80)]
    w2 = lib.mdx_photometric_train_workspace_bytes(C.byref(_lib.make_train_desc(12, 192, 640, 2, hw, True, 0.1, 100.0)))
    w3 = lib.mdx_photometric_train_workspace_bytes(C.byref(_lib.make_train_desc(12, 192, 640, 3, hw, True, 0.1, 100.0)))
    items = 4 * 12 * 11 * 7          # scales x images x strips of 60 columns x chunks of a 192-row column
    gup = 4 * 12 * 192 * 640 * 4     # the full-resolution gradient maps of the scales
    ring = items * 3 * 3 * 512
    assert gup + ring <= w3 < gup + ring + (1 << 20), (w3, gup, ring)
    assert gup <= w2 < gup + (2 << 20), (w2, gup)          # S = 2: the register form, no ring
    td = _lib.make_train_desc(1, 32, 64, 3, [(32, 64)], True, 0.1, 100.0)
    fake = C.c_void_p(4096)
    arr = (C.c_void_p * 1)(4096)
    src = _lib.Sources()
    for f in range(3):
        src.img[f] = 4096
    need = lib.mdx_photometric_train_workspace_bytes(C.byref(td))
    rc = lib.mdx_photometric_train_pre(C.byref(td), arr, fake, C.byref(src), fake, arr, fake, arr, None, arr, fake, arr, fake,
                                       None, None, fake, C.c_size_t(need - 1), None, None)
    assert rc == -3, rc                                   # MDX_ERR_WORKSPACE
    one = lib.mdx_smooth_workspace_bytes(12, 192, 640)
    assert one == 12 * ((192 * 640 + 255) // 256) * 4 * 8
    hs, ws = (C.c_int32 * 4)(192, 96, 48, 24), (C.c_int32 * 4)(640, 320, 160, 80)
    assert lib.mdx_smooth_multi_workspace_bytes(4, 12, hs, ws) >= one
    assert lib.mdx_smooth_loss_multi(4, 12, hs, ws, arr, arr, 1, fake, None, fake, C.c_size_t(one), None) in (-2, -3)


def test_kernel_entry_points_refuse_cpu_tensors_loudly():
    """The kernel layer (mdx.functional -> libmdx_hip.so) never computes anything on the host: a CPU tensor raises.  (The
    reference-named ops one level up, model_layer / model_loss, dispatch CPU tensors to the package's plain-PyTorch composite --
    by the tensor's device, tests/test_cpu_dispatch.py -- and GPU tensors to these entry points, with no way back.)"""
    import torch
    from mdx import functional as F
    with pytest.raises(_lib.MdxError):
        F.reprojection_loss(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
    with pytest.raises(_lib.MdxError):
        F.disparity2depth(torch.rand(1, 1, 8, 8), 0.1, 100)
    with pytest.raises(_lib.MdxError):
        F.bn_act(torch.rand(2, 8, 4, 4), torch.ones(8), torch.zeros(8), None, None)


def test_graft_entry_build():
    """the driver's "does it build" entry point: compiles what is stale (nothing, after this module's own import), builds the oracle,
    checks the library's version against the header's -- it asserted a stale MDX_VERSION once"""
    import re
    import __graft_entry__ as g
    g.build()
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mdx.h")).read()
    want = int(re.search(r"#define\s+MDX_VERSION\s+(\d+)", header).group(1))
    from mdx import _lib
    assert _lib.lib().mdx_version() == want
