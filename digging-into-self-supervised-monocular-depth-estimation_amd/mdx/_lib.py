"""ctypes binding of libmdx_hip.so (include/mdx.h).  No torch types cross this boundary: tensors are
handed over as raw device pointers + sizes, the current HIP stream as a void*.

There is NO fallback: if the library is missing or a tensor is not a contiguous float32 CUDA/HIP
tensor, the call raises.
"""
import ctypes as C
import os

import torch

LIB_PATH = os.environ.get("MDX_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "libmdx_hip.so")   # MDX_LIB: developer override (kernel A/B builds)
MAX_SRC = 4
MAX_SCALES = 4
FLAG_AUTOMASK = 1
_lib = None


class MdxError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("S", C.c_int32), ("flags", C.c_uint32), ("disp_a", C.c_float), ("disp_b", C.c_float)]


class Sources(C.Structure):
    _fields_ = [("img", C.c_void_p * MAX_SRC)]


class TrainDesc(C.Structure):   # mdx_train_desc
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("S", C.c_int32), ("nscales", C.c_int32),
                ("flags", C.c_uint32), ("disp_a", C.c_float), ("disp_b", C.c_float),
                ("h", C.c_int32 * MAX_SCALES), ("w", C.c_int32 * MAX_SCALES), ("rows_per_chunk", C.c_int32)]


class Timing(C.Structure):   # mdx_timing: hipEvent_t pair recorded right before / after the fused kernel
    _fields_ = [("start", C.c_void_p), ("stop", C.c_void_p)]


# every symbol include/mdx.h declares: name -> restype (None = int status)
SYMBOLS = {
    "mdx_version": C.c_int, "mdx_status_string": C.c_char_p, "mdx_desc_init": C.c_int,
    "mdx_compose_projection": C.c_int, "mdx_identity_loss": C.c_int,
    "mdx_photometric_workspace_bytes": C.c_size_t, "mdx_photometric_fwd": C.c_int,
    "mdx_photometric_bwd": C.c_int, "mdx_smooth_workspace_bytes": C.c_size_t, "mdx_smooth_loss": C.c_int,
    "mdx_interpolate_bilinear_fwd": C.c_int, "mdx_interpolate_bilinear_bwd": C.c_int,
    "mdx_disparity2depth_fwd": C.c_int, "mdx_disparity2depth_bwd": C.c_int,
    "mdx_backproject_fwd": C.c_int, "mdx_backproject_bwd": C.c_int,
    "mdx_project_workspace_bytes": C.c_size_t, "mdx_project_fwd": C.c_int, "mdx_project_bwd": C.c_int,
    "mdx_grid_sample_border_fwd": C.c_int, "mdx_grid_sample_border_bwd": C.c_int,
    "mdx_reprojection_loss_fwd": C.c_int, "mdx_reprojection_loss_bwd": C.c_int, "mdx_ssim_fwd": C.c_int, "mdx_ssim_bwd": C.c_int,
    "mdx_min_automask_fwd": C.c_int,
    "mdx_loss_total_fwd": C.c_int, "mdx_loss_total_bwd": C.c_int,
    "mdx_adam_max_tensors": C.c_int, "mdx_adam_chunk": C.c_int, "mdx_adam_table_entry_bytes": C.c_size_t, "mdx_adam_step": C.c_int,
    "mdx_pose_projection_fwd": C.c_int, "mdx_pose_projection_bwd": C.c_int,
    "mdx_bias_act_nhwc_workspace_bytes": C.c_size_t, "mdx_bias_act_nhwc_fwd": C.c_int, "mdx_bias_act_nhwc_bwd": C.c_int,
    "mdx_mean_bias_nhwc_fwd": C.c_int, "mdx_mean_bias_nhwc_bwd": C.c_int, "mdx_encoder_input_nhwc": C.c_int,
    "mdx_thin_conv3x3_wgrad_workspace_bytes": C.c_size_t, "mdx_thin_conv3x3_wgrad": C.c_int,
    "mdx_disp_head_nhwc_workspace_bytes": C.c_size_t, "mdx_disp_head_nhwc_fwd": C.c_int, "mdx_disp_head_nhwc_bwd": C.c_int,
    "mdx_event_create": C.c_void_p, "mdx_event_destroy": None, "mdx_event_elapsed_us": C.c_int,
    "mdx_photometric_fwd_timed": C.c_int, "mdx_photometric_bwd_timed": C.c_int,
    "mdx_decoder_glue_fwd": C.c_int, "mdx_decoder_glue_bwd": C.c_int, "mdx_decoder_glue_workspace_bytes": C.c_size_t,
    "mdx_maxpool3s2_fwd": C.c_int, "mdx_maxpool3s2_bwd": C.c_int,
    "mdx_bn_workspace_bytes": C.c_size_t, "mdx_bn_act_fwd": C.c_int, "mdx_bn_act_bwd": C.c_int,
    "mdx_bn_nhwc_workspace_bytes": C.c_size_t, "mdx_bn_act_nhwc_fwd": C.c_int, "mdx_bn_act_nhwc_bwd": C.c_int,
    "mdx_decoder_glue_nhwc_fwd": C.c_int, "mdx_decoder_glue_nhwc_bwd": C.c_int,
    "mdx_decoder_glue_nhwc_workspace_bytes": C.c_size_t, "mdx_maxpool3s2_nhwc_fwd": C.c_int, "mdx_maxpool3s2_nhwc_bwd": C.c_int,
    "mdx_param2matrix_fwd": C.c_int, "mdx_param2matrix_bwd": C.c_int,
    "mdx_train_desc_init": C.c_int, "mdx_photometric_train_workspace_bytes": C.c_size_t,
    "mdx_photometric_train": C.c_int, "mdx_photometric_prologue": C.c_int, "mdx_photometric_train_pre": C.c_int,
    "mdx_smooth_multi_workspace_bytes": C.c_size_t, "mdx_smooth_loss_multi": C.c_int,
    "mdx_depth_monitor_workspace_bytes": C.c_size_t, "mdx_depth_monitor": C.c_int,
    "mdx_resample_ksize": C.c_int, "mdx_resample_plan": C.c_int, "mdx_resample_plan_cols": C.c_int,
    "mdx_resample_lanczos_u8": C.c_int,
    "mdx_color_jitter_u8": C.c_int, "mdx_color_convert_u8": C.c_int, "mdx_to_tensor_u8": C.c_int,
}


def lib():
    """Loads libmdx_hip.so; raises MdxError loudly if it is absent or incomplete."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and not os.environ.get("MDX_LIB"):
            # source-only checkout on a machine with the ROCm toolchain: compile the kernels now (this is the
            # product path building itself, not a fallback -- without hipcc the error below stands)
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location(
                    "_mdx_build", os.path.join(os.path.dirname(LIB_PATH), "build.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.build()
            except Exception as exc:  # noqa: BLE001
                raise MdxError("libmdx_hip.so is missing and could not be built with hipcc: %s" % exc)
        if not os.path.exists(LIB_PATH):
            raise MdxError("libmdx_hip.so not found at %s -- build it with "
                           "`python __graft_entry__.py build` (hipcc --offload-arch=gfx950); "
                           "there is no CPU/eager fallback" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, res in SYMBOLS.items():
            try:
                getattr(handle, name).restype = res
            except AttributeError:
                raise MdxError("libmdx_hip.so does not export %s (stale build?)" % name)
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        raise MdxError("%s failed: %s (%d)" % (what, lib().mdx_status_string(status).decode(), status))


def ptr(t, dtype=torch.float32, optional=False, cl=False):
    """Raw device pointer of a contiguous CUDA/HIP tensor (cl: of a 4-D tensor whose memory is channels-last, [B][H][W][C];
    cl="any": dense in either of the two formats -- the caller hands the strides to the kernel)."""
    if t is None:
        if optional:
            return None
        raise MdxError("required tensor is None")
    if not t.is_cuda:
        raise MdxError("mdx kernels run on the GPU only: got a %s tensor (no CPU fallback)" % t.device)
    if cl == "any":
        if t.dtype != dtype or t.dim() != 4 or not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
            raise MdxError("expected a dense %s 4-D tensor, got %s %s strides %s" % (dtype, t.dtype, tuple(t.shape), t.stride()))
    elif cl:
        if t.dtype != dtype or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
            raise MdxError("expected a channels-last %s map, got %s %s strides %s" % (dtype, t.dtype, tuple(t.shape), t.stride()))
    elif t.dtype != dtype or not t.is_contiguous():
        raise MdxError("expected contiguous %s, got %s contiguous=%s" % (dtype, t.dtype, t.is_contiguous()))
    if t.device.index != torch.cuda.current_device():
        # stream() hands the CURRENT device's stream to the library: kernels would run on that device with another
        # device's pointers (a GPU memory fault, not an error code)
        raise MdxError("tensor lives on %s but the current device is cuda:%d -- call torch.cuda.set_device / "
                       "`with torch.cuda.device(...)` first" % (t.device, torch.cuda.current_device()))
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_desc(B, H, W, h, w, S, automask, min_depth, max_depth):
    d = Desc()
    check(lib().mdx_desc_init(C.byref(d), B, H, W, h, w, S, int(bool(automask)), C.c_double(min_depth),
                              C.c_double(max_depth)), "mdx_desc_init")
    return d


def make_train_desc(B, H, W, S, hw, automask, min_depth, max_depth, rows_per_chunk=0):
    """hw: [(h_s, w_s)] per scale."""
    if not 1 <= len(hw) <= MAX_SCALES:
        raise MdxError("1..%d scales supported, got %d" % (MAX_SCALES, len(hw)))
    d = TrainDesc()
    hs = (C.c_int32 * len(hw))(*[int(x[0]) for x in hw])
    ws = (C.c_int32 * len(hw))(*[int(x[1]) for x in hw])
    check(lib().mdx_train_desc_init(C.byref(d), B, H, W, S, len(hw), hs, ws, int(bool(automask)),
                                    C.c_double(min_depth), C.c_double(max_depth), int(rows_per_chunk)),
          "mdx_train_desc_init")
    return d


def ptr_array(tensors, dtype=torch.float32, optional=False):
    """Host array of device pointers (one per scale); None entries allowed when optional."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        p = ptr(t, dtype, optional=optional)
        arr[i] = p.value if p is not None else None
    return arr


def make_sources(tensors):
    s = Sources()
    if not 1 <= len(tensors) <= MAX_SRC:
        raise MdxError("1..%d source frames supported, got %d" % (MAX_SRC, len(tensors)))
    for i, t in enumerate(tensors):
        s.img[i] = ptr(t).value
    return s
