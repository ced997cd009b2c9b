"""ctypes binding of libframewright_hip.so (C-ABI declared in include/framewright_hip.h).

The library is the product: if it is missing or does not load, every operator raises — there is no eager /
CPU fallback on this path.
"""
from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path
from typing import Optional

PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = PKG_DIR / "lib" / "libframewright_hip.so"

FW_OK, FW_ERR_INVALID, FW_ERR_OOM, FW_ERR_HIP, FW_ERR_INTERNAL = 0, 1, 2, 3, 4
FW_HOST, FW_DEVICE = 0, 1
FW_DTYPE_BF16, FW_DTYPE_F16 = 0, 1
DTYPES = {"bf16": FW_DTYPE_BF16, "bfloat16": FW_DTYPE_BF16, "f16": FW_DTYPE_F16, "fp16": FW_DTYPE_F16,
          "float16": FW_DTYPE_F16, "half": FW_DTYPE_F16}

# every symbol include/framewright_hip.h declares (tests/test_cabi.py checks the header against this list)
EXPORTS = [
    "fw_last_error", "fw_abi_version", "fw_device_count",
    "fw_rrdbnet_create", "fw_rrdbnet_set_conv", "fw_rrdbnet_finalize", "fw_rrdbnet_upscale_u8", "fw_rrdbnet_upscale_u16",
    "fw_rrdbnet_workspace_bytes", "fw_rrdbnet_flops", "fw_rrdbnet_profile_enable", "fw_rrdbnet_profile_read",
    "fw_rrdbnet_destroy", "fw_pack_conv3x3", "fw_conv3x3_nhwc",
    "fw_nafnet_create", "fw_nafnet_set_tensor", "fw_nafnet_finalize", "fw_nafnet_denoise_u8", "fw_nafnet_flops",
    "fw_nafnet_destroy", "fw_u8_crop", "fw_tile_blend_accumulate", "fw_tile_blend_finish", "fw_temporal_average_u8",
    "fw_strength_blend_u8",
    "fw_conv3x3_nhwc_ex", "fw_conv3x3_pair_nhwc", "fw_pack_conv_up2x_phase", "fw_conv_up2x_phase_nhwc", "fw_u8_to_rgb_f32", "fw_resize_bilinear_f32", "fw_ifnet_build_x", "fw_unshuffle2_cast",
    "fw_depth_to_space4_f32", "fw_ifnet_accumulate", "fw_ifnet_blend", "fw_unsharp_mask_u8",
    "fw_ifnet_create", "fw_ifnet_set_tensor", "fw_ifnet_finalize", "fw_ifnet_interp_u8", "fw_ifnet_workspace_bytes", "fw_ifnet_flops",
    "fw_ifnet_destroy",
    "fw_aesrgan_create", "fw_aesrgan_set_tensor", "fw_aesrgan_finalize", "fw_aesrgan_forward_rgb", "fw_aesrgan_workspace_bytes", "fw_aesrgan_destroy",
    "fw_srvgg_create", "fw_srvgg_set_tensor", "fw_srvgg_finalize", "fw_srvgg_upscale_u8", "fw_srvgg_upscale_u16", "fw_resize_lanczos4_u16", "fw_srvgg_workspace_bytes", "fw_srvgg_flops",
    "fw_srvgg_destroy",
    "fw_restormer_create", "fw_restormer_set_tensor", "fw_restormer_finalize", "fw_restormer_denoise_u8",
    "fw_restormer_workspace_bytes", "fw_restormer_destroy", "fw_preserve_edges_scratch_bytes", "fw_preserve_edges_u8",
    "fw_u8_to_nhwc", "fw_pixel_shuffle_add_u8",
    "fw_layernorm_nhwc", "fw_pack_pointwise", "fw_pointwise_nhwc", "fw_dwconv3x3_nhwc", "fw_attn_workspace_floats",
    "fw_attn_matrix", "fw_attn_apply", "fw_attn_pack", "fw_attn_proj_pack", "fw_pixel_shuffle2_f32", "fw_copy_channels_f32", "fw_f32_to_planar", "fw_tap_post_u8",
    "fw_flow_accumulate_u8", "fw_flow_accumulate_finish_u8", "fw_resize_lanczos4_u8", "fw_resize_linear_u8", "fw_face_paste_u8", "fw_grain_addback_u8", "fw_attn_softmax_rows", "fw_pack_pointwise_transposed", "fw_attn_qk_scratch_elems", "fw_attn_matrix_mfma",
]


class FramewrightHipError(RuntimeError):
    """Raised for every failing C-ABI call; ``code`` is the FW_ERR_* status."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class FramewrightOutOfMemory(FramewrightHipError):
    """FW_ERR_OOM — message contains "memory" so reference restorer.py:1746 retries with a smaller tile."""


_lock = threading.Lock()
_lib: Optional[C.CDLL] = None


def _declare(lib: C.CDLL) -> None:
    vp, i32, f32, f64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    lib.fw_last_error.restype = C.c_char_p
    lib.fw_last_error.argtypes = []
    lib.fw_abi_version.restype = i32
    lib.fw_abi_version.argtypes = []
    lib.fw_device_count.restype = i32
    lib.fw_device_count.argtypes = []
    lib.fw_rrdbnet_create.restype = i32
    lib.fw_rrdbnet_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    lib.fw_rrdbnet_set_conv.restype = i32
    lib.fw_rrdbnet_set_conv.argtypes = [vp, C.c_char_p, vp, vp, i32, i32]
    lib.fw_rrdbnet_finalize.restype = i32
    lib.fw_rrdbnet_finalize.argtypes = [vp]
    lib.fw_rrdbnet_upscale_u8.restype = i32
    lib.fw_rrdbnet_upscale_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_rrdbnet_upscale_u16.restype = i32
    lib.fw_rrdbnet_upscale_u16.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_rrdbnet_workspace_bytes.restype = sz
    lib.fw_rrdbnet_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fw_rrdbnet_flops.restype = f64
    lib.fw_rrdbnet_flops.argtypes = [vp, i32, i32]
    lib.fw_rrdbnet_profile_enable.restype = i32
    lib.fw_rrdbnet_profile_enable.argtypes = [vp, i32]
    lib.fw_rrdbnet_profile_read.restype = i32
    lib.fw_rrdbnet_profile_read.argtypes = [vp, C.POINTER(i32), C.POINTER(f64), C.POINTER(f64)]
    lib.fw_rrdbnet_destroy.restype = i32
    lib.fw_rrdbnet_destroy.argtypes = [vp]
    lib.fw_pack_conv3x3.restype = sz
    lib.fw_pack_conv3x3.argtypes = [i32, vp, i32, i32, i32, i32, vp]
    lib.fw_conv3x3_nhwc.restype = i32
    lib.fw_conv3x3_nhwc.argtypes = [i32, vp, i32, C.c_long, i32, i32, i32, vp, vp, i32, i32, i32, vp, f32, vp, f32, vp,
                                    i32, C.c_long, i32, vp, vp]


def _declare_tap(lib: C.CDLL) -> None:
    vp, i32, f32, f64, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    lib.fw_nafnet_create.restype = i32
    lib.fw_nafnet_create.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), i32, i32, C.POINTER(vp)]
    lib.fw_nafnet_set_tensor.restype = i32
    lib.fw_nafnet_set_tensor.argtypes = [vp, C.c_char_p, vp, sz]
    lib.fw_nafnet_finalize.restype = i32
    lib.fw_nafnet_finalize.argtypes = [vp]
    lib.fw_nafnet_denoise_u8.restype = i32
    lib.fw_nafnet_denoise_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_nafnet_flops.restype = f64
    lib.fw_nafnet_flops.argtypes = [vp, i32, i32]
    lib.fw_nafnet_destroy.restype = i32
    lib.fw_nafnet_destroy.argtypes = [vp]
    lib.fw_u8_crop.restype = i32
    lib.fw_u8_crop.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.fw_tile_blend_accumulate.restype = i32
    lib.fw_tile_blend_accumulate.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp]
    lib.fw_tile_blend_finish.restype = i32
    lib.fw_tile_blend_finish.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.fw_temporal_average_u8.restype = i32
    lib.fw_temporal_average_u8.argtypes = [C.POINTER(vp), C.POINTER(f32), i32, sz, vp, vp]
    lib.fw_strength_blend_u8.restype = i32
    lib.fw_strength_blend_u8.argtypes = [vp, vp, f64, sz, vp, vp]
    lib.fw_grain_addback_u8.restype = i32
    lib.fw_grain_addback_u8.argtypes = [vp, vp, i32, i32, f64, vp, vp, vp]
    lib.fw_resize_lanczos4_u8.restype = i32
    lib.fw_resize_lanczos4_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp]
    lib.fw_resize_linear_u8.restype = i32
    lib.fw_resize_linear_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp]
    lib.fw_face_paste_u8.restype = i32
    lib.fw_face_paste_u8.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, f64, vp]


def _declare_ifnet(lib: C.CDLL) -> None:
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.fw_conv3x3_nhwc_ex.restype = i32
    lib.fw_conv3x3_nhwc_ex.argtypes = [i32, vp, i32, C.c_long, i32, i32, i32, vp, vp, i32, i32, i32, vp, f32, vp, f32, vp, i32,
                                       i32, i32, vp, i32, C.c_long, i32, vp, vp]
    lib.fw_pack_conv_up2x_phase.restype = sz
    lib.fw_pack_conv_up2x_phase.argtypes = [i32, vp, vp]
    lib.fw_conv_up2x_phase_nhwc.restype = i32
    lib.fw_conv_up2x_phase_nhwc.argtypes = [i32, vp, i32, C.c_long, i32, i32, vp, vp, i32, vp, i32, C.c_long, vp]
    lib.fw_conv3x3_pair_nhwc.restype = i32
    lib.fw_conv3x3_pair_nhwc.argtypes = [i32, vp, i32, C.c_long, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.fw_u8_to_rgb_f32.restype = i32
    lib.fw_u8_to_rgb_f32.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.fw_resize_bilinear_f32.restype = i32
    lib.fw_resize_bilinear_f32.argtypes = [vp, i32, i32, i32, vp, i32, i32, i32, i32, f32, f32, vp]
    lib.fw_ifnet_build_x.restype = i32
    lib.fw_ifnet_build_x.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp, vp]
    lib.fw_unshuffle2_cast.restype = i32
    lib.fw_unshuffle2_cast.argtypes = [i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]
    lib.fw_depth_to_space4_f32.restype = i32
    lib.fw_depth_to_space4_f32.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.fw_ifnet_accumulate.restype = i32
    lib.fw_ifnet_accumulate.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, i32, vp]
    lib.fw_ifnet_blend.restype = i32
    lib.fw_ifnet_blend.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.fw_aesrgan_create.restype = i32
    lib.fw_aesrgan_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.fw_aesrgan_set_tensor.restype = i32
    lib.fw_aesrgan_set_tensor.argtypes = [vp, C.c_char_p, vp, sz]
    lib.fw_aesrgan_finalize.restype = i32
    lib.fw_aesrgan_finalize.argtypes = [vp]
    lib.fw_aesrgan_forward_rgb.restype = i32
    lib.fw_aesrgan_forward_rgb.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.fw_aesrgan_workspace_bytes.restype = sz
    lib.fw_aesrgan_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fw_aesrgan_destroy.restype = i32
    lib.fw_aesrgan_destroy.argtypes = [vp]
    lib.fw_srvgg_create.restype = i32
    lib.fw_srvgg_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.fw_srvgg_set_tensor.restype = i32
    lib.fw_srvgg_set_tensor.argtypes = [vp, C.c_char_p, vp, sz]
    lib.fw_srvgg_finalize.restype = i32
    lib.fw_srvgg_finalize.argtypes = [vp]
    lib.fw_srvgg_upscale_u8.restype = i32
    lib.fw_srvgg_upscale_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_srvgg_upscale_u16.restype = i32
    lib.fw_srvgg_upscale_u16.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_resize_lanczos4_u16.restype = i32
    lib.fw_resize_lanczos4_u16.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp]
    lib.fw_srvgg_workspace_bytes.restype = sz
    lib.fw_srvgg_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fw_srvgg_flops.restype = C.c_double
    lib.fw_srvgg_flops.argtypes = [vp, i32, i32]
    lib.fw_srvgg_destroy.restype = i32
    lib.fw_srvgg_destroy.argtypes = [vp]
    lib.fw_ifnet_create.restype = i32
    lib.fw_ifnet_create.argtypes = [i32, i32, C.POINTER(vp)]
    lib.fw_ifnet_set_tensor.restype = i32
    lib.fw_ifnet_set_tensor.argtypes = [vp, C.c_char_p, vp, sz]
    lib.fw_ifnet_finalize.restype = i32
    lib.fw_ifnet_finalize.argtypes = [vp]
    lib.fw_ifnet_interp_u8.restype = i32
    lib.fw_ifnet_interp_u8.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, i32, vp, vp]
    lib.fw_ifnet_workspace_bytes.restype = sz
    lib.fw_ifnet_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fw_ifnet_flops.restype = C.c_double
    lib.fw_ifnet_flops.argtypes = [vp, i32, i32]
    lib.fw_ifnet_destroy.restype = i32
    lib.fw_ifnet_destroy.argtypes = [vp]
    lib.fw_restormer_create.restype = i32
    lib.fw_restormer_create.argtypes = [i32, i32, C.POINTER(i32), i32, C.POINTER(i32), C.c_double, i32, C.POINTER(vp)]
    lib.fw_restormer_set_tensor.restype = i32
    lib.fw_restormer_set_tensor.argtypes = [vp, C.c_char_p, vp, sz]
    lib.fw_restormer_finalize.restype = i32
    lib.fw_restormer_finalize.argtypes = [vp]
    lib.fw_restormer_denoise_u8.restype = i32
    lib.fw_restormer_denoise_u8.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
    lib.fw_restormer_workspace_bytes.restype = sz
    lib.fw_restormer_workspace_bytes.argtypes = [vp, i32, i32]
    lib.fw_restormer_destroy.restype = i32
    lib.fw_restormer_destroy.argtypes = [vp]
    lib.fw_preserve_edges_scratch_bytes.restype = sz
    lib.fw_preserve_edges_scratch_bytes.argtypes = [i32, i32]
    lib.fw_preserve_edges_u8.restype = i32
    lib.fw_preserve_edges_u8.argtypes = [vp, vp, i32, i32, C.c_double, C.c_double, vp, vp, vp]
    lib.fw_unsharp_mask_u8.restype = i32
    lib.fw_unsharp_mask_u8.argtypes = [vp, i32, i32, i32, i32, C.c_uint, C.c_uint, i32, i32, i32, vp, vp, vp, vp]
    lib.fw_u8_to_nhwc.restype = i32
    lib.fw_u8_to_nhwc.argtypes = [i32, vp, i32, i32, vp, i32, vp]
    lib.fw_pixel_shuffle_add_u8.restype = i32
    lib.fw_pixel_shuffle_add_u8.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp]
    i64 = C.c_long
    lib.fw_layernorm_nhwc.restype = i32
    lib.fw_layernorm_nhwc.argtypes = [i32, vp, i64, i64, i32, vp, vp, f32, vp, i64, i32, vp]
    lib.fw_pack_pointwise.restype = sz
    lib.fw_pack_pointwise.argtypes = [i32, vp, i32, i32, vp]
    lib.fw_pointwise_nhwc.restype = i32
    lib.fw_pointwise_nhwc.argtypes = [i32, vp, i32, i64, i64, i32, vp, vp, i32, vp, i64, vp, i64, vp, vp, vp]
    lib.fw_dwconv3x3_nhwc.restype = i32
    lib.fw_dwconv3x3_nhwc.argtypes = [i32, vp, i64, i32, i32, i32, vp, i32, vp, i64, vp]
    lib.fw_attn_workspace_floats.restype = sz
    lib.fw_attn_workspace_floats.argtypes = [i32, i32]
    lib.fw_attn_matrix.restype = i32
    lib.fw_attn_matrix.argtypes = [i32, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp]
    lib.fw_attn_qk_scratch_elems.restype = sz
    lib.fw_attn_qk_scratch_elems.argtypes = [i64, i32, i32]
    lib.fw_attn_matrix_mfma.restype = i32
    lib.fw_attn_matrix_mfma.argtypes = [i32, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.fw_attn_apply.restype = i32
    lib.fw_attn_apply.argtypes = [i32, vp, i64, i64, i32, i32, i32, vp, vp, i64, i32, vp]
    lib.fw_attn_pack.restype = i32
    lib.fw_attn_pack.argtypes = [i32, vp, i32, i32, i32, vp, vp]
    lib.fw_attn_proj_pack.restype = i32
    lib.fw_attn_proj_pack.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.fw_pixel_shuffle2_f32.restype = i32
    lib.fw_pixel_shuffle2_f32.argtypes = [vp, i64, i32, i32, i32, vp, i64, i32, i32, vp]
    lib.fw_copy_channels_f32.restype = i32
    lib.fw_copy_channels_f32.argtypes = [vp, i64, i64, i32, vp, i64, i32, vp]
    lib.fw_attn_softmax_rows.restype = i32
    lib.fw_attn_softmax_rows.argtypes = [i32, vp, i64, vp, i64, i64, i32, vp, i64, vp]
    lib.fw_pack_pointwise_transposed.restype = i32
    lib.fw_pack_pointwise_transposed.argtypes = [i32, vp, i64, i64, i32, i32, vp, vp]
    lib.fw_f32_to_planar.restype = i32
    lib.fw_f32_to_planar.argtypes = [i32, vp, i64, i32, vp, vp]
    lib.fw_flow_accumulate_u8.restype = i32
    lib.fw_flow_accumulate_u8.argtypes = [vp, vp, vp, vp, C.c_double, vp, f32, i32, i32, i32, vp, vp, vp]
    lib.fw_flow_accumulate_finish_u8.restype = i32
    lib.fw_flow_accumulate_finish_u8.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.fw_tap_post_u8.restype = i32
    lib.fw_tap_post_u8.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]


def load() -> C.CDLL:
    """Load the shared library (once).  Raises FramewrightHipError when it is absent — build it with
    ``python __graft_entry__.py build`` (hipcc, gfx950)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise FramewrightHipError(FW_ERR_INTERNAL,
                                      f"{LIB_PATH} not found: the HIP extension is not built "
                                      "(run `python __graft_entry__.py build`); there is no CPU fallback")
        # torch bundles its own libamdhip64.so.7; importing it first makes both share ONE HIP runtime, so
        # torch.cuda device pointers and streams can be handed to the library.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, the library also works stand-alone
            pass
        try:
            lib = C.CDLL(str(LIB_PATH))
        except OSError as e:
            raise FramewrightHipError(FW_ERR_INTERNAL, f"cannot load {LIB_PATH}: {e}") from e
        _declare(lib)
        _declare_tap(lib)
        _declare_ifnet(lib)
        _lib = lib
        return lib


def check(status: int) -> None:
    if status == FW_OK:
        return
    msg = load().fw_last_error().decode("utf-8", "replace")
    if status == FW_ERR_OOM:
        raise FramewrightOutOfMemory(status, msg)
    raise FramewrightHipError(status, msg)


def require_gpu() -> int:
    """Number of visible devices; raises when there is none (the product path never runs on the CPU)."""
    n = load().fw_device_count()
    if n <= 0:
        raise FramewrightHipError(FW_ERR_HIP, "no HIP device visible: framewright_amd has no CPU fallback")
    return n


def on_tensor_device(fn):
    """Decorator for device-level entry methods: run the body with the CUDA device of the first tensor argument current, so
    that ``torch.cuda.current_stream()``, scratch allocations and the library's block-level launchers (which launch on the
    current device: zero pages, default streams) all refer to the device that owns the buffers - an engine created with
    ``device_id != 0`` must not depend on the caller's current device."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        import torch
        dev = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dev = a.device
                break
            if isinstance(a, (list, tuple)):
                t = next((x for x in a if isinstance(x, torch.Tensor) and x.is_cuda), None)   # any element: [None, tensor, ...] is a rank's block
                if t is not None:
                    dev = t.device
                    break
        if dev is None and args:
            # no CUDA tensor among the arguments (numpy frames, None placeholders): the owner says where it lives -
            # an engine's device (`_dev` / `device_id`) or a driver's `config.gpu_id`
            owner = args[0]
            cand = getattr(owner, "_dev", None)
            if cand is None:
                cand = getattr(owner, "device_id", None)
            if cand is None:
                cand = getattr(getattr(owner, "config", None), "gpu_id", None)
            if isinstance(cand, torch.device):
                dev = cand if cand.type == "cuda" else None
            elif isinstance(cand, int) and not isinstance(cand, bool) and cand >= 0 and torch.cuda.is_available():
                dev = torch.device("cuda", cand)
        if dev is None:
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def empty_like_many(frames) -> list:
    """Output buffers for a batch of frames: ONE allocation cut into views when the frames agree in shape (the usual clip), else one
    per frame.  A hipMalloc is a device-wide synchronisation: a hundred of them in front of a hundred forwards that are meant to
    overlap on side streams serialise those streams (measured on the `rife` bench line: 1.21 - 1.40 ms per 1080p pair with one
    allocation per mid frame on a cold allocator, 0.83 with the buffers in place - profiles/r03_ab/rife_output_buffers.txt)."""
    import torch
    frames = list(frames)
    if len(frames) > 1 and all(f.shape == frames[0].shape and f.dtype == frames[0].dtype and f.device == frames[0].device for f in frames):
        buf = torch.empty((len(frames),) + tuple(frames[0].shape), dtype=frames[0].dtype, device=frames[0].device)
        return list(buf.unbind(0))
    return [torch.empty_like(f) for f in frames]


def side_streams(env_name: str, default: int) -> int:
    """How many forwards of one engine kind may be in flight on their own streams (with engine clones).  ``env_name`` overrides.
    Measured on MI355X / ROCm 7.2: up to three worker streams beside the caller's overlap as intended; with a fourth the work
    serialises (RIFE) or runs 1.7x slower than one stream (NAFNet) - the process's hardware queues are shared by then - so the
    defaults stay below that, and a process that is one rank of several (RCCL brings streams of its own) runs one at a time."""
    import os
    v = os.environ.get(env_name)
    if v is not None:
        return max(1, int(v))
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return 1
    except Exception:  # noqa: BLE001 - no torch.distributed: a single process
        pass
    return default
