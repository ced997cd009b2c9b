"""SRVGGNetCompact on the MI355X conv kernels — the network of the Real-ESRGAN checkpoints `realesr-animevideov3` and
`realesr-general-x4v3`.

The reference lists both in its model table but builds an RRDBNet for them (processors/pytorch_realesrgan.py:119-128), which
cannot load the published weights; SURVEY.md §8(f) item 4 asks for the intended behaviour.  `realesrgan.get_upsampler`
routes a checkpoint here when its state dict has SRVGG keys (`body.N.weight`), and to the RRDBNet engine otherwise, so the
reference's behaviour for RRDB-shaped checkpoints under those names is unchanged.

Every conv is the 64-output-channel instantiation of csrc/conv3x3_mfma.hip (chunk-planar typed activations, PReLU fused
into the epilogue); the last conv leaves fp32, and one small kernel does PixelShuffle + nearest-upsampled input + uint8.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Mapping, Tuple

import numpy as np

from . import _lib
from ._lib import FramewrightHipError

SRVGG_MODELS = {
    # name: (num_conv, netscale)            (Real-ESRGAN release notes; realesrgan/archs/srvgg_arch.py)
    "realesr-animevideov3": (16, 4),
    "realesr-general-x4v3": (32, 4),
}
NUM_FEAT = 64


def unwrap_state(state: Mapping[str, object]) -> Mapping[str, object]:
    """``params_ema`` / ``params`` is unwrapped like RealESRGANer does."""
    if "params_ema" in state:
        return state["params_ema"]  # type: ignore[return-value]
    if "params" in state:
        return state["params"]  # type: ignore[return-value]
    return state


def is_srvgg_state_dict(state: Mapping[str, object]) -> bool:
    state = unwrap_state(state)
    return "body.0.weight" in state and "conv_first.weight" not in state


def srvgg_tensor_shapes(num_conv: int, scale: int) -> List[Tuple[str, Tuple[int, ...]]]:
    shapes: List[Tuple[str, Tuple[int, ...]]] = []
    for i in range(num_conv + 2):
        cin = 3 if i == 0 else NUM_FEAT
        cout = 3 * scale * scale if i == num_conv + 1 else NUM_FEAT
        shapes += [(f"body.{2 * i}.weight", (cout, cin, 3, 3)), (f"body.{2 * i}.bias", (cout,))]
        if i < num_conv + 1:
            shapes.append((f"body.{2 * i + 1}.weight", (NUM_FEAT,)))
    return shapes


def synthetic_srvgg_state(num_conv: int, scale: int, seed: int = 0):
    """Seeded weights with the published keys/shapes; scaled so activations stay O(1) through the stack."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in srvgg_tensor_shapes(num_conv, scale):
        if key.endswith(".bias"):
            sd[key] = (rng.standard_normal(shape) * 0.02).astype(np.float32)
        elif len(shape) == 1:
            sd[key] = (0.1 + 0.2 * rng.random(shape)).astype(np.float32)           # PReLU slopes
        else:
            fan_in = shape[1] * 9
            gain = 0.1 if shape[0] != NUM_FEAT else 1.3                            # last conv: a small residual on the base
            sd[key] = (rng.standard_normal(shape) * gain / np.sqrt(fan_in)).astype(np.float32)
    return sd


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    return t.detach().cpu().float().numpy()


class SRVGGNetEngine:
    """SRVGGNetCompact resident on one GPU; same surface as RRDBNetEngine (load_state_dict / upscale_device / flops).  Thin owner of
    an ``fw_srvgg*`` (csrc/srvgg.hip): weight packing, the workspace and the launches of a forward live behind the C-ABI
    (``fw_srvgg_upscale_u8``), serialised per handle by its mutex."""

    def __init__(self, num_conv: int, scale: int = 4, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if scale not in (1, 2, 3, 4) or 3 * scale * scale > 64:
            raise ValueError("SRVGGNetEngine: scale must be 1..4")
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
        self.num_conv, self.scale, self.dtype, self.device_id = int(num_conv), int(scale), dtype, int(device_id)
        self._dev = torch.device("cuda", self.device_id)
        h = C.c_void_p()
        _lib.check(self._lib.fw_srvgg_create(self.device_id, NUM_FEAT, self.num_conv, self.scale, _lib.DTYPES[dtype], C.byref(h)))
        self._h = h
        self._loaded = False

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        state = unwrap_state(state)
        for key, shape in srvgg_tensor_shapes(self.num_conv, self.scale):
            if key not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {key}")
            a = np.ascontiguousarray(_to_numpy(state[key]), dtype=np.float32)
            if a.ndim == 1 and len(shape) == 1:
                a = a.reshape(-1)
            if tuple(a.shape) != tuple(shape):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{key}: expected shape {shape}, got {a.shape}")
            _lib.check(self._lib.fw_srvgg_set_tensor(self._h, key.encode(), C.c_void_p(a.ctypes.data), a.size))
        _lib.check(self._lib.fw_srvgg_finalize(self._h))
        self._loaded = True

    def flops(self, H: int, W: int) -> float:
        return float(self._lib.fw_srvgg_flops(self._h, H, W))

    @_lib.on_tensor_device
    def upscale_device(self, frame_bgr, out=None, out_rgb_f32=None):
        """frame_bgr: uint8 CUDA tensor H x W x 3.  Returns the uint8 BGR result (asynchronous on torch's current stream)."""
        import torch
        if not self._loaded:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "SRVGGNetEngine: no weights loaded")
        t = frame_bgr
        if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 3 or t.shape[2] != 3 or not t.is_contiguous():
            raise ValueError("upscale_device expects a contiguous uint8 CUDA tensor H x W x 3")
        if t.device != self._dev:
            raise ValueError(f"tensor is on {t.device}, engine on {self._dev}")
        s = self.scale
        H, W = int(t.shape[0]), int(t.shape[1])
        if out is None:
            out = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=self._dev)
        for x, dt in ((out, torch.uint8), (out_rgb_f32, torch.float32)):
            if x is not None and (x.dtype != dt or tuple(x.shape) != (H * s, W * s, 3) or not x.is_contiguous() or x.device != self._dev):
                raise ValueError("output tensor has the wrong dtype/shape/device")
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self._lib.fw_srvgg_upscale_u8(self._h, p(t), _lib.FW_DEVICE, H, W, p(out), _lib.FW_DEVICE, p(out_rgb_f32), st))
        return out

    def upscale(self, frame_bgr: np.ndarray) -> np.ndarray:
        """H x W x 3 uint8 BGR (host) -> sH x sW x 3 uint8 BGR (host)."""
        import torch
        if isinstance(frame_bgr, np.ndarray) and frame_bgr.dtype == np.uint16:
            # a 16-bit frame (range 65535) comes back as uint16, as RealESRGANer.enhance returns it
            if frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3:
                raise ValueError("expected an H x W x 3 uint16 BGR frame")
            if not self._loaded:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, "SRVGGNetEngine: no weights loaded")
            f16 = np.ascontiguousarray(frame_bgr)
            h, w = f16.shape[:2]
            out16 = np.empty((h * self.scale, w * self.scale, 3), dtype=np.uint16)
            _lib.check(self._lib.fw_srvgg_upscale_u16(self._h, C.c_void_p(f16.ctypes.data), _lib.FW_HOST, h, w, C.c_void_p(out16.ctypes.data),
                                                      _lib.FW_HOST, None, None))
            return out16
        if not isinstance(frame_bgr, np.ndarray) or frame_bgr.dtype != np.uint8 or frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR frame")
        with torch.cuda.device(self._dev):
            out = self.upscale_device(torch.from_numpy(np.ascontiguousarray(frame_bgr)).to(self._dev))
            torch.cuda.synchronize(self._dev)
        return out.cpu().numpy()

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.fw_srvgg_destroy(h)
        self._loaded = False
