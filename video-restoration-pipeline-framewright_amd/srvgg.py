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
from typing import List, Mapping, Optional, Tuple

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
    """SRVGGNetCompact resident on one GPU; same surface as RRDBNetEngine (load_state_dict / upscale_device / flops)."""

    def __init__(self, num_conv: int, scale: int = 4, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if scale not in (1, 2, 3, 4) or 3 * scale * scale > 64:
            raise ValueError("SRVGGNetEngine: scale must be 1..4")
        self.num_conv, self.scale, self.dtype, self.device_id = int(num_conv), int(scale), dtype, int(device_id)
        self._dt = _lib.DTYPES[dtype]
        self._tdt = torch.float16 if self._dt == _lib.FW_DTYPE_F16 else torch.bfloat16
        self._dev = torch.device("cuda", self.device_id)
        self._layers: List[Tuple[object, object, Optional[object], int]] = []   # packed w, bias, slopes, cin chunks

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        import torch
        lib = self._lib
        state = unwrap_state(state)
        layers = []
        for i in range(self.num_conv + 2):
            wk, bk, pk = f"body.{2 * i}.weight", f"body.{2 * i}.bias", f"body.{2 * i + 1}.weight"
            for k in (wk, bk) + ((pk,) if i < self.num_conv + 1 else ()):
                if k not in state:
                    raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {k}")
            w = np.ascontiguousarray(_to_numpy(state[wk]), dtype=np.float32)
            b = np.ascontiguousarray(_to_numpy(state[bk]), dtype=np.float32)
            cin = 3 if i == 0 else NUM_FEAT
            cout = 3 * self.scale ** 2 if i == self.num_conv + 1 else NUM_FEAT
            if w.shape != (cout, cin, 3, 3) or b.shape != (cout,):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{wk}: expected shape {(cout, cin, 3, 3)}, got {w.shape}")
            chunks = (cin + 31) // 32
            wp = np.zeros((64, 32 * chunks, 3, 3), np.float32)
            wp[:cout, :cin] = w
            bp = np.zeros((64,), np.float32)
            bp[:cout] = b
            n = lib.fw_pack_conv3x3(self._dt, None, 64, 32 * chunks, 2, chunks, None)
            buf = np.zeros(n, np.uint16)
            if lib.fw_pack_conv3x3(self._dt, C.c_void_p(wp.ctypes.data), 64, 32 * chunks, 2, chunks,
                                   C.c_void_p(buf.ctypes.data)) != n:
                raise FramewrightHipError(_lib.FW_ERR_INTERNAL, "fw_pack_conv3x3 failed")
            slopes = None
            if i < self.num_conv + 1:
                sl = np.ascontiguousarray(_to_numpy(state[pk]), dtype=np.float32).reshape(-1)
                if sl.shape != (NUM_FEAT,):
                    raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{pk}: expected shape ({NUM_FEAT},), got {sl.shape}")
                slopes = torch.from_numpy(sl).to(self._dev)
            layers.append((torch.from_numpy(buf.view(np.int16)).to(self._dev), torch.from_numpy(bp).to(self._dev), slopes, chunks))
        self._layers = layers

    def flops(self, H: int, W: int) -> float:
        mac = 9.0 * (3 * NUM_FEAT + self.num_conv * NUM_FEAT * NUM_FEAT + NUM_FEAT * 3 * self.scale ** 2)
        return 2.0 * mac * H * W

    @_lib.on_tensor_device
    def upscale_device(self, frame_bgr, out=None, out_rgb_f32=None):
        """frame_bgr: uint8 CUDA tensor H x W x 3.  Returns the uint8 BGR result (asynchronous on torch's current stream)."""
        import torch
        if not self._layers:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "SRVGGNetEngine: no weights loaded")
        t = frame_bgr
        if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 3 or t.shape[2] != 3 or not t.is_contiguous():
            raise ValueError("upscale_device expects a contiguous uint8 CUDA tensor H x W x 3")
        lib, dev, s = self._lib, t.device, self.scale
        H, W = int(t.shape[0]), int(t.shape[1])
        if out is None:
            out = torch.empty((H * s, W * s, 3), dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        PL = H * W * 32                                        # elements per 32-channel plane
        x0 = torch.empty((H, W, 32), dtype=self._tdt, device=dev)
        _lib.check(lib.fw_u8_to_nhwc(self._dt, p(t), H, W, p(x0), 32, st))
        bufs = [torch.empty((2, H, W, 32), dtype=self._tdt, device=dev) for _ in range(2)]   # chunk-planar, ping-pong
        last = torch.empty((H, W, 64), dtype=torch.float32, device=dev)
        cur = x0
        for i, (wp, b, slopes, chunks) in enumerate(self._layers):
            final = i == len(self._layers) - 1
            dst = None if final else bufs[i & 1]
            _lib.check(lib.fw_conv3x3_nhwc_ex(
                self._dt, p(cur), 32, PL if chunks > 1 else 0, chunks, H, W, p(wp), p(b), 2, 0 if final else 2, 0,
                None, 1.0, None, 1.0, p(slopes), 0, 0, 0, p(dst), 32, PL, 0, p(last) if final else None, st))
            cur = dst
        _lib.check(lib.fw_pixel_shuffle_add_u8(p(last), 64, p(t), H, W, s, p(out), p(out_rgb_f32), st))
        return out

    def upscale(self, frame_bgr: np.ndarray) -> np.ndarray:
        """H x W x 3 uint8 BGR (host) -> sH x sW x 3 uint8 BGR (host)."""
        import torch
        if not isinstance(frame_bgr, np.ndarray) or frame_bgr.dtype != np.uint8 or frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR frame")
        with torch.cuda.device(self._dev):
            out = self.upscale_device(torch.from_numpy(np.ascontiguousarray(frame_bgr)).to(self._dev))
            torch.cuda.synchronize(self._dev)
        return out.cpu().numpy()

    def close(self) -> None:
        self._layers = []
