"""Real-ESRGAN on MI355X behind the reference's own function boundary (boundary B1).

Mirrors reference ``src/framewright/processors/pytorch_realesrgan.py`` name for name:

* ``PyTorchESRGANConfig`` (:36-61), ``is_pytorch_esrgan_available`` (:64), ``get_upsampler`` (:85-173),
  ``enhance_frame_pytorch`` (:176-247), ``clear_upsampler_cache`` (:250), ``NCNN_TO_PYTORCH_MODEL`` /
  ``convert_ncnn_model_name`` (:263-275)
* ``HipRealESRGANer.enhance(img, outscale) -> (output, img_mode)`` stands where the reference uses the
  third-party ``realesrgan.RealESRGANer`` (:160-170, :223); semantics per SURVEY.md §A.2.

All arithmetic happens in libframewright_hip.so (``fw_rrdbnet_*``); this file only moves bytes, slices tiles and
maps errors to the reference's ``(ok, message)`` convention.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import logging
import math
import os
import threading
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Mapping, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import FramewrightHipError, FramewrightOutOfMemory
from .synth import RRDB_MODELS, rrdbnet_conv_shapes, synthetic_rrdbnet_state

logger = logging.getLogger(__name__)

# weights live where the reference's ModelManager puts them (utils/model_manager.py:469)
DEFAULT_MODEL_DIR = Path.home() / ".framewright" / "models"
MODEL_FILES = {
    "RealESRGAN_x4plus": "RealESRGAN_x4plus.pth",
    "RealESRGAN_x4plus_anime_6B": "RealESRGAN_x4plus_anime_6B.pth",
    "RealESRGAN_x2plus": "RealESRGAN_x2plus.pth",
    "realesr-animevideov3": "realesr-animevideov3.pth",
    "realesr-general-x4v3": "realesr-general-x4v3.pth",
}


def _np_ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class RRDBNetEngine:
    """One RRDBNet resident on one GPU (thin owner of an ``fw_rrdbnet*``)."""

    def __init__(self, num_block: int = 23, scale: int = 4, dtype: str = "f16", device_id: int = 0):
        self._lib = _lib.load()
        _lib.require_gpu()
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
        self.num_block, self.scale, self.dtype, self.device_id = int(num_block), int(scale), dtype, int(device_id)
        h = C.c_void_p()
        _lib.check(self._lib.fw_rrdbnet_create(self.device_id, self.num_block, self.scale, _lib.DTYPES[dtype],
                                               C.byref(h)))
        self._h = h
        self._loaded = False
        # close() vs. calls in flight: the reference drives one shared upsampler from a thread pool (restorer.py:1830-1973) and
        # its OOM path clears the cache from whichever worker failed - the handle must not be destroyed under a sibling's call
        self._cv = threading.Condition()
        self._in_use = 0

    @contextlib.contextmanager
    def _handle(self):
        with self._cv:
            h = self._h
            if not h:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, "RRDBNetEngine: the engine has been closed")
            self._in_use += 1
        try:
            yield h
        finally:
            with self._cv:
                self._in_use -= 1
                self._cv.notify_all()

    # -- weights ------------------------------------------------------------------------------------
    def load_state_dict(self, state: Mapping[str, object]) -> None:
        """``state``: BasicSR RRDBNet state-dict (numpy arrays or torch tensors); a checkpoint dict with
        ``params_ema`` / ``params`` is unwrapped like RealESRGANer does."""
        if "params_ema" in state:
            state = state["params_ema"]  # type: ignore[assignment]
        elif "params" in state:
            state = state["params"]  # type: ignore[assignment]
        for key, cout, cin in rrdbnet_conv_shapes(self.num_block, self.scale):
            try:
                w = state[key + ".weight"]
                b = state[key + ".bias"]
            except KeyError as e:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {e.args[0]}") from None
            w = np.ascontiguousarray(_to_numpy(w), dtype=np.float32)
            b = np.ascontiguousarray(_to_numpy(b), dtype=np.float32)
            if w.shape != (cout, cin, 3, 3) or b.shape != (cout,):
                raise FramewrightHipError(_lib.FW_ERR_INVALID,
                                          f"{key}: expected weight {(cout, cin, 3, 3)}, got {w.shape}")
            with self._handle() as hd:
                _lib.check(self._lib.fw_rrdbnet_set_conv(hd, key.encode(), _np_ptr(w), _np_ptr(b), cout, cin))
        with self._handle() as hd:
            _lib.check(self._lib.fw_rrdbnet_finalize(hd))
        self._loaded = True

    # -- inference ----------------------------------------------------------------------------------
    def upscale(self, frame_bgr: np.ndarray) -> np.ndarray:
        """H x W x 3 uint8 BGR (host) -> sH x sW x 3 uint8 BGR (host); a uint16 frame (range 65535) comes back as uint16."""
        if isinstance(frame_bgr, np.ndarray) and frame_bgr.dtype == np.uint16:
            if frame_bgr.ndim != 3 or frame_bgr.shape[2] != 3:
                raise ValueError("expected an H x W x 3 uint16 BGR frame")
            f16 = np.ascontiguousarray(frame_bgr)
            h, w = f16.shape[:2]
            out16 = np.empty((h * self.scale, w * self.scale, 3), dtype=np.uint16)
            with self._handle() as hd:
                _lib.check(self._lib.fw_rrdbnet_upscale_u16(hd, _np_ptr(f16), _lib.FW_HOST, h, w, _np_ptr(out16),
                                                            _lib.FW_HOST, None, None))
            return out16
        frame_bgr = _check_frame(frame_bgr)
        h, w = frame_bgr.shape[:2]
        out = np.empty((h * self.scale, w * self.scale, 3), dtype=np.uint8)
        with self._handle() as hd:
            _lib.check(self._lib.fw_rrdbnet_upscale_u8(hd, _np_ptr(frame_bgr), _lib.FW_HOST, h, w, _np_ptr(out),
                                                       _lib.FW_HOST, None, None))
        return out

    def upscale_device(self, frame_bgr, out=None, out_rgb_f32=None, stream: Optional[int] = None):
        """torch.uint8 CUDA tensor H x W x 3 -> torch.uint8 CUDA tensor sH x sW x 3, asynchronous on ``stream``
        (default: torch's current stream).  ``out_rgb_f32`` (optional float32 CUDA tensor sH x sW x 3) receives
        the un-clamped RGB network output."""
        import torch

        if frame_bgr.dtype != torch.uint8 or not frame_bgr.is_cuda or frame_bgr.dim() != 3 or \
                frame_bgr.shape[2] != 3 or not frame_bgr.is_contiguous():
            raise ValueError("upscale_device expects a contiguous uint8 CUDA tensor H x W x 3")
        if frame_bgr.device.index != self.device_id:
            raise ValueError(f"tensor is on {frame_bgr.device}, engine on cuda:{self.device_id}")
        h, w = int(frame_bgr.shape[0]), int(frame_bgr.shape[1])
        s = self.scale
        if out is None and out_rgb_f32 is None:
            out = torch.empty((h * s, w * s, 3), dtype=torch.uint8, device=frame_bgr.device)
        for t, dt in ((out, torch.uint8), (out_rgb_f32, torch.float32)):
            if t is not None and (t.dtype != dt or tuple(t.shape) != (h * s, w * s, 3) or not t.is_contiguous()
                                  or t.device != frame_bgr.device):
                raise ValueError("output tensor has the wrong dtype/shape/device")
        if stream is None:
            stream = torch.cuda.current_stream(frame_bgr.device).cuda_stream
        with self._handle() as hd:
            _lib.check(self._lib.fw_rrdbnet_upscale_u8(
                hd, C.c_void_p(frame_bgr.data_ptr()), _lib.FW_DEVICE, h, w,
                C.c_void_p(out.data_ptr()) if out is not None else None, _lib.FW_DEVICE,
                C.c_void_p(out_rgb_f32.data_ptr()) if out_rgb_f32 is not None else None, C.c_void_p(stream)))
        return out if out is not None else out_rgb_f32

    def upscale_stream(self, frames, depth: int = 2):
        """Host frames in, host frames out, with the PCIe copies overlapped with compute: generator over uint8 BGR results
        in input order.  Three streams (upload / compute / download) and ``depth`` + 1 pinned staging slots each way: frame
        n+1 uploads and frame n-1 downloads while frame n computes (6.2 MB in, 99.5 MB out per 1080p x4 frame).  The
        yielded array is a view of a pinned slot that is reused ``depth`` frames later: copy it if it must outlive that."""
        import torch
        dev = torch.device("cuda", self.device_id)
        s = self.scale
        up, comp, down = (torch.cuda.Stream(device=dev) for _ in range(3))
        slots = depth + 1
        pin_in = pin_out = d_in = d_out = None
        cache = self.__dict__.setdefault("_stream_slots", {})   # pinning 100 MB slots costs tens of ms: keep them
        pending = []   # (slot, download-done event)
        ev_in_free = [None] * slots    # compute finished reading d_in[slot]
        ev_out_free = [None] * slots   # download finished reading d_out[slot]

        def drain(limit):
            while len(pending) > limit:
                k, ev = pending.pop(0)
                ev.synchronize()
                yield pin_out[k].numpy()

        for n, f in enumerate(frames):
            f = _check_frame(f)
            h, w = f.shape[:2]
            if pin_in is None:
                key = (h, w, slots)
                if key not in cache:
                    cache.clear()
                    cache[key] = ([torch.empty((h, w, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)],
                                  [torch.empty((h * s, w * s, 3), dtype=torch.uint8).pin_memory() for _ in range(slots)],
                                  [torch.empty((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(slots)],
                                  [torch.empty((h * s, w * s, 3), dtype=torch.uint8, device=dev) for _ in range(slots)])
                pin_in, pin_out, d_in, d_out = cache[key]
            elif tuple(pin_in[0].shape) != f.shape:
                raise ValueError("upscale_stream: all frames of a stream must have the same size")
            k = n % slots
            yield from drain(depth)           # frees pinned slot k (its download was ``slots`` frames ago)
            pin_in[k].numpy()[...] = f
            with torch.cuda.stream(up):
                if ev_in_free[k] is not None:
                    up.wait_event(ev_in_free[k])
                d_in[k].copy_(pin_in[k], non_blocking=True)
                ev_up = torch.cuda.Event()
                ev_up.record(up)
            with torch.cuda.stream(comp):
                comp.wait_event(ev_up)
                if ev_out_free[k] is not None:
                    comp.wait_event(ev_out_free[k])
                self.upscale_device(d_in[k], out=d_out[k], stream=comp.cuda_stream)
                ev_c = torch.cuda.Event()
                ev_c.record(comp)
                ev_in_free[k] = ev_c
            with torch.cuda.stream(down):
                down.wait_event(ev_c)
                pin_out[k].copy_(d_out[k], non_blocking=True)
                ev_d = torch.cuda.Event()
                ev_d.record(down)
                ev_out_free[k] = ev_d
            pending.append((k, ev_d))
        yield from drain(0)

    # -- introspection ------------------------------------------------------------------------------
    def flops(self, height: int, width: int) -> float:
        with self._handle() as hd:
            return float(self._lib.fw_rrdbnet_flops(hd, height, width))

    def workspace_bytes(self, height: int, width: int) -> int:
        with self._handle() as hd:
            return int(self._lib.fw_rrdbnet_workspace_bytes(hd, height, width))

    def profile_enable(self, on: bool) -> None:
        with self._handle() as hd:
            _lib.check(self._lib.fw_rrdbnet_profile_enable(hd, 1 if on else 0))

    def profile_read(self) -> Tuple[int, float, float]:
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        with self._handle() as hd:
            _lib.check(self._lib.fw_rrdbnet_profile_read(hd, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def close(self) -> None:
        """Destroys the native engine once no call is inside it; calls that arrive afterwards raise."""
        cv = getattr(self, "_cv", None)
        if cv is None:
            return
        with cv:
            h, self._h = self._h, None
            while h and self._in_use:
                cv.wait()
        if h:
            self._lib.fw_rrdbnet_destroy(h)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _to_numpy(t) -> np.ndarray:
    if isinstance(t, np.ndarray):
        return t
    if hasattr(t, "detach"):
        return t.detach().float().cpu().numpy()
    return np.asarray(t)


def _check_frame(frame: np.ndarray) -> np.ndarray:
    if not isinstance(frame, np.ndarray) or frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
        raise ValueError("expected an H x W x 3 uint8 BGR frame")
    if frame.shape[0] < 1 or frame.shape[1] < 1:
        raise ValueError("empty frame")
    return np.ascontiguousarray(frame)


def resize_lanczos4_u8(img: np.ndarray, dst_w: int, dst_h: int, device_id: int = 0) -> np.ndarray:
    """``cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LANCZOS4)`` for an 8-bit or 16-bit H x W [x C] array, on the GPU
    (fw_resize_lanczos4_u8: OpenCV's fixed-point arithmetic for uchar; fw_resize_lanczos4_u16: its float path for ushort;
    oracle/lanczos_ref.py holds the CPU restatements)."""
    import torch
    if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3) or dst_w < 1 or dst_h < 1:
        raise ValueError("resize_lanczos4_u8: uint8 / uint16 H x W [x C] image and a positive size expected")
    c = 1 if img.ndim == 2 else img.shape[2]
    wide = img.dtype == np.uint16
    with torch.cuda.device(device_id):
        a = np.ascontiguousarray(img)
        src = torch.from_numpy(a.view(np.int16) if wide else a).cuda()          # the bytes only: torch's uint16 support is partial
        dst = torch.empty((dst_h, dst_w) if img.ndim == 2 else (dst_h, dst_w, c), dtype=torch.int16 if wide else torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        fn = _lib.load().fw_resize_lanczos4_u16 if wide else _lib.load().fw_resize_lanczos4_u8
        _lib.check(fn(C.c_void_p(src.data_ptr()), img.shape[0], img.shape[1], c, C.c_void_p(dst.data_ptr()), dst_h, dst_w, C.c_void_p(st)))
        out = dst.cpu().numpy()
        return out.view(np.uint16) if wide else out


# ---------------------------------------------------------------------------------------------------
# RealESRGANer-compatible object (what get_upsampler() returns in the reference)
# ---------------------------------------------------------------------------------------------------
class HipRealESRGANer:
    """Drop-in for ``realesrgan.RealESRGANer`` as the reference uses it (pytorch_realesrgan.py:160-170,223):
    ``enhance(img, outscale=None, alpha_upsampler='realesrgan') -> (output, img_mode)``."""

    def __init__(self, scale: int, engine: RRDBNetEngine, tile: int = 0, tile_pad: int = 10, pre_pad: int = 0,
                 half: bool = True, gpu_id: Optional[int] = 0):
        self.scale, self.engine = int(scale), engine
        self.tile_size, self.tile_pad, self.pre_pad, self.half = int(tile or 0), int(tile_pad), int(pre_pad), half
        self.gpu_id = gpu_id
        self._mu = threading.Lock()

    # ---- whole-frame / tiled run on a 3-channel uint8 image ------------------------------------
    def _run_u8(self, bgr: np.ndarray) -> np.ndarray:
        s = self.scale
        if self.pre_pad:
            # F.pad(img, (0, pre_pad, 0, pre_pad), 'reflect') — SURVEY.md §A.2 pre_process
            bgr = np.pad(bgr, ((0, self.pre_pad), (0, self.pre_pad), (0, 0)), mode="reflect")
        h, w = bgr.shape[:2]
        if self.tile_size > 0 and (h > self.tile_size or w > self.tile_size):
            out = np.empty((h * s, w * s, 3), dtype=bgr.dtype)
            t, pad = self.tile_size, self.tile_pad
            for ty in range(math.ceil(h / t)):
                for tx in range(math.ceil(w / t)):
                    sx0, sx1 = tx * t, min(tx * t + t, w)
                    sy0, sy1 = ty * t, min(ty * t + t, h)
                    px0, px1 = max(sx0 - pad, 0), min(sx1 + pad, w)
                    py0, py1 = max(sy0 - pad, 0), min(sy1 + pad, h)
                    o = self.engine.upscale(np.ascontiguousarray(bgr[py0:py1, px0:px1]))
                    ox0, oy0 = (sx0 - px0) * s, (sy0 - py0) * s
                    out[sy0 * s:sy1 * s, sx0 * s:sx1 * s] = o[oy0:oy0 + (sy1 - sy0) * s, ox0:ox0 + (sx1 - sx0) * s]
        else:
            out = self.engine.upscale(np.ascontiguousarray(bgr))
        if self.pre_pad:
            out = out[:h * s - self.pre_pad * s, :w * s - self.pre_pad * s]
        return out

    def enhance(self, img: np.ndarray, outscale: Optional[float] = None, alpha_upsampler: str = "realesrgan"):
        out, mode = self._enhance_netscale(img)
        if outscale is not None and float(outscale) != float(self.scale):
            # RealESRGANer.enhance: cv2.resize(output, (int(w_input * outscale), int(h_input * outscale)), INTER_LANCZOS4)
            # on the quantised output, whatever its channel count (reached with scale_factor 2 and a x4 model,
            # pytorch_realesrgan.py:223)
            h_in, w_in = img.shape[:2]
            out = resize_lanczos4_u8(out, int(w_in * outscale), int(h_in * outscale), self.engine.device_id)   # uint8 or uint16
        return out, mode

    def _enhance_netscale(self, img: np.ndarray):
        img = np.asarray(img)
        # RealESRGANer.enhance: `if np.max(img) > 256: max_range = 65535` - a 16-bit image (cv2.imread(IMREAD_UNCHANGED) of a
        # 16-bit PNG) is normalised by 65535 and comes back as uint16; anything else is 8-bit
        sixteen = img.size > 0 and float(np.max(img)) > 256
        if sixteen:
            img = img if img.dtype == np.uint16 else np.clip(np.rint(img), 0, 65535).astype(np.uint16)
            top = 65535
        else:
            img = img if img.dtype == np.uint8 else np.clip(np.rint(img), 0, 255).astype(np.uint8)
            top = 255

        def gray_of(bgr_out):
            # cv2.COLOR_BGR2GRAY weights, on the float image before quantisation in the reference; here on the quantised
            # output (|diff| <= 1 LSB)
            g = 0.114 * bgr_out[:, :, 0].astype(np.float32) + 0.587 * bgr_out[:, :, 1] + 0.299 * bgr_out[:, :, 2]
            return np.clip(np.rint(g), 0, top).astype(img.dtype)

        with self._mu:
            if img.ndim == 2:
                return gray_of(self._run_u8(np.repeat(img[:, :, None], 3, axis=2))), "L"
            if img.shape[2] == 4:
                out = self._run_u8(np.ascontiguousarray(img[:, :, :3]))
                ag = gray_of(self._run_u8(np.repeat(img[:, :, 3:4], 3, axis=2)))
                return np.concatenate([out, ag[:, :, None]], axis=2), "RGBA"
            return self._run_u8(img if sixteen else _check_frame(img)), "RGB"


# ---------------------------------------------------------------------------------------------------
# reference module surface (pytorch_realesrgan.py)
# ---------------------------------------------------------------------------------------------------
@dataclass
class PyTorchESRGANConfig:
    """Field-for-field copy of the reference dataclass (pytorch_realesrgan.py:36-44) plus ``dtype``."""
    model_name: str = "RealESRGAN_x4plus"
    scale_factor: int = 4
    tile_size: int = 0  # 0 = auto
    tile_pad: int = 10
    pre_pad: int = 0
    half_precision: bool = True
    gpu_id: int = 0
    dtype: str = "f16"           # MFMA operand type: "f16" (the reference's half=True; meets the 1e-3 bar) or "bf16"
    model_path: Optional[str] = None   # local .pth; default ~/.framewright/models/<file>

    def validate(self) -> None:
        if self.model_name not in RRDB_MODELS:
            raise ValueError(f"Invalid model: {self.model_name}. Supported models: {', '.join(RRDB_MODELS)}")
        if self.scale_factor not in [2, 4]:
            raise ValueError(f"Scale factor must be 2 or 4, got {self.scale_factor}")


_UPSAMPLERS: Dict[tuple, HipRealESRGANer] = {}
_UPSAMPLER_LOCK = threading.Lock()


def is_pytorch_esrgan_available() -> bool:
    """True when the HIP library loads and a GPU is visible (reference: :64-82 checks its pip imports)."""
    try:
        return _lib.load().fw_device_count() > 0
    except FramewrightHipError:
        return False


def _load_checkpoint(cfg: PyTorchESRGANConfig, num_block: int, scale: int) -> Mapping[str, object]:
    path = Path(cfg.model_path) if cfg.model_path else \
        Path(os.environ.get("FRAMEWRIGHT_MODEL_DIR", str(DEFAULT_MODEL_DIR))) / MODEL_FILES[cfg.model_name]
    if path.exists():
        import torch
        return torch.load(str(path), map_location="cpu", weights_only=True)
    if os.environ.get("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS") == "1":
        logger.warning("using seeded synthetic weights for %s (no checkpoint at %s)", cfg.model_name, path)
        return synthetic_rrdbnet_state(num_block, scale)
    raise FileNotFoundError(
        f"Real-ESRGAN weights not found at {path}. Download is not attempted (the reference passes a URL to "
        f"RealESRGANer, pytorch_realesrgan.py:106); place the .pth there, set PyTorchESRGANConfig.model_path, or "
        f"set FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS=1 for seeded synthetic weights.")


def get_upsampler(config: PyTorchESRGANConfig) -> HipRealESRGANer:
    """pytorch_realesrgan.py:85-173.  Unlike the reference's single global (which ignores later configs —
    SURVEY.md §8f lists that as a defect), instances are cached per (model, gpu, dtype, tiling)."""
    if config.model_name not in RRDB_MODELS:
        raise ValueError(f"Unknown model: {config.model_name}")
    num_block, netscale = RRDB_MODELS[config.model_name]
    tile = config.tile_size or 0  # 288 GB of HBM: "auto" means no tiling, as the reference picks for >=24 GB (:141)
    key = (config.model_name, config.gpu_id, config.dtype, config.model_path)
    with _UPSAMPLER_LOCK:
        up = _UPSAMPLERS.get(key)
        if up is None:
            state = _load_checkpoint(config, num_block, netscale)
            from . import srvgg
            if config.model_name in srvgg.SRVGG_MODELS and srvgg.is_srvgg_state_dict(state):
                # the published realesr-animevideov3 / realesr-general-x4v3 checkpoints are SRVGGNetCompact, not the
                # RRDBNet the reference declares for them (pytorch_realesrgan.py:119-128)
                num_conv, netscale = srvgg.SRVGG_MODELS[config.model_name]
                engine = srvgg.SRVGGNetEngine(num_conv, netscale, config.dtype, config.gpu_id)
            else:
                engine = RRDBNetEngine(num_block, netscale, config.dtype, config.gpu_id)
            engine.load_state_dict(state)
            up = HipRealESRGANer(netscale, engine, tile, config.tile_pad, config.pre_pad, config.half_precision,
                                 config.gpu_id)
            _UPSAMPLERS[key] = up
        with up._mu:   # enhance() reads them under the same lock
            up.tile_size, up.tile_pad, up.pre_pad = tile, config.tile_pad, config.pre_pad
        return up


def get_bg_upsampler(bg_upsampler: Optional[str] = "realesrgan", gpu_id: int = 0, dtype: str = "f16") -> Optional[HipRealESRGANer]:
    """The background upsampler GFPGAN is handed by the reference's face restorer (``FaceRestorer._get_bg_upsampler``,
    processors/face_restore.py:379-401): ``None`` for no / ``'none'`` upsampler, else ``RealESRGANer(scale=4, model_path=
    'RealESRGAN_x4plus.pth', model=RRDBNet(3, 3, 64, 23, 32, scale=4), tile=400, tile_pad=10, pre_pad=0, half=True)`` - here the
    x4plus engine of `get_upsampler` behind an upsampler object of its own (its tiling must not follow the frame upscaler's).  Like the
    reference it answers ``None`` instead of raising when the upsampler cannot be built."""
    if not bg_upsampler or bg_upsampler == "none":
        return None
    try:
        base = get_upsampler(PyTorchESRGANConfig(model_name="RealESRGAN_x4plus", scale_factor=4, gpu_id=gpu_id, dtype=dtype))
        return HipRealESRGANer(4, base.engine, tile=400, tile_pad=10, pre_pad=0, half=True, gpu_id=gpu_id)
    except Exception:   # noqa: BLE001 - face_restore.py:399-400
        return None


def _imread(path: Path) -> Optional[np.ndarray]:
    try:
        import cv2  # the reference's reader (pytorch_realesrgan.py:198)
        return cv2.imread(str(path), cv2.IMREAD_UNCHANGED)
    except ImportError:
        from PIL import Image
        try:
            im = Image.open(str(path))
            im.load()
        except Exception:
            return None
        a = np.asarray(im.convert("RGBA") if im.mode in ("RGBA", "LA", "P") and "A" in im.getbands() else
                       im.convert("L") if im.mode in ("L", "1") else im.convert("RGB"))
        if a.ndim == 3 and a.shape[2] == 3:
            return np.ascontiguousarray(a[:, :, ::-1])
        if a.ndim == 3 and a.shape[2] == 4:
            return np.ascontiguousarray(a[:, :, [2, 1, 0, 3]])
        return a


def _imwrite(path: Path, img: np.ndarray) -> None:
    try:
        import cv2
        cv2.imwrite(str(path), img)
    except ImportError:
        from PIL import Image
        if img.ndim == 3 and img.shape[2] == 3:
            img = img[:, :, ::-1]
        elif img.ndim == 3 and img.shape[2] == 4:
            img = img[:, :, [2, 1, 0, 3]]
        Image.fromarray(np.ascontiguousarray(img)).save(str(path))


def enhance_frame_pytorch(input_path: Path, output_path: Path, config: PyTorchESRGANConfig
                          ) -> Tuple[bool, Optional[str]]:
    """pytorch_realesrgan.py:176-247: never raises, returns ``(ok, message)``; an out-of-memory message contains
    "memory" so the caller's tile downshift (restorer.py:1746) still triggers."""
    try:
        config.validate()
        input_path, output_path = Path(input_path), Path(output_path)
        img = _imread(input_path)
        if img is None:
            return False, f"Failed to read image: {input_path}"
        upsampler = get_upsampler(config)
        output, _ = upsampler.enhance(img, outscale=config.scale_factor)
        _imwrite(output_path, output)
        if not output_path.exists():
            return False, "Output file was not created"
        return True, None
    except FramewrightOutOfMemory as e:
        clear_upsampler_cache()
        return False, (f"GPU out of memory: {e}\n"
                       f"Try: 1) Reduce tile_size, 2) Use smaller model, 3) Close other GPU applications")
    except Exception as e:  # noqa: BLE001 - reference contract: report, never raise
        logger.error(f"HIP Real-ESRGAN failed: {e}")
        return False, str(e)


def clear_upsampler_cache() -> None:
    """pytorch_realesrgan.py:250-260: `_UPSAMPLER = None` + `torch.cuda.empty_cache()`.  Like the reference this only DROPS the
    cache entries: a worker of the thread pool (restorer.py:1830-1973) that already holds the upsampler - queued on its lock while a
    sibling's frame ran out of memory - finishes its frame on the old object, and the native engine is destroyed when the last
    reference to it goes away (RRDBNetEngine.__del__ -> close()).  Round 2 closed the engines here, which failed those queued
    frames with "the engine has been closed" although they would have succeeded."""
    with _UPSAMPLER_LOCK:
        ups = list(_UPSAMPLERS.values())
        _UPSAMPLERS.clear()
    del ups          # engines nobody else holds are destroyed here, their device memory returned


NCNN_TO_PYTORCH_MODEL = {
    "realesrgan-x4plus": "RealESRGAN_x4plus",
    "realesrgan-x4plus-anime": "RealESRGAN_x4plus_anime_6B",
    "realesr-animevideov3": "realesr-animevideov3",
    "realesrnet-x4plus": "realesr-general-x4v3",
    "realesrgan-x2plus": "RealESRGAN_x2plus",
}


def convert_ncnn_model_name(ncnn_name: str) -> str:
    """pytorch_realesrgan.py:273-275."""
    return NCNN_TO_PYTORCH_MODEL.get(ncnn_name, "RealESRGAN_x4plus")
