"""TAP neural temporal denoise on MI355X behind the reference's class boundary (boundary B2).

Mirrors reference ``src/framewright/processors/tap_denoise.py``: ``TAPModel`` (:64-91), ``TAPDenoiseConfig`` (:95-131),
``TAPDenoiseResult`` (:134-152), ``TAPDenoiser`` (:155-687: ``is_available``, ``denoise_frames``,
``_denoise_frame_tiled``, ``_denoise_with_temporal_window``, ``clear_cache``).

What runs where: the NAFNet forward, the pre/post-processing, the tile ramp blend, the temporal weighted average and
the strength blend all run in libframewright_hip.so on device-resident uint8 frames; this module walks directories,
keeps the per-frame cache and orders the calls.  Two deliberate differences from the reference's *schedule* (not its
results): every frame is denoised ONCE and reused by the up-to-5 windows that contain it (the reference re-denoises
it for every centre, tap_denoise.py:509-519; the forward is a pure function of the frame so the output is identical),
and frames stay in HBM between the stages.

Only ``TAPModel.NAFNET`` is accelerated; Restormer/TAP raise (SURVEY.md §8f lists Restormer as "next").
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import shutil
import time
from dataclasses import dataclass
from enum import Enum
from pathlib import Path
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import FramewrightHipError
from .realesrgan import _imread, _imwrite, _to_numpy
from .synth import nafnet_tensor_shapes, synthetic_nafnet_state

logger = logging.getLogger(__name__)

NAFNET_ARGS = dict(width=64, middle_blk_num=12, enc_blk_nums=(2, 2, 4, 8), dec_blk_nums=(2, 2, 2, 2))  # :340-346


class TAPModel(Enum):
    RESTORMER = "restormer"
    NAFNET = "nafnet"
    TAP = "tap"


@dataclass
class TAPDenoiseConfig:
    """Field-for-field the reference dataclass (tap_denoise.py:95-131), RESTORMER default included (:110)."""
    model: TAPModel = TAPModel.RESTORMER
    temporal_window: int = 5
    strength: float = 1.0
    preserve_grain: bool = False
    half_precision: bool = True
    tile_size: int = 512
    tile_overlap: int = 32
    gpu_id: int = 0
    batch_size: int = 1
    dtype: str = "f16"   # operand type of the HIP kernels; half_precision=True in the reference means fp16

    def __post_init__(self) -> None:
        if isinstance(self.model, str):
            self.model = TAPModel(self.model)
        if self.temporal_window < 1:
            raise ValueError(f"temporal_window must be >= 1, got {self.temporal_window}")
        if not 0.0 <= self.strength <= 1.0:
            raise ValueError(f"strength must be 0-1, got {self.strength}")
        if self.tile_size is not None and self.tile_size < 0:
            raise ValueError(f"tile_size must be >= 0, got {self.tile_size}")
        if self.tile_overlap < 0:
            raise ValueError(f"tile_overlap must be >= 0, got {self.tile_overlap}")


@dataclass
class TAPDenoiseResult:
    frames_processed: int = 0
    frames_failed: int = 0
    output_dir: Optional[Path] = None
    avg_psnr_improvement: float = 0.0
    processing_time_seconds: float = 0.0
    peak_vram_mb: int = 0
    model_used: Optional[str] = None


class NAFNetEngine:
    """One NAFNet resident on one GPU (owner of an ``fw_nafnet*``)."""

    def __init__(self, width: int = 64, middle_blk_num: int = 12, enc_blk_nums: Sequence[int] = (2, 2, 4, 8),
                 dec_blk_nums: Sequence[int] = (2, 2, 2, 2), dtype: str = "f16", device_id: int = 0):
        self._lib = _lib.load()
        _lib.require_gpu()
        if len(enc_blk_nums) != len(dec_blk_nums):
            raise ValueError("enc_blk_nums and dec_blk_nums must have the same length")
        self.args = dict(width=int(width), middle_blk_num=int(middle_blk_num), enc_blk_nums=tuple(enc_blk_nums),
                         dec_blk_nums=tuple(dec_blk_nums))
        self.dtype, self.device_id = dtype, int(device_id)
        n = len(enc_blk_nums)
        enc = (C.c_int * n)(*enc_blk_nums)
        dec = (C.c_int * n)(*dec_blk_nums)
        h = C.c_void_p()
        _lib.check(self._lib.fw_nafnet_create(self.device_id, width, middle_blk_num, enc, dec, n, _lib.DTYPES[dtype],
                                              C.byref(h)))
        self._h = h

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        """NAFNet state-dict; a checkpoint dict with ``params`` / ``state_dict`` is unwrapped as the reference does
        (tap_denoise.py:348-355)."""
        if "params" in state:
            state = state["params"]  # type: ignore[assignment]
        elif "state_dict" in state:
            state = state["state_dict"]  # type: ignore[assignment]
        kept = {}
        for key, shape in nafnet_tensor_shapes(**self.args):
            if key not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {key}")
            a = np.ascontiguousarray(_to_numpy(state[key]), dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{key}: expected shape {shape}, got {a.shape}")
            _lib.check(self._lib.fw_nafnet_set_tensor(self._h, key.encode(), C.c_void_p(a.ctypes.data), a.size))
            kept[key] = a
        _lib.check(self._lib.fw_nafnet_finalize(self._h))
        self._state = kept

    def clone(self) -> "NAFNetEngine":
        """A second handle with the same weights and its own workspace, so that two forwards can be in flight on two
        streams (the tiled TAP path runs several tiles concurrently)."""
        if getattr(self, "_state", None) is None:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "NAFNetEngine.clone: no weights loaded")
        e = NAFNetEngine(dtype=self.dtype, device_id=self.device_id, **self.args)
        e.load_state_dict(self._state)
        return e

    def denoise(self, frame_bgr: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(frame_bgr)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR frame")
        out = np.empty_like(f)
        _lib.check(self._lib.fw_nafnet_denoise_u8(self._h, C.c_void_p(f.ctypes.data), _lib.FW_HOST, f.shape[0], f.shape[1],
                                                  C.c_void_p(out.ctypes.data), _lib.FW_HOST, None, None))
        return out

    @_lib.on_tensor_device
    def denoise_device(self, frame, out=None, out_rgb_f32=None, stream: Optional[int] = None):
        """torch.uint8 CUDA tensor H x W x 3 -> same shape; asynchronous on torch's current stream."""
        import torch
        if frame.dtype != torch.uint8 or not frame.is_cuda or frame.dim() != 3 or frame.shape[2] != 3 or \
                not frame.is_contiguous():
            raise ValueError("denoise_device expects a contiguous uint8 CUDA tensor H x W x 3")
        h, w = int(frame.shape[0]), int(frame.shape[1])
        if out is None and out_rgb_f32 is None:
            out = torch.empty_like(frame)
        if stream is None:
            stream = torch.cuda.current_stream(frame.device).cuda_stream
        _lib.check(self._lib.fw_nafnet_denoise_u8(
            self._h, C.c_void_p(frame.data_ptr()), _lib.FW_DEVICE, h, w,
            C.c_void_p(out.data_ptr()) if out is not None else None, _lib.FW_DEVICE,
            C.c_void_p(out_rgb_f32.data_ptr()) if out_rgb_f32 is not None else None, C.c_void_p(stream)))
        return out if out is not None else out_rgb_f32

    def flops(self, h: int, w: int) -> float:
        return float(self._lib.fw_nafnet_flops(self._h, h, w))

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.fw_nafnet_destroy(h)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


# ---- pure host logic of the driver (also used by the multi-GPU sharding) ------------------------------------------
def tile_grid(h: int, w: int, tile_size: int, overlap: int) -> List[tuple]:
    """Tile origins in the reference's loop order (tap_denoise.py:435-450)."""
    stride = tile_size - overlap
    if stride <= 0:
        raise ValueError("tile_overlap must be smaller than tile_size")
    h_tiles = max(1, (h - overlap) // stride + (1 if (h - overlap) % stride else 0))
    w_tiles = max(1, (w - overlap) // stride + (1 if (w - overlap) % stride else 0))
    return [(min(i * stride, h - tile_size), min(j * stride, w - tile_size)) for i in range(h_tiles) for j in range(w_tiles)]


def temporal_window(n_frames: int, center_idx: int, window: int):
    """(start, end, normalised weights) — tap_denoise.py:508-528."""
    half = window // 2
    start, end = max(0, center_idx - half), min(n_frames, center_idx + half + 1)
    ws = [1.0 / (1.0 + abs(i - center_idx) * 0.5) for i in range(start, end)]
    tot = sum(ws)
    return start, end, [x / tot for x in ws]


class TAPDenoiser:
    """Drop-in for the reference class (tap_denoise.py:155)."""

    DEFAULT_MODEL_DIR = Path.home() / ".framewright" / "models" / "tap"
    MODEL_FILES = {
        TAPModel.RESTORMER: "restormer_deraining.pth",
        TAPModel.NAFNET: "NAFNet-SIDD-width64.pth",
        TAPModel.TAP: "tap_restormer.pth",
    }
    MODEL_VRAM = {TAPModel.RESTORMER: 4000, TAPModel.NAFNET: 2000, TAPModel.TAP: 6000}

    def __init__(self, config: Optional[TAPDenoiseConfig] = None, model_dir: Optional[Path] = None,
                 engine: Optional[NAFNetEngine] = None):
        self.config = config or TAPDenoiseConfig()
        self.model_dir = Path(model_dir) if model_dir else self.DEFAULT_MODEL_DIR
        self._engine = engine
        self._lib = None

    # -- availability / model ------------------------------------------------------------------------
    def is_available(self) -> bool:
        if self.config.model not in (TAPModel.NAFNET, TAPModel.RESTORMER, TAPModel.TAP):
            return False
        try:
            return _lib.load().fw_device_count() > 0
        except FramewrightHipError:
            return False

    def _get_model_path(self) -> Optional[Path]:
        p = self.model_dir / self.MODEL_FILES[self.config.model]
        return p if p.exists() else None

    def _load_model(self) -> None:
        if self._engine is not None:
            return
        # TAPModel.TAP: "TAP framework uses Restormer backbone ... fall back to Restormer" (tap_denoise.py:366-371)
        restormer = self.config.model in (TAPModel.RESTORMER, TAPModel.TAP)
        if restormer:
            from .restormer import RESTORMER_ARGS, RestormerEngine, synthetic_restormer_state
            eng = RestormerEngine(dtype=self.config.dtype, device_id=self.config.gpu_id, **RESTORMER_ARGS)
        else:
            eng = NAFNetEngine(dtype=self.config.dtype, device_id=self.config.gpu_id, **NAFNET_ARGS)
        path = self._get_model_path()
        if path is not None:
            import torch
            eng.load_state_dict(torch.load(str(path), map_location="cpu", weights_only=True))
        elif os.environ.get("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS") == "1":
            logger.warning("using seeded synthetic %s weights (no checkpoint under %s)", self.config.model.value, self.model_dir)
            eng.load_state_dict(synthetic_restormer_state(**RESTORMER_ARGS) if restormer else synthetic_nafnet_state(**NAFNET_ARGS))
        else:
            eng.close()
            raise FileNotFoundError(f"{self.config.model.value} weights not found: "
                                    f"{self.model_dir / self.MODEL_FILES[self.config.model]}")
        self._engine = eng

    # -- device helpers --------------------------------------------------------------------------------
    def _stream(self):
        # every caller runs under on_tensor_device: the current device is the one that owns the frames
        import torch
        return C.c_void_p(torch.cuda.current_stream(torch.cuda.current_device()).cuda_stream)

    @_lib.on_tensor_device
    def _denoise_frame_tiled_device(self, frame):
        """tap_denoise.py:417-488 on a uint8 CUDA tensor; returns a uint8 CUDA tensor."""
        import torch
        lib = _lib.load()
        h, w = int(frame.shape[0]), int(frame.shape[1])
        ts, ov = self.config.tile_size, self.config.tile_overlap
        if ts == 0 or ts is None or (h <= ts and w <= ts):
            return self._engine.denoise_device(frame)
        if h < ts or w < ts:
            # the reference slices frame[y1:y2] with a negative y1 here and fails on the shape mismatch (its per-frame
            # except then copies the input through); refuse explicitly instead of emulating the crash
            raise ValueError(f"frame {w}x{h} is smaller than tile_size {ts} in one dimension")
        acc = torch.zeros((h, w, 3), dtype=torch.float32, device=frame.device)
        wsum = torch.zeros((h, w), dtype=torch.float32, device=frame.device)
        tiles = tile_grid(h, w, ts, ov)
        # A 512x512 tile fills a fraction of the chip (the deep levels are 32x32 pixels), so up to TILE_STREAMS tiles run
        # concurrently, each on its own stream with its own engine clone (= its own workspace).  The ramp-blend
        # accumulation stays on the caller's stream in tile order: float adds in the reference's order, bit for bit.
        k = max(1, min(len(tiles), self.TILE_STREAMS))
        workers = self._tile_workers(k, frame.device, ts)
        main = torch.cuda.current_stream(frame.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        start = torch.cuda.Event()
        start.record(main)           # the frame (and acc / wsum) are ready once the caller's stream gets here
        done = [None] * len(tiles)
        for i, (y1, x1) in enumerate(tiles):
            wk = workers[i % k]
            stw = wk["stream"]
            stw.wait_event(start if wk["free"] is None else wk["free"])
            sp = C.c_void_p(stw.cuda_stream)
            _lib.check(lib.fw_u8_crop(p(frame), h, w, y1, x1, ts, ts, p(wk["tile"]), sp))
            with torch.cuda.stream(stw):   # the engines queue on torch's current stream (and allocate their scratch on it)
                wk["engine"].denoise_device(wk["tile"], out=wk["out"])
            done[i] = torch.cuda.Event()
            done[i].record(stw)
            # consume in order as soon as this worker's slot is needed again (or at the end)
            if i >= k - 1:
                j = i - (k - 1)
                self._blend_tile(lib, main, done[j], workers[j % k], acc, wsum, h, w, tiles[j], ts, ov)
        for j in range(max(0, len(tiles) - (k - 1)), len(tiles)):
            self._blend_tile(lib, main, done[j], workers[j % k], acc, wsum, h, w, tiles[j], ts, ov)
        for wk in workers:
            wk["free"] = None        # the next frame starts from the caller's stream again
        out = torch.empty_like(frame)
        _lib.check(lib.fw_tile_blend_finish(p(acc), p(wsum), h, w, p(out), C.c_void_p(main.cuda_stream)))
        return out

    TILE_STREAMS = int(os.environ.get("FW_TAP_TILE_STREAMS", "3"))   # tools/time_tiled.py: 1 / 2 / 3 / 4 / 6 streams = 72 / 42 / 32 / 44 / 33 ms (NAFNet), 169 / 168 / 114 / 115 / 121 (Restormer)

    def _tile_workers(self, k: int, device, ts: int):
        import torch
        ws = getattr(self, "_workers", None)
        if ws is None:
            ws = self._workers = []
        while len(ws) < k:
            eng = self._engine if not ws else self._engine.clone()
            ws.append({"engine": eng, "stream": torch.cuda.Stream(device=device), "free": None, "tile": None, "out": None})
        for wk in ws[:k]:
            if wk["tile"] is None or wk["tile"].shape[0] != ts:
                wk["tile"] = torch.empty((ts, ts, 3), dtype=torch.uint8, device=device)
                wk["out"] = torch.empty_like(wk["tile"])
        return ws[:k]

    @staticmethod
    def _blend_tile(lib, main, done_ev, wk, acc, wsum, h, w, origin, ts, ov):
        import torch
        main.wait_event(done_ev)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.fw_tile_blend_accumulate(p(acc), p(wsum), h, w, p(wk["out"]), origin[0], origin[1], ts, ts, ov,
                                                C.c_void_p(main.cuda_stream)))
        wk["free"] = torch.cuda.Event()
        wk["free"].record(main)      # the worker may overwrite its tile buffers after this point

    @_lib.on_tensor_device
    def _temporal_average_device(self, denoised: Sequence, weights: Sequence[float]):
        import torch
        lib = _lib.load()
        k = len(denoised)
        ptrs = (C.c_void_p * k)(*[d.data_ptr() for d in denoised])
        ws = (C.c_float * k)(*[float(np.float32(x)) for x in weights])
        out = torch.empty_like(denoised[0])
        _lib.check(lib.fw_temporal_average_u8(ptrs, ws, k, denoised[0].numel(), C.c_void_p(out.data_ptr()), self._stream()))
        return out

    @_lib.on_tensor_device
    def _strength_blend_device(self, original, denoised):
        import torch
        out = torch.empty_like(original)
        _lib.check(_lib.load().fw_strength_blend_u8(C.c_void_p(original.data_ptr()), C.c_void_p(denoised.data_ptr()),
                                                    float(self.config.strength), original.numel(),
                                                    C.c_void_p(out.data_ptr()), self._stream()))
        return out

    @_lib.on_tensor_device
    def _grain_addback_device(self, original, denoised, factor: float = 0.3):
        """preserve_grain (tap_denoise.py:621-632): high-pass of the original's gray image added back, on the device."""
        import torch
        h, w = int(original.shape[0]), int(original.shape[1])
        out = torch.empty_like(original)
        scratch = torch.empty((h, w), dtype=torch.int16, device=original.device)
        _lib.check(_lib.load().fw_grain_addback_u8(C.c_void_p(original.data_ptr()), C.c_void_p(denoised.data_ptr()), h, w,
                                                   float(factor), C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()),
                                                   self._stream()))
        return out

    # -- reference method names (numpy in / numpy out) -------------------------------------------------
    def _denoise_frame_tiled(self, frame: np.ndarray) -> np.ndarray:
        import torch
        self._load_model()
        return self._denoise_frame_tiled_device(torch.from_numpy(np.ascontiguousarray(frame)).cuda(self.config.gpu_id)).cpu().numpy()

    def _denoise_with_temporal_window(self, frames: List[np.ndarray], center_idx: int) -> np.ndarray:
        return self.denoise_clip(frames, only=[center_idx])[0]

    def denoise_clip(self, frames: Sequence[Optional[np.ndarray]], only: Optional[Sequence[int]] = None,
                     halo_before: Sequence[np.ndarray] = (), halo_after: Sequence[np.ndarray] = ()) -> List[Optional[np.ndarray]]:
        """In-memory form of the hot loop (tap_denoise.py:603-618): temporal window + strength blend for every frame
        (or the indices in ``only``).  ``halo_before`` / ``halo_after`` are already DENOISED uint8 neighbour frames
        owned by adjacent ranks (multi-GPU block partition); clip ends clamp like :510-511."""
        out = self.denoise_clip_device(frames, only=only, halo_before=halo_before, halo_after=halo_after)
        return [None if d is None else d.cpu().numpy() for d in out]

    @_lib.on_tensor_device
    def denoise_clip_device(self, frames: Sequence, only: Optional[Sequence[int]] = None, halo_before: Sequence = (),
                            halo_after: Sequence = (), denoised: Optional[Sequence] = None) -> List:
        """``denoise_clip`` with the results left in HBM (uint8 CUDA tensors): the stage hand-off of SURVEY.md §8(f) item 1.
        ``frames`` / halos may be numpy arrays (uploaded on first use) or uint8 CUDA tensors.  ``denoised[i]`` (optional) is
        frame i already through the network (``denoise_only_device``): the multi-GPU path denoises a block once, ships its
        edge frames to the neighbours and then runs the windows here without a second forward."""
        import torch
        self._load_model()
        dev = torch.device("cuda", self.config.gpu_id)
        n = len(frames)
        hb, ha = len(halo_before), len(halo_after)
        cache: Dict[int, object] = {}

        def up(a):
            if isinstance(a, np.ndarray):
                return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            return a.contiguous()

        def den(i):  # global index in [-hb, n + ha)
            if i not in cache:
                if i < 0:
                    cache[i] = up(halo_before[hb + i])
                elif i >= n:
                    cache[i] = up(halo_after[i - n])
                elif denoised is not None and denoised[i] is not None:
                    cache[i] = denoised[i]
                else:
                    # a miss pulls the next FW_TAP_FRAME_STREAMS - 1 frames of the clip through the network with it (they are needed by the
                    # windows that follow, and whole-frame forwards overlap on their own streams: denoise_only_device)
                    look = _lib.side_streams("FW_TAP_FRAME_STREAMS", 2) if only is None else 1
                    batch = [j for j in range(i, min(n, i + look))
                             if j not in cache and frames[j] is not None and not (denoised is not None and denoised[j] is not None)]
                    for j, o in zip(batch, self.denoise_only_device([up(frames[j]) for j in batch])):
                        cache[j] = o
            return cache[i]

        out: List = []
        half = self.config.temporal_window // 2
        for i in (range(n) if only is None else only):
            if frames[i] is None:
                out.append(None)
                continue
            if self.config.temporal_window <= 1:
                d = den(i)
            else:
                lo, hi = max(-hb, i - half), min(n + ha, i + half + 1)
                idx = [j for j in range(lo, hi) if j < 0 or j >= n or frames[j] is not None]
                ws = [1.0 / (1.0 + abs(j - i) * 0.5) for j in idx]
                tot = sum(ws)
                d = self._temporal_average_device([den(j) for j in idx], [x / tot for x in ws])
            if self.config.strength < 1.0:
                d = self._strength_blend_device(up(frames[i]), d)
            if self.config.preserve_grain:
                d = self._grain_addback_device(up(frames[i]), d)
            out.append(d)
            for j in [k for k in cache if k < i - half]:  # frames that no later window needs
                del cache[j]
        return out

    @_lib.on_tensor_device
    def denoise_only_device(self, frames: Sequence) -> List:
        """Every frame (uint8 CUDA tensors) through the network with the configured tiling, no temporal window: what a rank
        computes once per owned frame before the halo exchange."""
        self._load_model()
        frames = [f.contiguous() for f in frames]
        ts = self.config.tile_size
        whole = all(ts == 0 or ts is None or (int(f.shape[0]) <= ts and int(f.shape[1]) <= ts) for f in frames)
        k = min(len(frames), _lib.side_streams("FW_TAP_FRAME_STREAMS", 2))
        if k <= 1 or not whole:
            return [self._denoise_frame_tiled_device(f) for f in frames]
        # Whole-frame forwards of independent frames: the low-resolution levels of the U-Net (a few thousand pixels, more than half of
        # a NAFNet forward's time) occupy a fraction of the chip, so FW_TAP_FRAME_STREAMS (default 2) frames are in flight at once, each on its own
        # stream with its own engine clone (= its own workspace).  Every frame goes through the same kernels with the same
        # launch geometry as alone: identical outputs.
        import torch
        device = frames[0].device
        workers = self._frame_workers(k, device)
        main = torch.cuda.current_stream(device)
        outs = _lib.empty_like_many(frames)
        start = torch.cuda.Event()
        start.record(main)           # the frames and the output buffers are ready once the caller's stream gets here
        for i, f in enumerate(frames):
            wk = workers[i % k]
            if i < k:
                wk["stream"].wait_event(start)
            with torch.cuda.stream(wk["stream"]):
                wk["engine"].denoise_device(f, out=outs[i])
            f.record_stream(wk["stream"])
            outs[i].record_stream(wk["stream"])
        for wk in workers:
            ev = torch.cuda.Event()
            ev.record(wk["stream"])
            main.wait_event(ev)
        return outs

    def _frame_workers(self, k: int, device):
        import torch
        ws = getattr(self, "_fworkers", None)
        if ws is None:
            ws = self._fworkers = []
        while len(ws) < k:
            eng = self._engine if not ws else self._engine.clone()
            ws.append({"engine": eng, "stream": torch.cuda.Stream(device=device)})
        return ws[:k]

    def denoise_halo_frames(self, frames: Sequence[np.ndarray], count: int, head: bool) -> List[np.ndarray]:
        """The first/last ``count`` frames of this rank's block, denoised (tiled) but not yet temporally averaged —
        what a neighbouring rank needs as halo."""
        import torch
        self._load_model()
        dev = torch.device("cuda", self.config.gpu_id)
        idx = list(range(min(count, len(frames)))) if head else list(range(max(0, len(frames) - count), len(frames)))
        return [self._denoise_frame_tiled_device(torch.from_numpy(np.ascontiguousarray(frames[i])).to(dev)).cpu().numpy()
                for i in idx]

    def denoise_frames(self, input_dir: Path, output_dir: Path,
                       progress_callback: Optional[Callable[[float], None]] = None) -> TAPDenoiseResult:
        """tap_denoise.py:536-677."""
        result = TAPDenoiseResult(model_used=self.config.model.value)
        t0 = time.time()
        if not self.is_available():
            logger.error("TAP denoising not available")
            return result
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        result.output_dir = output_dir
        input_dir = Path(input_dir)
        files = sorted(input_dir.glob("*.png")) or sorted(input_dir.glob("*.jpg"))
        if not files:
            logger.warning(f"No frames found in {input_dir}")
            return result
        try:
            self._load_model()
        except Exception as e:  # noqa: BLE001 - reference contract: log and return the empty result
            logger.error(f"Failed to load TAP model: {e}")
            return result
        frames: List[Optional[np.ndarray]] = []
        for f in files:
            img = _imread(f)
            frames.append(img[:, :, :3] if img is not None and img.ndim == 3 else None)
        gains = []
        total = len(files)
        try:  # whole clip with the per-frame cache (each frame denoised once)
            outs: List[Optional[np.ndarray]] = self.denoise_clip(frames)
        except Exception as e:  # noqa: BLE001 - fall back to frame-at-a-time so one bad frame does not lose the clip
            logger.error(f"clip-level denoise failed ({e}); retrying frame by frame")
            outs = []
            for i in range(total):
                try:
                    outs.append(self.denoise_clip(frames, only=[i])[0] if frames[i] is not None else None)
                except Exception as e2:  # noqa: BLE001
                    logger.error(f"Failed to denoise {files[i]}: {e2}")
                    outs.append(None)
        for i, f in enumerate(files):
            den = outs[i]
            if frames[i] is None:
                logger.warning(f"Skipping invalid frame: {f}")
                result.frames_failed += 1
            elif den is None:
                result.frames_failed += 1
                try:
                    shutil.copy2(f, output_dir / f.name)  # reference fallback: copy the original (:653-658)
                except Exception:  # noqa: BLE001
                    pass
            else:
                _imwrite(output_dir / f.name, den)
                result.frames_processed += 1
                nb, na = np.std(frames[i].astype(np.float32)), np.std(den.astype(np.float32))
                if nb > 0:
                    gains.append(20 * np.log10(nb / max(na, 1)))
            if progress_callback:
                progress_callback((i + 1) / total)
        result.processing_time_seconds = time.time() - t0
        if gains:
            result.avg_psnr_improvement = float(np.mean(gains))
        return result

    def clear_cache(self) -> None:
        for wk in (getattr(self, "_workers", None) or []) + (getattr(self, "_fworkers", None) or []):
            if wk["engine"] is not self._engine:
                wk["engine"].close()
        self._workers = None
        self._fworkers = None
        if self._engine is not None:
            self._engine.close()
            self._engine = None


# ---- thin wrappers of the reference module (tap_denoise.py:689-1080), kept name for name so that the module can stand in for it ----
class MotionLevel(Enum):
    """processors/scene_intelligence.py:38-44 (imported by the reference's tap_denoise module)."""
    STATIC = "static"
    MINIMAL = "minimal"
    MODERATE = "moderate"
    HIGH = "high"
    EXTREME = "extreme"


class AutoTAPDenoiser:
    """tap_denoise.py:689-768: tries TAP > Restormer > NAFNet by `MODEL_VRAM` against the device's memory."""

    def __init__(self, strength: float = 1.0, preserve_grain: bool = False, model_dir: Optional[Path] = None, gpu_id: int = 0):
        self.strength, self.preserve_grain, self.model_dir, self.gpu_id = strength, preserve_grain, model_dir, gpu_id
        self._denoiser: Optional[TAPDenoiser] = None

    def _select_best_model(self) -> Optional[TAPModel]:
        available_vram = 0
        try:
            import torch
            if torch.cuda.is_available():
                available_vram = torch.cuda.get_device_properties(self.gpu_id).total_memory // (1024 * 1024)
        except Exception:  # noqa: BLE001
            pass
        for model in (TAPModel.TAP, TAPModel.RESTORMER, TAPModel.NAFNET):
            if available_vram >= TAPDenoiser.MODEL_VRAM.get(model, 4000):
                cfg = TAPDenoiseConfig(model=model, strength=self.strength, preserve_grain=self.preserve_grain, gpu_id=self.gpu_id)
                if TAPDenoiser(cfg, self.model_dir).is_available():
                    return model
        return None

    def denoise_frames(self, input_dir: Path, output_dir: Path,
                       progress_callback: Optional[Callable[[float], None]] = None) -> TAPDenoiseResult:
        best = self._select_best_model()
        if best is None:
            logger.error("No TAP denoising model available")
            return TAPDenoiseResult()
        logger.info(f"Auto-selected TAP model: {best.value}")
        cfg = TAPDenoiseConfig(model=best, strength=self.strength, preserve_grain=self.preserve_grain, gpu_id=self.gpu_id)
        self._denoiser = TAPDenoiser(cfg, self.model_dir)
        return self._denoiser.denoise_frames(input_dir, output_dir, progress_callback)


def create_tap_denoiser(model: str = "restormer", strength: float = 1.0, preserve_grain: bool = False, gpu_id: int = 0) -> TAPDenoiser:
    """tap_denoise.py:771-795."""
    return TAPDenoiser(TAPDenoiseConfig(model=TAPModel(model), strength=strength, preserve_grain=preserve_grain, gpu_id=gpu_id))


@dataclass
class MotionAdaptiveConfig:
    """tap_denoise.py:797-821."""
    base_strength: float = 0.8
    motion_sensitivity: float = 0.5
    static_boost: float = 1.2
    motion_penalty: float = 0.4

    def __post_init__(self) -> None:
        if not 0.0 <= self.base_strength <= 1.0:
            raise ValueError(f"base_strength must be 0-1, got {self.base_strength}")
        if not 0.0 <= self.motion_sensitivity <= 1.0:
            raise ValueError(f"motion_sensitivity must be 0-1, got {self.motion_sensitivity}")
        if self.static_boost < 0:
            raise ValueError(f"static_boost must be >= 0, got {self.static_boost}")
        if self.motion_penalty < 0:
            raise ValueError(f"motion_penalty must be >= 0, got {self.motion_penalty}")


class MotionAdaptiveTAPDenoiser:
    """tap_denoise.py:824-1080: per-frame strength from the detected motion level, then the usual temporal window."""

    MOTION_MULTIPLIERS = {MotionLevel.STATIC: 1.2, MotionLevel.MINIMAL: 1.0, MotionLevel.MODERATE: 0.8, MotionLevel.HIGH: 0.6,
                          MotionLevel.EXTREME: 0.4}

    def __init__(self, config: Optional[MotionAdaptiveConfig] = None, tap_config: Optional[TAPDenoiseConfig] = None,
                 model_dir: Optional[Path] = None):
        self.config = config or MotionAdaptiveConfig()
        self.tap_config = tap_config or TAPDenoiseConfig()
        self.model_dir = model_dir
        self._denoiser: Optional[TAPDenoiser] = None

    def _ensure_denoiser(self) -> TAPDenoiser:
        if self._denoiser is None:
            self._denoiser = TAPDenoiser(self.tap_config, self.model_dir)
        return self._denoiser

    def is_available(self) -> bool:
        return self._ensure_denoiser().is_available()

    def get_motion_adjusted_strength(self, motion_level) -> float:
        """tap_denoise.py:878-904 (the arithmetic order is the reference's: the result is compared bit for bit)."""
        level = motion_level if isinstance(motion_level, MotionLevel) else MotionLevel(getattr(motion_level, "value", motion_level))
        base_multiplier = self.MOTION_MULTIPLIERS.get(level, 1.0)
        adjusted_multiplier = 1.0 + (base_multiplier - 1.0) * self.config.motion_sensitivity
        if level == MotionLevel.STATIC:
            adjusted_multiplier = min(adjusted_multiplier, self.config.static_boost)
        elif level == MotionLevel.EXTREME:
            adjusted_multiplier = max(adjusted_multiplier, self.config.motion_penalty)
        adjusted_strength = self.config.base_strength * adjusted_multiplier
        return max(0.0, min(1.0, adjusted_strength))

    def denoise_frames_motion_aware(self, input_dir: Path, output_dir: Path, motion_levels: Sequence,
                                    progress_callback: Optional[Callable[[float], None]] = None) -> TAPDenoiseResult:
        """tap_denoise.py:906-1074.  Each frame goes through the temporal window at strength 1 and is then blended with its
        original at the motion-adjusted strength (:1008-1013); a missing motion level counts as MODERATE (:950-961)."""
        result = TAPDenoiseResult(model_used=self.tap_config.model.value)
        t0 = time.time()
        den = self._ensure_denoiser()
        if not den.is_available():
            logger.error("TAP denoising not available")
            return result
        output_dir, input_dir = Path(output_dir), Path(input_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        result.output_dir = output_dir
        files = sorted(input_dir.glob("*.png")) or sorted(input_dir.glob("*.jpg"))
        if not files:
            logger.warning(f"No frames found in {input_dir}")
            return result
        levels = list(motion_levels)
        if len(levels) != len(files):
            logger.warning(f"motion_levels count ({len(levels)}) doesn't match frame count ({len(files)}). Using MODERATE as default.")
            levels = (levels + [MotionLevel.MODERATE] * len(files))[:len(files)]
        try:
            den._load_model()
        except Exception as e:  # noqa: BLE001
            logger.error(f"Failed to load TAP model: {e}")
            return result
        frames: List[Optional[np.ndarray]] = []
        for f in files:
            img = _imread(f)
            frames.append(img[:, :, :3] if img is not None and img.ndim == 3 else None)
        saved, saved_grain = den.config.strength, den.config.preserve_grain
        want_grain = bool(self.tap_config.preserve_grain)   # (den.config may be the very same object)
        gains = []
        try:
            den.config.strength = 1.0                      # the window result un-blended, as _denoise_with_temporal_window returns it
            den.config.preserve_grain = False              # grain goes in after the motion-adjusted blend, with its own factor
            outs = den.denoise_clip_device(frames)
            for i, f in enumerate(files):
                if frames[i] is None or outs[i] is None:
                    logger.warning(f"Skipping invalid frame: {f}")
                    result.frames_failed += 1
                else:
                    s = self.get_motion_adjusted_strength(levels[i])
                    d = outs[i]
                    import torch
                    orig_dev = None
                    if s < 1.0:
                        den.config.strength = s
                        orig_dev = torch.from_numpy(np.ascontiguousarray(frames[i])).to(d.device)
                        d = den._strength_blend_device(orig_dev, d)
                    if want_grain:                          # tap_denoise.py:1015-1023: less grain for high motion
                        if orig_dev is None:
                            orig_dev = torch.from_numpy(np.ascontiguousarray(frames[i])).to(d.device)
                        d = den._grain_addback_device(orig_dev, d, 0.3 * (s / self.config.base_strength))
                    out = d.cpu().numpy()
                    _imwrite(output_dir / f.name, out)
                    result.frames_processed += 1
                    nb, na = np.std(frames[i].astype(np.float32)), np.std(out.astype(np.float32))
                    if nb > 0:
                        gains.append(20 * np.log10(nb / max(na, 1)))
                if progress_callback:
                    progress_callback((i + 1) / len(files))
        finally:
            den.config.strength, den.config.preserve_grain = saved, saved_grain
        result.processing_time_seconds = time.time() - t0
        if gains:
            result.avg_psnr_improvement = float(np.mean(gains))
        return result

    def clear_cache(self) -> None:
        if self._denoiser is not None:
            self._denoiser.clear_cache()
