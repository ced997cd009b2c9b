"""Classical motion-compensated temporal denoise on the device — the accumulate/warp half of the reference's
`processors/temporal_denoise.py`: `OpticalFlowEstimator.warp_frame` (:440-477),
`TemporalDenoiser._denoise_with_flow` (:1521-1580) and `_denoise_simple` (:1582-1605).

The dense optical flow is cv2's (Farneback / DIS) in the reference and stays a host computation: pass an estimator
(`flow_fn(frame, center) -> FlowField`); without one the flow-compensated method raises like the reference does without
OpenCV ("OpenCV required for optical flow estimation"), and the simple weighted average needs none.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib


@dataclass
class FlowField:
    """Field-for-field the reference dataclass (temporal_denoise.py:190-207)."""
    flow_x: np.ndarray
    flow_y: np.ndarray
    magnitude: np.ndarray
    confidence: np.ndarray
    frame_idx_from: int = 0
    frame_idx_to: int = 1


def _default_flow_fn(frame: np.ndarray, center: np.ndarray) -> FlowField:
    try:
        import cv2  # noqa: F401
    except ImportError:
        raise RuntimeError("OpenCV required for optical flow estimation")   # temporal_denoise.py:275-276
    raise RuntimeError("pass flow_fn: the dense-flow estimator is a host computation outside the accelerated path")


class DeviceTemporalAccumulator:
    """The float64 accumulate of `_denoise_with_flow` / `_denoise_simple` on one GPU."""

    def __init__(self, temporal_weight_decay: float = 0.5, gpu_id: int = 0, flow_fn: Optional[Callable] = None):
        self._lib = _lib.load()
        _lib.require_gpu()
        self.decay, self.gpu_id = float(temporal_weight_decay), int(gpu_id)
        self.flow_fn = flow_fn or _default_flow_fn

    def _dev(self):
        import torch
        return torch.device("cuda", self.gpu_id)

    def warp_frame(self, frame: np.ndarray, flow: FlowField, inverse: bool = False) -> np.ndarray:
        """`OpticalFlowEstimator.warp_frame`: cv2.remap(frame, grid +/- flow, INTER_LINEAR, BORDER_REFLECT_101)."""
        import torch
        dev = self._dev()
        h, w = frame.shape[:2]
        acc = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
        ws = torch.zeros((h, w), dtype=torch.float64, device=dev)
        self._accumulate(torch.from_numpy(np.ascontiguousarray(frame)).to(dev), flow, 1.0, None, None, inverse, acc, ws)
        return self._finish(acc, ws)

    @_lib.on_tensor_device
    def _accumulate(self, frame_dev, flow: Optional[FlowField], scale: float, wmap, thr, inverse, acc, ws) -> None:
        import torch
        dev = acc.device
        h, w = int(acc.shape[0]), int(acc.shape[1])
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        keep = []
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        fx = fy = mg = wm = None
        if flow is not None:
            fx, fy = f32(flow.flow_x), f32(flow.flow_y)
            keep += [fx, fy]
        if wmap is not None:
            wm = f32(wmap)
            keep.append(wm)
        if thr is not None:
            mg = f32(flow.magnitude)
            keep.append(mg)
        _lib.check(self._lib.fw_flow_accumulate_u8(p(frame_dev), p(fx), p(fy), p(wm), float(scale), p(mg),
                                                   float(thr) if thr is not None else 0.0, int(bool(inverse)), h, w, p(acc), p(ws), st))
        torch.cuda.current_stream(dev).synchronize()   # the uploaded maps above are temporaries

    @_lib.on_tensor_device
    def _finish(self, acc, ws) -> np.ndarray:
        import torch
        h, w = int(acc.shape[0]), int(acc.shape[1])
        out = torch.empty((h, w, 3), dtype=torch.uint8, device=acc.device)
        st = C.c_void_p(torch.cuda.current_stream(acc.device).cuda_stream)
        _lib.check(self._lib.fw_flow_accumulate_finish_u8(C.c_void_p(acc.data_ptr()), C.c_void_p(ws.data_ptr()), h, w,
                                                          C.c_void_p(out.data_ptr()), st))
        torch.cuda.synchronize(acc.device)
        return out.cpu().numpy()

    def denoise_with_flow(self, center_local_idx: int, window: Sequence[np.ndarray]) -> np.ndarray:
        """`_denoise_with_flow` (temporal_denoise.py:1521-1580) for the frames of one window."""
        import torch
        dev = self._dev()
        center = window[center_local_idx]
        h, w = center.shape[:2]
        acc = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
        ws = torch.zeros((h, w), dtype=torch.float64, device=dev)
        for local_i, frame in enumerate(window):
            distance = abs(local_i - center_local_idx)
            fd = torch.from_numpy(np.ascontiguousarray(frame)).to(dev)
            if distance == 0:
                self._accumulate(fd, None, 1.0, None, None, False, acc, ws)
                continue
            temporal = math.exp(-distance * self.decay)       # np.exp on a Python float: the same libm value
            try:
                flow = self.flow_fn(frame, center)
                thr = np.percentile(flow.magnitude, 90)       # host: the flow and its statistics are host data
                self._accumulate(fd, flow, temporal, flow.confidence, thr, False, acc, ws)
            except Exception:                                 # "Flow failed, using unaligned" (:1565-1569)
                self._accumulate(fd, None, temporal, None, None, False, acc, ws)
        return self._finish(acc, ws)

    def preserve_edges(self, original: np.ndarray, denoised: np.ndarray, edge_threshold: int = 30) -> np.ndarray:
        """`TemporalDenoiser._preserve_edges` (temporal_denoise.py:1636-1667): the original frame shows through a blurred,
        dilated Canny edge mask (thresholds ``edge_threshold`` and three times that, config default 30, :146).  uint8 BGR frames
        in and out; fw_preserve_edges_u8."""
        import torch
        if original.shape != denoised.shape or original.dtype != np.uint8 or denoised.dtype != np.uint8 or original.ndim != 3 or \
                original.shape[2] != 3:
            raise ValueError("preserve_edges expects two uint8 BGR frames of one size")
        dev = self._dev()
        h, w = original.shape[:2]
        with torch.cuda.device(dev):
            o = torch.from_numpy(np.ascontiguousarray(original)).to(dev)
            d = torch.from_numpy(np.ascontiguousarray(denoised)).to(dev)
            scratch = torch.empty(int(self._lib.fw_preserve_edges_scratch_bytes(h, w)), dtype=torch.uint8, device=dev)
            out = torch.empty_like(o)
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(self._lib.fw_preserve_edges_u8(C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()), h, w, float(edge_threshold),
                                                      float(edge_threshold * 3), C.c_void_p(scratch.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), st))
            torch.cuda.synchronize(dev)
        return out.cpu().numpy()

    def denoise_simple(self, window: Sequence[np.ndarray]) -> np.ndarray:
        """`_denoise_simple` (temporal_denoise.py:1582-1605).  The reference divides by a scalar weight sum; per-pixel sums
        of the same scalars give the same float64 quotient."""
        import torch
        dev = self._dev()
        h, w = window[0].shape[:2]
        acc = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
        ws = torch.zeros((h, w), dtype=torch.float64, device=dev)
        center_idx = len(window) // 2
        for i, frame in enumerate(window):
            self._accumulate(torch.from_numpy(np.ascontiguousarray(frame)).to(dev), None, math.exp(-abs(i - center_idx) * self.decay),
                             None, None, False, acc, ws)
        return self._finish(acc, ws)
