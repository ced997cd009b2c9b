"""ORACLE — test infrastructure only (never imported by the product path).

numpy restatement of the TAP driver arithmetic in reference ``src/framewright/processors/tap_denoise.py``:

* ``preprocess`` / ``postprocess``            <- ``_preprocess_frame`` :373-397 / ``_postprocess_frame`` :399-415
  (cv2.cvtColor BGR<->RGB is the channel reversal; the uint8 cast TRUNCATES)
* ``denoise_frame_tiled``                     <- ``_denoise_frame_tiled`` :417-488
* ``denoise_with_temporal_window``            <- ``_denoise_with_temporal_window`` :490-534
* ``strength_blend``                          <- ``denoise_frames`` :613-618

The reference functions themselves need cv2 and a loaded third-party model and cannot run in this image, so these
are line-for-line transcriptions with the model passed in as a callable (``model(tensor NCHW) -> tensor``); their pure
integer parts (tile grid) are pinned in tests/test_tap_host.py against values computed by hand from the reference
formulas.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np
import torch

Model = Callable[[torch.Tensor], torch.Tensor]


def preprocess(frame_bgr: np.ndarray) -> torch.Tensor:
    frame = frame_bgr[:, :, ::-1]                       # cv2.COLOR_BGR2RGB
    frame = frame.astype(np.float32) / 255.0
    return torch.from_numpy(np.ascontiguousarray(np.transpose(frame, (2, 0, 1)))).unsqueeze(0)


def postprocess(t: torch.Tensor) -> np.ndarray:
    frame = t.squeeze(0).float().numpy()
    frame = np.transpose(frame, (1, 2, 0))
    frame = np.clip(frame * 255.0, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(frame[:, :, ::-1])      # cv2.COLOR_RGB2BGR


def tile_grid(h: int, w: int, tile_size: int, overlap: int):
    """[(y1, x1)] in the reference's loop order (:435-450)."""
    stride = tile_size - overlap
    h_tiles = max(1, (h - overlap) // stride + (1 if (h - overlap) % stride else 0))
    w_tiles = max(1, (w - overlap) // stride + (1 if (w - overlap) % stride else 0))
    return [(min(i * stride, h - tile_size), min(j * stride, w - tile_size)) for i in range(h_tiles) for j in range(w_tiles)]


def denoise_frame_tiled(model: Model, frame: np.ndarray, tile_size, overlap: int) -> np.ndarray:
    h, w = frame.shape[:2]
    if tile_size == 0 or tile_size is None or (h <= tile_size and w <= tile_size):
        with torch.no_grad():
            return postprocess(model(preprocess(frame)))
    output = np.zeros((h, w, 3), dtype=np.float32)
    weight = np.zeros((h, w, 1), dtype=np.float32)
    for y1, x1 in tile_grid(h, w, tile_size, overlap):
        y2, x2 = y1 + tile_size, x1 + tile_size
        with torch.no_grad():
            tile_result = postprocess(model(preprocess(frame[y1:y2, x1:x2]))).astype(np.float32)
        tile_weight = np.ones((tile_size, tile_size, 1), dtype=np.float32)
        if overlap > 0:
            ramp = np.linspace(0, 1, overlap)
            if y1 > 0:
                tile_weight[:overlap, :, :] *= ramp.reshape(-1, 1, 1)
            if y2 < h:
                tile_weight[-overlap:, :, :] *= ramp[::-1].reshape(-1, 1, 1)
            if x1 > 0:
                tile_weight[:, :overlap, :] *= ramp.reshape(1, -1, 1)
            if x2 < w:
                tile_weight[:, -overlap:, :] *= ramp[::-1].reshape(1, -1, 1)
        output[y1:y2, x1:x2] += tile_result * tile_weight
        weight[y1:y2, x1:x2] += tile_weight
    weight = np.maximum(weight, 1e-8)
    return (output / weight).astype(np.uint8)


def temporal_weights(n_frames: int, center_idx: int, temporal_window: int):
    """(start, end, normalised weights) of :508-528."""
    half = temporal_window // 2
    start, end = max(0, center_idx - half), min(n_frames, center_idx + half + 1)
    ws = [1.0 / (1.0 + abs(i - center_idx) * 0.5) for i in range(start, end)]
    tot = sum(ws)
    return start, end, [x / tot for x in ws]


def temporal_average(denoised: Sequence[np.ndarray], weights: Sequence[float]) -> np.ndarray:
    """:530-534 given the already denoised uint8 frames of the window."""
    fl = [d.astype(np.float32) for d in denoised]
    result = np.zeros_like(fl[0])
    for f, wgt in zip(fl, weights):
        result += f * wgt
    return result.astype(np.uint8)


def denoise_with_temporal_window(model: Model, frames: List[np.ndarray], center_idx: int, temporal_window: int,
                                 tile_size, overlap: int) -> np.ndarray:
    if temporal_window <= 1:
        return denoise_frame_tiled(model, frames[center_idx], tile_size, overlap)
    start, end, ws = temporal_weights(len(frames), center_idx, temporal_window)
    return temporal_average([denoise_frame_tiled(model, frames[i], tile_size, overlap) for i in range(start, end)], ws)


def strength_blend(original: np.ndarray, denoised: np.ndarray, strength: float) -> np.ndarray:
    if strength >= 1.0:
        return denoised
    o = original.astype(np.float32)
    d = denoised.astype(np.float32)
    return (o * (1 - strength) + d * strength).astype(np.uint8)
