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


# ---- preserve_grain (tap_denoise.py:621-632, motion-adaptive :1015-1023) ------------------------------------------------
# cv2 is absent: BGR2GRAY, GaussianBlur(sigma 3 -> ksize 19, the 8-bit bit-exact fixed-point path), subtract and add are
# restated from OpenCV's published algorithms - unpinned.  Written as OpenCV runs it (separate passes), unlike the kernels.
def gaussian_kernel_fixed_point(n: int = 19, sigma: float = 3.0, bits: int = 8):
    """getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED: integer taps that sum to 2**bits."""
    import math
    scale2x = -0.125 / (sigma * sigma)
    vals = [math.exp(scale2x * (x * x)) for x in range(1 - n, 0, 2)]
    total = 2 * sum(vals) + 1
    res, err, acc = [0] * n, 0.0, 0
    for i in range(n // 2):
        adj = vals[i] / total * (1 << bits) + err
        v0 = int(np.rint(adj))
        err = adj - v0
        res[i] = res[n - 1 - i] = v0
        acc += v0
    res[n // 2] = (1 << bits) - 2 * acc
    return np.array(res, np.int64)


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    idx = np.abs(idx)
    period = 2 * (n - 1)
    idx = idx % period
    return np.where(idx >= n, period - idx, idx)


def bgr2gray_u8(img: np.ndarray) -> np.ndarray:
    b, g, r = (img[:, :, c].astype(np.int64) for c in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def gaussian_blur_u8_sigma3(gray: np.ndarray) -> np.ndarray:
    k = gaussian_kernel_fixed_point()
    h, w = gray.shape
    g = gray.astype(np.int64)
    xs = _reflect101(np.arange(w)[:, None] + np.arange(19)[None, :] - 9, w)           # [w][19]
    hb = (g[:, xs] * k[None, None, :]).sum(axis=2)                                      # 8.8 sums, exact
    ys = _reflect101(np.arange(h)[:, None] + np.arange(19)[None, :] - 9, h)
    vb = (hb[ys, :] * k[None, :, None]).sum(axis=1)                                     # [h][19][w] -> [h][w], 16.16
    return np.minimum((vb + (1 << 15)) >> 16, 255).astype(np.uint8)


def grain_addback(original: np.ndarray, denoised: np.ndarray, factor: float = 0.3) -> np.ndarray:
    gray = bgr2gray_u8(original)
    blurred = gaussian_blur_u8_sigma3(gray)
    grain = np.maximum(gray.astype(np.int64) - blurred.astype(np.int64), 0).astype(np.uint8)      # cv2.subtract
    grain_3ch = np.repeat(grain[:, :, None], 3, axis=2)
    add = (grain_3ch * factor).astype(np.uint8)
    return np.minimum(denoised.astype(np.int64) + add.astype(np.int64), 255).astype(np.uint8)       # cv2.add
