"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing on the product path).

numpy restatement of the classical motion-compensated temporal denoise of the reference,
src/framewright/processors/temporal_denoise.py: `OpticalFlowEstimator.warp_frame` (:440-477), `_denoise_with_flow`
(:1521-1580) and `_denoise_simple` (:1582-1605).  `cv2.remap(INTER_LINEAR, BORDER_REFLECT_101)` on uint8 is restated from
OpenCV's published fixed-point algorithm (coordinates rounded half-to-even to 1/32 pixel, 15-bit coefficients,
(sum + 2**14) >> 15); cv2 is not installed in the build container, so that part is PARITY UNPINNED.  The dense flow
(cv2 Farneback) is an input here: the reference computes it on the CPU and it is outside the accelerated path.
"""
import numpy as np


def _reflect101(p, n):
    if n == 1:
        return np.zeros_like(p)
    p = p.copy()
    while True:
        bad = (p < 0) | (p >= n)
        if not bad.any():
            return p
        p = np.where(p < 0, -p, np.where(p >= n, 2 * n - p - 2, p))


def remap_linear_reflect101(frame, map_x, map_y):
    """cv2.remap(frame, map_x, map_y, INTER_LINEAR, borderMode=BORDER_REFLECT_101) for uint8 H x W x 3."""
    h, w = frame.shape[:2]
    sx = np.rint(map_x.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(map_y.astype(np.float32) * np.float32(32)).astype(np.int64)
    ix, iy, ax, ay = sx >> 5, sy >> 5, sx & 31, sy & 31
    x0, x1, y0, y1 = _reflect101(ix, w), _reflect101(ix + 1, w), _reflect101(iy, h), _reflect101(iy + 1, h)
    f = frame.astype(np.int64)
    w00, w01, w10, w11 = (32 - ax) * (32 - ay) * 32, ax * (32 - ay) * 32, (32 - ax) * ay * 32, ax * ay * 32
    out = (f[y0, x0] * w00[..., None] + f[y0, x1] * w01[..., None] + f[y1, x0] * w10[..., None] + f[y1, x1] * w11[..., None]
           + (1 << 14)) >> 15
    return out.astype(np.uint8)


def warp_frame(frame, flow_x, flow_y, inverse=False):
    """temporal_denoise.py:440-477."""
    h, w = frame.shape[:2]
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    if inverse:
        map_x, map_y = (x - flow_x).astype(np.float32), (y - flow_y).astype(np.float32)
    else:
        map_x, map_y = (x + flow_x).astype(np.float32), (y + flow_y).astype(np.float32)
    return remap_linear_reflect101(frame, map_x, map_y)


def denoise_with_flow(center_local_idx, window, flows, decay):
    """temporal_denoise.py:1521-1580.  window: list of frames; flows[i] = None (flow estimation failed: unaligned frame,
    temporal weight only) or dict(flow_x, flow_y, magnitude, confidence) for the neighbour i (ignored for the centre)."""
    h, w = window[0].shape[:2]
    accumulated = np.zeros((h, w, 3), dtype=np.float64)
    weight_sum = np.zeros((h, w), dtype=np.float64)
    for local_i, frame in enumerate(window):
        distance = abs(local_i - center_local_idx)
        if distance == 0:
            weight = np.ones((h, w), dtype=np.float64)
            aligned = frame
        elif flows[local_i] is not None:
            fl = flows[local_i]
            aligned = warp_frame(frame, fl["flow_x"], fl["flow_y"])
            temporal_weight = np.exp(-distance * decay)
            weight = temporal_weight * fl["confidence"]
            motion_mask = fl["magnitude"] > np.percentile(fl["magnitude"], 90)
            weight[motion_mask] *= 0.5
        else:
            aligned = frame
            weight = np.exp(-distance * decay) * np.ones((h, w))
        accumulated += aligned.astype(np.float64) * weight[:, :, np.newaxis]
        weight_sum += weight
    weight_sum = np.maximum(weight_sum, 1e-6)[:, :, np.newaxis]
    return (accumulated / weight_sum).astype(np.uint8)


def denoise_simple(window, decay):
    """temporal_denoise.py:1582-1605."""
    accumulated = np.zeros_like(window[0], dtype=np.float64)
    weight_sum = 0.0
    center_idx = len(window) // 2
    for i, frame in enumerate(window):
        weight = np.exp(-abs(i - center_idx) * decay)
        accumulated += frame.astype(np.float64) * weight
        weight_sum += weight
    return (accumulated / weight_sum).astype(np.uint8)
