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


# ---- `_preserve_edges` (temporal_denoise.py:1636-1667) -------------------------------------------------------------------------
# gray = cv2.cvtColor(original, BGR2GRAY); edges = cv2.Canny(gray, t, 3t); edges = cv2.dilate(edges, ones(3, 3));
# mask = cv2.GaussianBlur(edges / 255, (5, 5), 0); result = (original * mask + denoised * (1 - mask)).astype(uint8)
# OpenCV is absent from the build container, so every cv2 call is restated from its published 8-bit algorithm (parity
# unpinned; known answers in tests/test_temporal_host.py):
#   BGR2GRAY : (B * 1868 + G * 9617 + R * 4899 + 2^13) >> 14                              (color_rgb.simd.hpp, 14-bit weights)
#   Canny    : Sobel 3x3 (BORDER_REPLICATE) -> L1 magnitude -> non-maximum suppression with the fixed-point tangent test
#              (TG22 = round(tan(22.5 deg) * 2^15) = 13573; horizontal / vertical / diagonal sectors, ">" on one side, ">=" on
#              the other for the axis-aligned sectors) -> hysteresis from the pixels above `high` through 8-connected pixels
#              above `low`                                                                  (imgproc/src/canny.cpp)
#   dilate   : 3x3 maximum; the border does not contribute (BORDER_CONSTANT with the morphology default value)
#   GaussianBlur((5, 5), 0) on float32: the fixed kernel [1, 4, 6, 4, 1] / 16 (small_gaussian_tab), rows then columns,
#              BORDER_REFLECT_101; symmetric form  c*k2 + (l1 + r1)*k1 + (l2 + r2)*k0  in float32
def bgr2gray_u8(img: np.ndarray) -> np.ndarray:
    b, g, r = (img[:, :, i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def sobel3_replicate(gray: np.ndarray):
    p = np.pad(gray.astype(np.int32), 1, mode="edge")
    h, w = gray.shape
    s = lambda dy, dx: p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    dx = (s(-1, 1) + 2 * s(0, 1) + s(1, 1)) - (s(-1, -1) + 2 * s(0, -1) + s(1, -1))
    dy = (s(1, -1) + 2 * s(1, 0) + s(1, 1)) - (s(-1, -1) + 2 * s(-1, 0) + s(-1, 1))
    return dx, dy


def canny_u8(gray: np.ndarray, low: float, high: float) -> np.ndarray:
    """cv2.Canny(gray, low, high) with the defaults (apertureSize 3, L2gradient False): 255 on edges, 0 elsewhere."""
    if low > high:
        low, high = high, low
    lo, hi = int(np.floor(low)), int(np.floor(high))
    dx, dy = sobel3_replicate(gray)
    h, w = gray.shape
    mag = np.zeros((h + 2, w + 2), np.int64)
    mag[1:-1, 1:-1] = np.abs(dx) + np.abs(dy)
    m = mag[1:-1, 1:-1]
    nb = lambda oy, ox: mag[1 + oy:1 + oy + h, 1 + ox:1 + ox + w]
    x, y = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22 = x * 13573
    tg67 = tg22 + (x << 16)
    horiz = y < tg22
    vert = ~horiz & (y > tg67)
    diag = ~horiz & ~vert
    s = np.where((dx ^ dy) < 0, -1, 1)
    up_d = np.where(s < 0, nb(-1, 1), nb(-1, -1))       # (y - 1, x - s)
    dn_d = np.where(s < 0, nb(1, -1), nb(1, 1))         # (y + 1, x + s)
    keep = (horiz & (m > nb(0, -1)) & (m >= nb(0, 1))) | (vert & (m > nb(-1, 0)) & (m >= nb(1, 0))) | (diag & (m > up_d) & (m > dn_d))
    cand = keep & (m > lo)
    strong = cand & (m > hi)
    # hysteresis: grow the strong set through 8-connected candidates until nothing changes
    edge = strong.copy()
    while True:
        p = np.pad(edge, 1)
        grown = np.zeros_like(edge)
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                grown |= p[1 + oy:1 + oy + h, 1 + ox:1 + ox + w]
        new = edge | (grown & cand)
        if np.array_equal(new, edge):
            break
        edge = new
    return (edge * 255).astype(np.uint8)


def dilate3x3_u8(img: np.ndarray) -> np.ndarray:
    p = np.pad(img, 1, mode="constant", constant_values=0)
    h, w = img.shape
    out = np.zeros_like(img)
    for oy in (0, 1, 2):
        for ox in (0, 1, 2):
            out = np.maximum(out, p[oy:oy + h, ox:ox + w])
    return out


def gaussian5_f32(img: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(float32 image, (5, 5), 0)."""
    f = np.float32
    k0, k1, k2 = f(0.0625), f(0.25), f(0.375)

    def one(a, axis):
        a = np.moveaxis(a, axis, 0)
        p = np.pad(a, [(2, 2)] + [(0, 0)] * (a.ndim - 1), mode="reflect")       # numpy 'reflect' == BORDER_REFLECT_101
        n = a.shape[0]
        out = (p[2:2 + n] * k2 + (p[1:1 + n] + p[3:3 + n]) * k1).astype(f) + ((p[0:n] + p[4:4 + n]) * k0).astype(f)
        return np.moveaxis(out.astype(f), 0, axis)
    return one(one(img.astype(f), 1), 0)


def preserve_edges(original: np.ndarray, denoised: np.ndarray, edge_threshold: int = 30) -> np.ndarray:
    """temporal_denoise.py:1636-1667."""
    gray = bgr2gray_u8(original)
    edges = dilate3x3_u8(canny_u8(gray, edge_threshold, edge_threshold * 3))
    mask = gaussian5_f32(edges.astype(np.float32) / np.float32(255.0))[:, :, None]
    res = original.astype(np.float32) * mask + denoised.astype(np.float32) * (np.float32(1) - mask)
    return res.astype(np.uint8)
