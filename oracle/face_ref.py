"""ORACLE — test infrastructure only (never imported by the product path).

The host arithmetic of ``AESRGANFaceRestorer`` around its network (reference ``src/framewright/processors/aesrgan_face.py``):
``_extract_face`` (:493-513, pure index arithmetic), the tensor round trip of ``_enhance_face`` (:515-541: BGR -> RGB, / 255,
model, ``np.clip(y * 255, 0, 255).astype(np.uint8)`` - a TRUNCATING cast -, RGB -> BGR) and ``_paste_face_back`` (:543-584:
``cv2.resize(enhanced, (w, h))``, a feathered float32 mask, ``orig * (1 - mask * s) + enh * mask * s`` in float32, truncating
cast).

``cv2.resize`` with its default interpolation (INTER_LINEAR) on 8-bit images is third-party code that is absent from this image
(no cv2): restated here from OpenCV's published algorithm (imgproc/resize.cpp) - **parity unpinned**:
  * equal size: a copy;
  * an exact 2 x 2 decimation (what the default ``upscale_factor = 2`` produces: the enhanced crop is twice the region) takes
    OpenCV's "area fast" path (``if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2)
    interpolation = INTER_AREA``): ``(a + b + c + d + 2) >> 2`` per 2 x 2 block;
  * otherwise the fixed-point bilinear: ``fx = float((dx + 0.5) * scale - 0.5)``, ``sx = floor(fx)``, clamped at both borders with
    ``fx = 0``; coefficients ``saturate_cast<short>(c * 2048)``; the horizontal pass keeps ints (``S[sx] * a0 + S[sx + 1] * a1``),
    the vertical pass is ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`` (``VResizeLinear<uchar, int, short>``).
Known answers are tested instead (identity, constants, the 2 x 2 mean, the 4 : 1 phase = mean of the two middle pixels).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def _linear_tables(ssize: int, dsize: int):
    scale = 1.0 / (dsize / ssize)
    ofs = np.zeros(dsize, np.int64)
    coef = np.zeros((dsize, 2), np.int64)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s0 = math.floor(float(f))
        f = np.float32(f - np.float32(s0))
        if s0 < 0:
            f, s0 = np.float32(0), 0
        if s0 >= ssize - 1:
            f, s0 = np.float32(0), ssize - 1
        ofs[d] = s0
        c0, c1 = np.float32(1) - f, f
        coef[d, 0] = int(np.clip(np.rint(c0 * np.float32(2048)), -32768, 32767))   # saturate_cast<short>: round half to even
        coef[d, 1] = int(np.clip(np.rint(c1 * np.float32(2048)), -32768, 32767))
    return ofs, coef


def resize_linear_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """``cv2.resize(img, (dw, dh))`` for uint8 H x W [x C]."""
    squeeze = img.ndim == 2
    src = (img[:, :, None] if squeeze else img).astype(np.int64)
    hs, ws, _ = src.shape
    if (dw, dh) == (ws, hs):
        out = src
    elif ws == 2 * dw and hs == 2 * dh:
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
    else:
        xofs, ia = _linear_tables(ws, dw)
        yofs, ib = _linear_tables(hs, dh)
        x1 = np.minimum(xofs + 1, ws - 1)
        hor = src[:, xofs, :] * ia[None, :, 0, None] + src[:, x1, :] * ia[None, :, 1, None]      # [hs][dw][c] ints
        y1 = np.minimum(yofs + 1, hs - 1)
        s0, s1 = hor[yofs], hor[y1]
        b0, b1 = ib[:, 0][:, None, None], ib[:, 1][:, None, None]
        out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def extract_face(frame: np.ndarray, box: Tuple[int, int, int, int], padding: float = 0.3):
    """aesrgan_face.py:493-513: the box grown by ``int(size * padding)`` on every side, clipped to the frame."""
    h, w = frame.shape[:2]
    bx1, by1, bx2, by2 = box
    pad_w, pad_h = int((bx2 - bx1) * padding), int((by2 - by1) * padding)
    x1, y1 = max(0, bx1 - pad_w), max(0, by1 - pad_h)
    x2, y2 = min(w, bx2 + pad_w), min(h, by2 + pad_h)
    return frame[y1:y2, x1:x2].copy(), (x1, y1, x2, y2)


def feather_mask(h: int, w: int) -> np.ndarray:
    """aesrgan_face.py:556-567, statement for statement."""
    mask = np.ones((h, w), dtype=np.float32)
    feather = min(w, h) // 8
    if feather > 0:
        for i in range(feather):
            alpha = i / feather
            mask[i, :] *= alpha
            mask[-i - 1, :] *= alpha
            mask[:, i] *= alpha
            mask[:, -i - 1] *= alpha
    return mask


def paste_face_back(frame: np.ndarray, enhanced_face: np.ndarray, region: Tuple[int, int, int, int], strength: float) -> np.ndarray:
    """aesrgan_face.py:543-584."""
    x1, y1, x2, y2 = region
    th, tw = y2 - y1, x2 - x1
    enhanced_resized = resize_linear_u8(enhanced_face, tw, th)
    mask = feather_mask(th, tw)[:, :, np.newaxis]
    original_region = frame[y1:y2, x1:x2].astype(np.float32)
    enhanced_float = enhanced_resized.astype(np.float32)
    blended = original_region * (1 - mask * strength) + enhanced_float * mask * strength
    result = frame.copy()
    result[y1:y2, x1:x2] = blended.astype(np.uint8)
    return result


def postprocess_truncating(rgb01: np.ndarray) -> np.ndarray:
    """aesrgan_face.py:537-539: float RGB in (about) [0, 1], H x W x 3 -> uint8 BGR with a truncating cast."""
    return np.clip(rgb01.astype(np.float32) * np.float32(255.0), 0, 255).astype(np.uint8)[:, :, ::-1]
