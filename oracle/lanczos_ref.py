"""ORACLE — test infrastructure only (never imported by the product path).

``cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LANCZOS4)`` for 8-bit images: the last step of
``realesrgan.RealESRGANer.enhance`` when ``outscale != netscale`` (pip ``realesrgan``, unpinned and absent; reference call
site ``src/framewright/processors/pytorch_realesrgan.py:223``).  OpenCV itself is absent from this image, so this restates
its published algorithm (imgproc/resize.cpp: ``interpolateLanczos4``, ``resizeGeneric_`` with
``HResizeLanczos4<uchar,int,short>`` / ``VResizeLanczos4<uchar,int,short,FixedPtCast<int,uchar,22>>``) as the two separate
passes OpenCV runs — **parity unpinned** (no cv2 to check against).  Known-answer properties are tested instead: identity
at equal size, constants stay constant, exact 2:1 phase weights.

``resize_lanczos4_u16`` is the same resize on 16-bit images (the tail of ``RealESRGANer.enhance`` on a 16-bit frame): OpenCV's
path for ushort is float (``HResizeLanczos4<ushort,float,float>`` / ``VResizeLanczos4<ushort,float,float,Cast<float,ushort>>``):
the normalised float weights, eight float32 products added left to right per pass, ``saturate_cast<ushort>(cvRound(sum))``.
Also unpinned (and OpenCV's SIMD build may associate the vertical sum differently).
"""
from __future__ import annotations

import math

import numpy as np

_S45 = 0.70710678118654752440084436210485
_CS = [(1, 0), (-_S45, -_S45), (0, 1), (_S45, -_S45), (-1, 0), (_S45, _S45), (0, -1), (-_S45, _S45)]


def interpolate_lanczos4(x: np.float32) -> np.ndarray:
    c = np.zeros(8, np.float32)
    if x < np.finfo(np.float32).eps:
        c[3] = 1
        return c
    y0 = -(float(x) + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    total = np.float32(0)
    for i in range(8):
        y = -(float(x) + 3 - i) * math.pi * 0.25
        c[i] = np.float32((_CS[i][0] * s0 + _CS[i][1] * c0) / (y * y))
        total = np.float32(total + c[i])
    inv = np.float32(1) / total
    return (c * inv).astype(np.float32)


def tables(ssize: int, dsize: int):
    scale = 1.0 / (dsize / ssize)
    ofs = np.zeros(dsize, np.int64)
    coef = np.zeros((dsize, 8), np.int64)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s0 = math.floor(float(f))
        f = np.float32(f - np.float32(s0))
        ofs[d] = s0
        c = interpolate_lanczos4(f) * np.float32(2048)
        coef[d] = np.clip(np.rint(c), -32768, 32767).astype(np.int64)   # saturate_cast<short>: round half to even
    return ofs, coef


def resize_lanczos4_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    squeeze = img.ndim == 2
    src = img[:, :, None] if squeeze else img
    hs, ws, _ = src.shape
    xofs, ia = tables(ws, dw)
    yofs, ib = tables(hs, dh)
    s = src.astype(np.int64)
    # horizontal pass -> exact ints
    hbuf = np.zeros((hs, dw, src.shape[2]), np.int64)
    for j in range(8):
        xs = np.clip(xofs + j - 3, 0, ws - 1)
        hbuf += s[:, xs, :] * ia[:, j][None, :, None]
    out = np.zeros((dh, dw, src.shape[2]), np.int64)
    for k in range(8):
        ys = np.clip(yofs + k - 3, 0, hs - 1)
        out += hbuf[ys] * ib[:, k][:, None, None]
    out = np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def float_tables(ssize: int, dsize: int):
    scale = 1.0 / (dsize / ssize)
    ofs = np.zeros(dsize, np.int64)
    coef = np.zeros((dsize, 8), np.float32)
    for d in range(dsize):
        f = np.float32((d + 0.5) * scale - 0.5)
        s0 = math.floor(float(f))
        ofs[d] = s0
        coef[d] = interpolate_lanczos4(np.float32(f - np.float32(s0)))
    return ofs, coef


def resize_lanczos4_u16(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    squeeze = img.ndim == 2
    src = (img[:, :, None] if squeeze else img).astype(np.float32)
    hs, ws, _ = src.shape
    xofs, fa = float_tables(ws, dw)
    yofs, fb = float_tables(hs, dh)
    hbuf = None
    for j in range(8):                                   # float32 throughout, left to right, each product rounded before the add
        xs = np.clip(xofs + j - 3, 0, ws - 1)
        term = (src[:, xs, :] * fa[:, j][None, :, None]).astype(np.float32)
        hbuf = term if hbuf is None else (hbuf + term).astype(np.float32)
    out = None
    for k in range(8):
        ys = np.clip(yofs + k - 3, 0, hs - 1)
        term = (hbuf[ys] * fb[:, k][:, None, None]).astype(np.float32)
        out = term if out is None else (out + term).astype(np.float32)
    out = np.clip(np.rint(out), 0, 65535).astype(np.uint16)  # cvRound: half to even
    return out[:, :, 0] if squeeze else out
