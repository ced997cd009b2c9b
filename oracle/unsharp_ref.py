"""ORACLE — test infrastructure only (never imported by the product path).

numpy restatement of what the reference's motion-blur reduction runs in Pillow (third-party, present in the build container:
Pillow 12.2.0), reference call site ``src/framewright/processors/interpolation.py:440-455``::

    img.filter(ImageFilter.UnsharpMask(radius=2, percent=int(100 * strength), threshold=3))
    [strength > 1.5:] .filter(ImageFilter.UnsharpMask(radius=1, percent=int(50 * strength), threshold=2))

Pillow's algorithm (libImaging/BoxBlur.c, UnsharpMask.c):

* ``GaussianBlur(radius)`` = 3 passes of an "extended box blur" per axis (all horizontal passes first, then all vertical
  ones), every pass rounded to 8 bits.  Box radius r = l + a from the Gaussian's variance (Gwosdek et al., SSVM 2011)::

      sigma2 = radius^2 / 3;  L = sqrt(12 sigma2 + 1);  l = floor((L - 1) / 2)
      a = (2l + 1) (l (l + 1) - 3 sigma2) / (6 (sigma2 - (l + 1)^2))          (single precision)

  one pass, edge-replicated samples, 24-bit fixed point::

      ww = uint32(2^24 / (2 r + 1));  fw = (2^24 - (2 l + 1) ww) / 2
      out[x] = (ww * sum_{k=-l..l} in[x + k] + fw * (in[x - l - 1] + in[x + l + 1]) + 2^23) >> 24

* ``UnsharpMask``: ``d = in - blur``; ``out = clip8(in + d * percent / 100)`` (C integer division, truncating toward
  zero) where ``|d| > threshold``, else ``in``.

Pinned: bit-exact against Pillow itself in the build container (``oracle/gen_golden.py`` writes
``tests/golden/unsharp_reference.npz`` from Pillow's output; ``tests/test_oracle_golden.py`` checks this file against it).
"""
from __future__ import annotations

import numpy as np


def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    """libImaging/BoxBlur.c ``_gaussian_blur_radius`` (float variables, the square root in double)."""
    f = np.float32
    sigma2 = f(f(radius) * f(radius) / f(passes))
    L = f(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f(np.floor((float(L) - 1.0) / 2.0))
    a = f(f(f(2) * l + f(1)) * f(f(l * f(l + f(1))) - f(f(3) * sigma2)))
    a = f(a / f(f(6) * f(sigma2 - f(f(l + f(1)) * f(l + f(1))))))
    return f(l + a)


def box_params(float_radius: np.float32):
    """(integer radius, ww, fw) of ``ImagingHorizontalBoxBlur``."""
    r = int(float_radius)
    ww = int(np.float32(np.float32(1 << 24) / np.float32(np.float32(float_radius) * np.float32(2) + np.float32(1))))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def box_blur_pass(img: np.ndarray, axis: int, r: int, ww: int, fw: int) -> np.ndarray:
    """One extended-box pass along ``axis`` of a uint8 array, edge-replicated."""
    n = img.shape[axis]
    x = np.moveaxis(img, axis, 0).astype(np.int64)
    idx = np.arange(n)
    acc = np.zeros_like(x)
    for k in range(-r, r + 1):
        acc += x[np.clip(idx + k, 0, n - 1)]
    far = x[np.clip(idx - r - 1, 0, n - 1)] + x[np.clip(idx + r + 1, 0, n - 1)]
    out = ((acc * ww + far * fw + (1 << 23)) & 0xFFFFFFFF) >> 24
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def gaussian_blur_u8(img: np.ndarray, radius: float, passes: int = 3) -> np.ndarray:
    """``ImagingGaussianBlur``: H x W [x C] uint8."""
    r, ww, fw = box_params(gaussian_box_radius(radius, passes))
    out = img
    for _ in range(passes):
        out = box_blur_pass(out, 1, r, ww, fw)
    for _ in range(passes):
        out = box_blur_pass(out, 0, r, ww, fw)
    return out


def unsharp_mask_u8(img: np.ndarray, radius: float, percent: int, threshold: int) -> np.ndarray:
    """``ImagingUnsharpMask``."""
    blur = gaussian_blur_u8(img, radius).astype(np.int64)
    src = img.astype(np.int64)
    d = src - blur
    prod = d * int(percent)
    corr = np.sign(prod) * (np.abs(prod) // 100)          # C division truncates toward zero
    out = np.where(np.abs(d) > threshold, np.clip(src + corr, 0, 255), src)
    return out.astype(np.uint8)


def motion_blur_reduction(img: np.ndarray, strength: float = 1.0) -> np.ndarray:
    """interpolation.py:440-455."""
    out = unsharp_mask_u8(img, 2, int(100 * strength), 3)
    if strength > 1.5:
        out = unsharp_mask_u8(out, 1, int(50 * strength), 2)
    return out
