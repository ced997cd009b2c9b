"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of ``realesrgan.RealESRGANer.enhance`` — the third-party call the reference makes at
``src/framewright/processors/pytorch_realesrgan.py:223,227``, ``processors/enhancement/super_resolution.py:524``
and ``cli.py:769`` with the constructor arguments of ``pytorch_realesrgan.py:160-170``
(``tile``, ``tile_pad=10``, ``pre_pad=0``, ``half``).  The pip package ``realesrgan`` is an UNPINNED dependency
(``pytorch_realesrgan.py:79``: "pip install realesrgan basicsr") and is absent from /root/reference and from this
image, so this file restates its published algorithm as recorded in SURVEY.md §A.2; parity with the package itself
is unpinned.  The network is oracle/rrdbnet_ref.py.

Steps:
  enhance      : float32, /255 (or /65535 if max > 256); gray -> RGB, BGRA -> BGR + alpha, BGR -> RGB
  pre_process  : HWC -> NCHW; reflect pre_pad (right/bottom); mod-pad (scale 2 -> multiple of 2, scale 1 -> 4) reflect
  process      : whole frame, or tile_process: tile x tile input tiles extended by tile_pad (clamped to the image),
                 un-padded centre copied into the output (hard seams)
  post_process : crop mod-pad, crop pre-pad
  back         : clamp(0,1), CHW -> HWC, RGB -> BGR, (x*255).round() -> uint8   (x*65535 -> uint16)
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .rrdbnet_ref import StateDict, rrdbnet_forward

Model = Callable[[torch.Tensor], torch.Tensor]


def _pre_process(img_rgb: np.ndarray, scale: int, pre_pad: int) -> Tuple[torch.Tensor, int, int]:
    x = torch.from_numpy(np.ascontiguousarray(np.transpose(img_rgb, (2, 0, 1)))).float().unsqueeze(0)
    if pre_pad != 0:
        x = F.pad(x, (0, pre_pad, 0, pre_pad), "reflect")
    mod_scale = 2 if scale == 2 else (4 if scale == 1 else None)
    mod_pad_h = mod_pad_w = 0
    if mod_scale is not None:
        _, _, h, w = x.shape
        if h % mod_scale != 0:
            mod_pad_h = mod_scale - h % mod_scale
        if w % mod_scale != 0:
            mod_pad_w = mod_scale - w % mod_scale
        x = F.pad(x, (0, mod_pad_w, 0, mod_pad_h), "reflect")
    return x, mod_pad_h, mod_pad_w


def _tile_process(model: Model, x: torch.Tensor, scale: int, tile: int, tile_pad: int) -> torch.Tensor:
    b, c, h, w = x.shape
    out = x.new_zeros((b, c, h * scale, w * scale))
    tiles_x = math.ceil(w / tile)
    tiles_y = math.ceil(h / tile)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            ofs_x, ofs_y = tx * tile, ty * tile
            sx0, sx1 = ofs_x, min(ofs_x + tile, w)
            sy0, sy1 = ofs_y, min(ofs_y + tile, h)
            px0, px1 = max(sx0 - tile_pad, 0), min(sx1 + tile_pad, w)
            py0, py1 = max(sy0 - tile_pad, 0), min(sy1 + tile_pad, h)
            o = model(x[:, :, py0:py1, px0:px1])
            ox0, oy0 = (sx0 - px0) * scale, (sy0 - py0) * scale
            out[:, :, sy0 * scale:sy1 * scale, sx0 * scale:sx1 * scale] = \
                o[:, :, oy0:oy0 + (sy1 - sy0) * scale, ox0:ox0 + (sx1 - sx0) * scale]
    return out


def run_network(model: Model, img_rgb: np.ndarray, scale: int, tile: int = 0, tile_pad: int = 10,
                pre_pad: int = 0) -> np.ndarray:
    """img_rgb: H x W x 3 float32 in [0,1].  Returns sH x sW x 3 float32, clamped to [0,1] (RGB)."""
    x, mph, mpw = _pre_process(img_rgb, scale, pre_pad)
    with torch.no_grad():
        y = _tile_process(model, x, scale, tile, tile_pad) if tile > 0 else model(x)
    _, _, h, w = y.shape
    y = y[:, :, 0:h - mph * scale, 0:w - mpw * scale]
    if pre_pad != 0:
        _, _, h, w = y.shape
        y = y[:, :, 0:h - pre_pad * scale, 0:w - pre_pad * scale]
    return np.transpose(y.squeeze(0).float().clamp_(0, 1).numpy(), (1, 2, 0))


def enhance(model: Model, img: np.ndarray, scale: int, outscale: Optional[float] = None, tile: int = 0,
            tile_pad: int = 10, pre_pad: int = 0) -> Tuple[np.ndarray, str]:
    """``RealESRGANer.enhance(img, outscale)`` for a cv2-style array (BGR / BGRA / gray, uint8 or uint16)."""
    h_input, w_input = img.shape[:2]
    img = img.astype(np.float32)
    if np.max(img) > 256:
        max_range = 65535
    else:
        max_range = 255
    img = img / max_range
    alpha = None
    if img.ndim == 2:
        img_mode = "L"
        rgb = np.repeat(img[:, :, None], 3, axis=2)
    elif img.shape[2] == 4:
        img_mode = "RGBA"
        alpha = np.repeat(img[:, :, 3:4], 3, axis=2)
        rgb = img[:, :, 2::-1]
    else:
        img_mode = "RGB"
        rgb = img[:, :, ::-1]
    out = run_network(model, rgb, scale, tile, tile_pad, pre_pad)
    if img_mode == "L":
        # cv2.COLOR_BGR2GRAY of the BGR-converted output
        bgr = out[:, :, ::-1]
        out_img = (0.114 * bgr[:, :, 0] + 0.587 * bgr[:, :, 1] + 0.299 * bgr[:, :, 2]).astype(np.float32)
    else:
        out_img = out[:, :, ::-1]
    if img_mode == "RGBA":
        a = run_network(model, alpha, scale, tile, tile_pad, pre_pad)
        a_gray = (0.114 * a[:, :, 2] + 0.587 * a[:, :, 1] + 0.299 * a[:, :, 0]).astype(np.float32)
        out_img = np.concatenate([out_img, a_gray[:, :, None]], axis=2)
    if max_range == 65535:
        res = (out_img * 65535.0).round().astype(np.uint16)
    else:
        res = (out_img * 255.0).round().astype(np.uint8)
    if outscale is not None and float(outscale) != float(scale):
        # cv2.resize(output, (int(w_input * outscale), int(h_input * outscale)), interpolation=cv2.INTER_LANCZOS4)
        from .lanczos_ref import resize_lanczos4_u16, resize_lanczos4_u8
        res = (resize_lanczos4_u8 if res.dtype == np.uint8 else resize_lanczos4_u16)(res, int(w_input * outscale), int(h_input * outscale))
    return res, img_mode


def make_model(sd: StateDict, num_block: int, scale: int) -> Model:
    return lambda x: rrdbnet_forward(sd, x, num_block, scale)
