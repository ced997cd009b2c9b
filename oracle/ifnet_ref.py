"""ORACLE — test infrastructure only (never imported by the product path).

fp32 CPU restatement of IFNet as shipped for ``rife-v4.6`` (Practical-RIFE ``IFNet_HDv3`` v4.6) — the network inside
the external binary ``rife-ncnn-vulkan`` (release 20221029) that the reference shells out to at
``src/framewright/processors/interpolation.py:628-650`` (model name ``rife-v4.6``, ``:106-124``).  Neither the binary
nor its model is in /root/reference or in this image; the published architecture is restated from SURVEY.md §A.5 and
**parity vs upstream is unpinned** (no reference test or fixture pins a numeric result on this path).

State-dict keys follow the upstream module tree: ``block{i}.conv0.{0,1}.0.{weight,bias}``,
``block{i}.convblock.{j}.conv.{weight,bias}``, ``block{i}.convblock.{j}.beta``, ``block{i}.lastconv.0.{weight,bias}``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]
CHANNELS = (192, 128, 96, 64)
SCALES = (8, 4, 2, 1)


def warp(img: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """grid_sample(bilinear, border, align_corners=True) with the flow in pixels."""
    _, _, h, w = img.shape
    gx = torch.linspace(-1.0, 1.0, w).view(1, 1, 1, w).expand(1, 1, h, w)
    gy = torch.linspace(-1.0, 1.0, h).view(1, 1, h, 1).expand(1, 1, h, w)
    grid = torch.cat([gx, gy], 1)
    f = torch.cat([flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    g = (grid + f).permute(0, 2, 3, 1)
    return F.grid_sample(img, g, mode="bilinear", padding_mode="border", align_corners=True)


def ifblock(sd: StateDict, p: str, x: torch.Tensor, flow: Optional[torch.Tensor], scale: int) -> Tuple[torch.Tensor, torch.Tensor]:
    x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False)
    if flow is not None:
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False) * (1.0 / scale)
        x = torch.cat([x, flow], 1)
    lr = lambda t: F.leaky_relu(t, 0.2)
    feat = lr(F.conv2d(x, sd[p + "conv0.0.0.weight"], sd[p + "conv0.0.0.bias"], stride=2, padding=1))
    feat = lr(F.conv2d(feat, sd[p + "conv0.1.0.weight"], sd[p + "conv0.1.0.bias"], stride=2, padding=1))
    for j in range(8):
        q = f"{p}convblock.{j}."
        feat = lr(F.conv2d(feat, sd[q + "conv.weight"], sd[q + "conv.bias"], padding=1) * sd[q + "beta"] + feat)
    tmp = F.pixel_shuffle(F.conv_transpose2d(feat, sd[p + "lastconv.0.weight"], sd[p + "lastconv.0.bias"], stride=2, padding=1), 2)
    tmp = F.interpolate(tmp, scale_factor=float(scale), mode="bilinear", align_corners=False)
    return tmp[:, :4] * scale, tmp[:, 4:5]


def ifnet_forward(sd: StateDict, img0: torch.Tensor, img1: torch.Tensor, timestep: float = 0.5) -> torch.Tensor:
    """img0/img1: 1 x 3 x H x W fp32 RGB in [0,1]; returns the interpolated frame (un-clamped)."""
    _, _, h, w = img0.shape
    ph, pw = ((h - 1) // 32 + 1) * 32, ((w - 1) // 32 + 1) * 32
    i0 = F.pad(img0, (0, pw - w, 0, ph - h))
    i1 = F.pad(img1, (0, pw - w, 0, ph - h))
    t = torch.full((1, 1, ph, pw), float(timestep))
    flow = mask = None
    w0, w1 = i0, i1
    for i, s in enumerate(SCALES):
        if flow is None:
            flow, mask = ifblock(sd, f"block{i}.", torch.cat([i0, i1, t], 1), None, s)
        else:
            fd, md = ifblock(sd, f"block{i}.", torch.cat([w0, w1, t, mask], 1), flow, s)
            flow = flow + fd
            mask = mask + md
        w0 = warp(i0, flow[:, :2])
        w1 = warp(i1, flow[:, 2:4])
    m = torch.sigmoid(mask)
    return (w0 * m + w1 * (1 - m))[:, :, :h, :w]
