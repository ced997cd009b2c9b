"""ORACLE — test infrastructure only (never imported by the product path).

fp32 CPU restatement of ``basicsr.archs.nafnet_arch.NAFNet`` as the reference constructs it at
``src/framewright/processors/tap_denoise.py:338-346`` (``img_channel=3, width=64, middle_blk_num=12,
enc_blk_nums=[2,2,4,8], dec_blk_nums=[2,2,2,2]``) and calls it at ``:433`` / ``:458``.  The class lives in the
megvii-research/NAFNet fork of BasicSR, an UNPINNED dependency that is absent from /root/reference and from this image
(mainline basicsr does not ship it at all — SURVEY.md §8c); the published architecture is restated from SURVEY.md §A.3.
No reference test or fixture pins its numerics, so **parity vs upstream is unpinned**; the HIP kernels are checked
against THIS restatement.

State-dict keys follow the upstream module tree: intro, ending, encoders.{l}.{j}, middle_blks.{j}, decoders.{i}.{j},
downs.{l}, ups.{i}.0; per block conv1..conv5, sca.1, norm1, norm2, beta, gamma.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]


def layernorm2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    y = (x - mu) / (var + eps).sqrt()
    return w.view(1, -1, 1, 1) * y + b.view(1, -1, 1, 1)


def simple_gate(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x.chunk(2, dim=1)
    return x1 * x2


def nafblock(sd: StateDict, p: str, inp: torch.Tensor) -> torch.Tensor:
    c = inp.shape[1]
    x = layernorm2d(inp, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    x = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    x = F.conv2d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1, groups=2 * c)
    x = simple_gate(x)
    x = x * F.conv2d(F.adaptive_avg_pool2d(x, 1), sd[p + "sca.1.weight"], sd[p + "sca.1.bias"])
    x = F.conv2d(x, sd[p + "conv3.weight"], sd[p + "conv3.bias"])
    y = inp + x * sd[p + "beta"]
    x = F.conv2d(layernorm2d(y, sd[p + "norm2.weight"], sd[p + "norm2.bias"]), sd[p + "conv4.weight"], sd[p + "conv4.bias"])
    x = simple_gate(x)
    x = F.conv2d(x, sd[p + "conv5.weight"], sd[p + "conv5.bias"])
    return y + x * sd[p + "gamma"]


def nafnet_forward(sd: StateDict, inp: torch.Tensor, middle_blk_num: int, enc_blk_nums: Sequence[int],
                   dec_blk_nums: Sequence[int]) -> torch.Tensor:
    """inp: N x 3 x H x W fp32 RGB in [0,1]; returns N x 3 x H x W (un-clamped)."""
    _, _, H, W = inp.shape
    mult = 2 ** len(enc_blk_nums)
    ph, pw = (mult - H % mult) % mult, (mult - W % mult) % mult
    inp_p = F.pad(inp, (0, pw, 0, ph))                                   # check_image_size: zero pad
    x = F.conv2d(inp_p, sd["intro.weight"], sd["intro.bias"], padding=1)
    encs = []
    for l, nb in enumerate(enc_blk_nums):
        for j in range(nb):
            x = nafblock(sd, f"encoders.{l}.{j}.", x)
        encs.append(x)
        x = F.conv2d(x, sd[f"downs.{l}.weight"], sd[f"downs.{l}.bias"], stride=2)
    for j in range(middle_blk_num):
        x = nafblock(sd, f"middle_blks.{j}.", x)
    for i, nb in enumerate(dec_blk_nums):
        x = F.pixel_shuffle(F.conv2d(x, sd[f"ups.{i}.0.weight"]), 2)
        x = x + encs[len(encs) - 1 - i]
        for j in range(nb):
            x = nafblock(sd, f"decoders.{i}.{j}.", x)
    x = F.conv2d(x, sd["ending.weight"], sd["ending.bias"], padding=1)
    x = x + inp_p
    return x[:, :, :H, :W]
