"""TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing on the product path).

fp32 CPU restatement of SRVGGNetCompact, the network behind the Real-ESRGAN checkpoints `realesr-animevideov3`
(num_conv=16) and `realesr-general-x4v3` (num_conv=32).  The reference lists both checkpoints in its model table
(src/framewright/processors/pytorch_realesrgan.py:119-128, and mis-declares them as RRDBNet — SURVEY.md §8f item 4); the
network itself lives in the third-party pip package `realesrgan` (`realesrgan.archs.srvgg_arch`, unpinned, absent from
/root/reference), so this follows the published architecture:  PARITY UNPINNED at the third-party boundary.

    body = [conv3x3(3 -> F), PReLU(F)] + num_conv x [conv3x3(F -> F), PReLU(F)] + [conv3x3(F -> 3*s*s)]
    out  = pixel_shuffle(body(x), s) + nearest_upsample(x, s)
State-dict keys: body.{2k}.weight / .bias for the convs, body.{2k+1}.weight (shape [F]) for the PReLU slopes.
"""
import torch
import torch.nn.functional as F


def srvgg_layout(num_conv: int):
    """(index of the i-th conv in `body`, index of its PReLU or None) for the num_conv + 2 convs."""
    out = []
    for i in range(num_conv + 2):
        out.append((2 * i, 2 * i + 1 if i < num_conv + 1 else None))
    return out


def srvgg_forward(sd, x, num_conv: int, scale: int):
    """x: [1, 3, H, W] RGB in [0, 1] -> [1, 3, s*H, s*W]."""
    out = x
    for ci, pi in srvgg_layout(num_conv):
        out = F.conv2d(out, sd[f"body.{ci}.weight"], sd[f"body.{ci}.bias"], 1, 1)
        if pi is not None:
            out = F.prelu(out, sd[f"body.{pi}.weight"])
    out = F.pixel_shuffle(out, scale)
    return out + F.interpolate(x, scale_factor=scale, mode="nearest")
