"""Synthetic weights and clips (there is no network access for checkpoints or datasets).

The recipes are the ones fixed in SURVEY.md §8d so that tests, bench.py and the golden fixtures all see the
same numbers:

* weights: ``numpy.random.default_rng(seed)`` (PCG64 — stream stable across numpy versions), PyTorch's
  default Conv2d initialisation restated (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias), every
  ``conv5`` multiplied by 0.1 so the 23-block trunk stays O(1), and ``conv_last`` rescaled so that the network
  output occupies the [0,1] image range like a trained generator's (tolerances in the tests are stated on that
  range).  Keys/shapes are BasicSR's RRDBNet state-dict (reference call site
  src/framewright/processors/pytorch_realesrgan.py:107-127).
* clips: smooth moving content (6 random 2-D sinusoids + 32 random rectangles, translated per frame) plus
  N(0, 4) noise, uint8 BGR.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Tuple

import numpy as np

RRDB_MODELS = {
    # name -> (num_block, netscale)  — pytorch_realesrgan.py:103-129
    "RealESRGAN_x4plus": (23, 4),
    "RealESRGAN_x4plus_anime_6B": (6, 4),
    "RealESRGAN_x2plus": (23, 2),
    # the reference declares these two as RRDBNet as well (SURVEY.md §8f lists that as a reference defect;
    # the declared architecture is what get_upsampler() would build, so it is what is built here)
    "realesr-animevideov3": (6, 4),
    "realesr-general-x4v3": (23, 4),
}


def rrdbnet_conv_shapes(num_block: int, scale: int, num_feat: int = 64, grow: int = 32) -> List[Tuple[str, int, int]]:
    """[(key, cout, cin)] in forward order."""
    in_ch = 3 * (4 if scale == 2 else 1)
    shapes = [("conv_first", num_feat, in_ch)]
    for b in range(num_block):
        for r in (1, 2, 3):
            for c in range(1, 6):
                shapes.append((f"body.{b}.rdb{r}.conv{c}", num_feat if c == 5 else grow, num_feat + grow * (c - 1)))
    shapes += [("conv_body", num_feat, num_feat), ("conv_up1", num_feat, num_feat), ("conv_up2", num_feat, num_feat),
               ("conv_hr", num_feat, num_feat), ("conv_last", 3, num_feat)]
    return shapes


def synthetic_rrdbnet_state(num_block: int, scale: int, seed: int = 1234, num_feat: int = 64, grow: int = 32,
                            image_range: bool = True) -> Dict[str, np.ndarray]:
    """Seeded state-dict (numpy fp32 arrays) with BasicSR key names."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for key, cout, cin in rrdbnet_conv_shapes(num_block, scale, num_feat, grow):
        bound = 1.0 / np.sqrt(cin * 9)
        w = rng.uniform(-bound, bound, size=(cout, cin, 3, 3)).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(cout,)).astype(np.float32)
        if key.endswith(".conv5"):
            w *= np.float32(0.1)
            b *= np.float32(0.1)
        if key == "conv_last" and image_range:
            w *= np.float32(0.25)
            b = b * np.float32(0.25) + np.float32(0.5)
        sd[key + ".weight"] = w
        sd[key + ".bias"] = b
    return sd


def aesrgan_attention_positions(num_block: int, num_attention: int) -> List[int]:
    """RRDB indices followed by an AttentionBlock (aesrgan_face.py:229: range(0, num_block, num_block // num_attention))."""
    return sorted(set(range(0, num_block, num_block // num_attention)))


def synthetic_attention_state(num_block: int, num_attention: int, seed: int = 77, num_feat: int = 64) -> Dict[str, np.ndarray]:
    """Seeded parameters of AESRGAN's AttentionBlocks (aesrgan_face.py:142-168): ``attn.{i}.query|key|value.weight|bias`` and
    ``attn.{i}.gamma`` for the block behind RRDB ``i``; gamma is non-zero (the constructor's zero makes the block inert)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for i in aesrgan_attention_positions(num_block, num_attention):
        bound = 1.0 / np.sqrt(num_feat)
        for name, cout in (("query", num_feat // 8), ("key", num_feat // 8), ("value", num_feat)):
            sd[f"attn.{i}.{name}.weight"] = rng.uniform(-bound, bound, size=(cout, num_feat, 1, 1)).astype(np.float32) * \
                np.float32(4.0 if name != "value" else 1.0)          # sharper logits than the default init: a softmax worth testing
            sd[f"attn.{i}.{name}.bias"] = rng.uniform(-bound, bound, size=(cout,)).astype(np.float32)
        sd[f"attn.{i}.gamma"] = np.array([rng.uniform(0.3, 0.9) * (1 if i % 2 == 0 else -1)], np.float32)
    return sd


def synthetic_clip(num_frames: int, height: int, width: int, seed: int) -> Iterator[np.ndarray]:
    """Yields ``num_frames`` uint8 BGR frames (H x W x 3) of smooth moving content + noise."""
    rng = np.random.default_rng(seed)
    freqs = rng.uniform(0.5, 6.0, size=(6, 2)) * 2 * np.pi
    phases = rng.uniform(0, 2 * np.pi, size=(6, 3))
    amps = rng.uniform(8.0, 28.0, size=(6, 3))
    rects = rng.uniform(0, 1, size=(32, 4))
    rect_col = rng.uniform(-60, 60, size=(32, 3))
    vx, vy = rng.uniform(-3, 3, size=2)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    for t in range(num_frames):
        u = (xx + vx * t) / max(width, 1)
        v = (yy + vy * t) / max(height, 1)
        img = np.full((height, width, 3), 120.0, dtype=np.float32)
        for k in range(6):
            arg = freqs[k, 0] * u + freqs[k, 1] * v
            for c in range(3):
                img[:, :, c] += amps[k, c] * np.sin(arg + phases[k, c])
        uf = u - np.floor(u)
        vf = v - np.floor(v)
        for k in range(32):
            x0, y0 = rects[k, 0], rects[k, 1]
            w = 0.02 + 0.15 * rects[k, 2]
            h = 0.02 + 0.15 * rects[k, 3]
            m = (uf >= x0) & (uf < x0 + w) & (vf >= y0) & (vf < y0 + h)
            img[m] += rect_col[k]
        img += rng.normal(0.0, 4.0, size=img.shape).astype(np.float32)
        yield np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synthetic_frames(num_frames: int, height: int, width: int, seed: int) -> np.ndarray:
    return np.stack(list(synthetic_clip(num_frames, height, width, seed)))


def nafnet_tensor_shapes(width: int, middle_blk_num: int, enc_blk_nums, dec_blk_nums) -> List[Tuple[str, Tuple[int, ...]]]:
    """[(key, shape)] of the NAFNet state-dict (module tree of SURVEY.md §A.3; reference constructor arguments at
    src/framewright/processors/tap_denoise.py:340-346)."""
    out: List[Tuple[str, Tuple[int, ...]]] = [("intro.weight", (width, 3, 3, 3)), ("intro.bias", (width,)),
                                              ("ending.weight", (3, width, 3, 3)), ("ending.bias", (3,))]

    def block(prefix: str, c: int):
        return [(prefix + "conv1.weight", (2 * c, c, 1, 1)), (prefix + "conv1.bias", (2 * c,)),
                (prefix + "conv2.weight", (2 * c, 1, 3, 3)), (prefix + "conv2.bias", (2 * c,)),
                (prefix + "conv3.weight", (c, c, 1, 1)), (prefix + "conv3.bias", (c,)),
                (prefix + "sca.1.weight", (c, c, 1, 1)), (prefix + "sca.1.bias", (c,)),
                (prefix + "conv4.weight", (2 * c, c, 1, 1)), (prefix + "conv4.bias", (2 * c,)),
                (prefix + "conv5.weight", (c, c, 1, 1)), (prefix + "conv5.bias", (c,)),
                (prefix + "norm1.weight", (c,)), (prefix + "norm1.bias", (c,)),
                (prefix + "norm2.weight", (c,)), (prefix + "norm2.bias", (c,)),
                (prefix + "beta", (1, c, 1, 1)), (prefix + "gamma", (1, c, 1, 1))]

    c = width
    for l, nb in enumerate(enc_blk_nums):
        for j in range(nb):
            out += block(f"encoders.{l}.{j}.", c)
        out += [(f"downs.{l}.weight", (2 * c, c, 2, 2)), (f"downs.{l}.bias", (2 * c,))]
        c *= 2
    for j in range(middle_blk_num):
        out += block(f"middle_blks.{j}.", c)
    for i, nb in enumerate(dec_blk_nums):
        out += [(f"ups.{i}.0.weight", (2 * c, c, 1, 1))]
        c //= 2
        for j in range(nb):
            out += block(f"decoders.{i}.{j}.", c)
    return out


def synthetic_nafnet_state(width: int = 64, middle_blk_num: int = 12, enc_blk_nums=(2, 2, 4, 8),
                           dec_blk_nums=(2, 2, 2, 2), seed: int = 4321) -> Dict[str, np.ndarray]:
    """Seeded NAFNet weights: PyTorch default conv init restated (U(+-1/sqrt(fan_in))), LayerNorm weight 1 +- 0.1 /
    bias +-0.1, beta/gamma U(-0.3, 0.3) (upstream initialises them to 0, which would make every block the identity),
    ending scaled by 0.1 so that the output stays an image-like perturbation of the input."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for key, shape in nafnet_tensor_shapes(width, middle_blk_num, enc_blk_nums, dec_blk_nums):
        if key.endswith("beta") or key.endswith("gamma"):
            v = rng.uniform(-0.3, 0.3, size=shape)
        elif ".norm" in key:
            v = (1.0 + rng.uniform(-0.1, 0.1, size=shape)) if key.endswith("weight") else rng.uniform(-0.1, 0.1, size=shape)
        else:
            if key.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
            else:
                wshape = dict(nafnet_tensor_shapes(width, middle_blk_num, enc_blk_nums, dec_blk_nums))[key[:-4] + "weight"]
                fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-bound, bound, size=shape)
            if key.startswith("ending."):
                v = v * 0.1
        sd[key] = v.astype(np.float32)
    return sd


IFNET_CHANNELS = (192, 128, 96, 64)   # IFNet_HDv3 v4.6
IFNET_SCALES = (8, 4, 2, 1)


def ifnet_tensor_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []
    for i, c in enumerate(IFNET_CHANNELS):
        cin = 7 if i == 0 else 12
        p = f"block{i}."
        out += [(p + "conv0.0.0.weight", (c // 2, cin, 3, 3)), (p + "conv0.0.0.bias", (c // 2,)),
                (p + "conv0.1.0.weight", (c, c // 2, 3, 3)), (p + "conv0.1.0.bias", (c,))]
        for j in range(8):
            out += [(f"{p}convblock.{j}.conv.weight", (c, c, 3, 3)), (f"{p}convblock.{j}.conv.bias", (c,)),
                    (f"{p}convblock.{j}.beta", (1, c, 1, 1))]
        out += [(p + "lastconv.0.weight", (c, 24, 4, 4)), (p + "lastconv.0.bias", (24,))]
    return out


def synthetic_ifnet_state(seed: int = 2468, flow_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded IFNet weights: PyTorch default init restated; ResConv beta U(0.2, 0.6); lastconv scaled so that the flow
    increments are a few pixels at most (a trained net's are) and the masks moderate; ``flow_gain`` scales them
    (tests use a larger gain to exercise multi-pixel warps)."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    shapes = dict(ifnet_tensor_shapes())
    for key, shape in ifnet_tensor_shapes():
        if key.endswith("beta"):
            v = rng.uniform(0.2, 0.6, size=shape)
        else:
            wshape = shape if key.endswith("weight") else shapes[key[:-4] + "weight"]
            if "lastconv" in key:
                fan_in = wshape[0] * 4  # transposed conv: 2x2 taps of c inputs contribute to an output
            else:
                fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            v = rng.uniform(-bound, bound, size=shape)
            if "lastconv" in key:
                v = v * 0.5 * flow_gain
        sd[key] = v.astype(np.float32)
    return sd
