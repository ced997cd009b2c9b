"""Restormer on MI355X — the reference's DEFAULT TAP model (`TAPModel.RESTORMER`, tap_denoise.py:110, `_load_restormer`
:299-333).  Host sequencing over the C-ABI building blocks of csrc/restormer_ops.hip, the pointwise MFMA GEMM and the 3x3
MFMA conv kernel; architecture per SURVEY.md §A.4 (the class itself is third-party and absent: parity unpinned, oracle
oracle/restormer_ref.py).

Layout: fp32 NHWC residual stream [pixels][pad32(dim)] (dim 48 is carried with stride 64, pad channels stay zero);
LayerNorm writes operand-typed tensors padded to 32 channels for the GEMMs; q / k / v live in one typed buffer at channel
offsets 0 / dp / 2*dp; the GDFN halves x1 / x2 at 0 / hp (hidden 127 -> 128 etc.), with the 1x1 and depthwise weights
re-laid to match on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Mapping, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FramewrightHipError

RESTORMER_ARGS = dict(dim=48, num_blocks=(4, 6, 6, 8), num_refinement_blocks=4, heads=(1, 2, 4, 8), ffn_expansion_factor=2.66)


def _pad(n: int, m: int = 32) -> int:
    return (n + m - 1) // m * m


def _stages(dim, num_blocks, num_refinement_blocks, heads):
    """(state-dict prefix, blocks, channel count, heads) in forward order."""
    return [("encoder_level1", num_blocks[0], dim, heads[0]), ("encoder_level2", num_blocks[1], 2 * dim, heads[1]),
            ("encoder_level3", num_blocks[2], 4 * dim, heads[2]), ("latent", num_blocks[3], 8 * dim, heads[3]),
            ("decoder_level3", num_blocks[2], 4 * dim, heads[2]), ("decoder_level2", num_blocks[1], 2 * dim, heads[1]),
            ("decoder_level1", num_blocks[0], 2 * dim, heads[0]), ("refinement", num_refinement_blocks, 2 * dim, heads[0])]


def restormer_tensor_shapes(dim=48, num_blocks=(4, 6, 6, 8), num_refinement_blocks=4, heads=(1, 2, 4, 8),
                            ffn_expansion_factor=2.66) -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = [("patch_embed.proj.weight", (dim, 3, 3, 3))]
    for name, n, c, h in _stages(dim, num_blocks, num_refinement_blocks, heads):
        hid = int(c * ffn_expansion_factor)
        for i in range(n):
            p = f"{name}.{i}."
            out += [(p + "norm1.body.weight", (c,)), (p + "norm1.body.bias", (c,)), (p + "attn.temperature", (h, 1, 1)),
                    (p + "attn.qkv.weight", (3 * c, c, 1, 1)), (p + "attn.qkv_dwconv.weight", (3 * c, 1, 3, 3)),
                    (p + "attn.project_out.weight", (c, c, 1, 1)), (p + "norm2.body.weight", (c,)), (p + "norm2.body.bias", (c,)),
                    (p + "ffn.project_in.weight", (2 * hid, c, 1, 1)), (p + "ffn.dwconv.weight", (2 * hid, 1, 3, 3)),
                    (p + "ffn.project_out.weight", (c, hid, 1, 1))]
    out += [("down1_2.body.0.weight", (dim // 2, dim, 3, 3)), ("down2_3.body.0.weight", (dim, 2 * dim, 3, 3)),
            ("down3_4.body.0.weight", (2 * dim, 4 * dim, 3, 3)), ("up4_3.body.0.weight", (16 * dim, 8 * dim, 3, 3)),
            ("reduce_chan_level3.weight", (4 * dim, 8 * dim, 1, 1)), ("up3_2.body.0.weight", (8 * dim, 4 * dim, 3, 3)),
            ("reduce_chan_level2.weight", (2 * dim, 4 * dim, 1, 1)), ("up2_1.body.0.weight", (4 * dim, 2 * dim, 3, 3)),
            ("output.weight", (3, 2 * dim, 3, 3))]
    return out


def synthetic_restormer_state(seed: int = 0, **args):
    """Seeded weights with the published keys/shapes, scaled so that the residual stream stays O(1)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in restormer_tensor_shapes(**args):
        if key.endswith("norm1.body.weight") or key.endswith("norm2.body.weight"):
            sd[key] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif key.endswith(".bias"):
            sd[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif key.endswith("temperature"):
            sd[key] = (1.0 + 0.5 * rng.random(shape)).astype(np.float32)
        elif "dwconv" in key:
            sd[key] = (rng.standard_normal(shape) / 3.0).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.1 if key == "output.weight" else 0.5 if "project_out" in key else 1.0
            sd[key] = (rng.standard_normal(shape) * gain / np.sqrt(fan_in)).astype(np.float32)
    return sd


def _to_numpy(t) -> np.ndarray:
    return t if isinstance(t, np.ndarray) else t.detach().cpu().float().numpy()


class RestormerEngine:
    """Restormer resident on one GPU: thin owner of an ``fw_restormer*`` (csrc/restormer.hip - weight re-layout, workspace arena
    and the ~800 launches of a forward live behind the C-ABI, one mutex per handle).  ``denoise_device`` mirrors NAFNetEngine
    (uint8 BGR in, uint8 BGR out)."""

    def __init__(self, dim: int = 48, num_blocks: Sequence[int] = (4, 6, 6, 8), num_refinement_blocks: int = 4,
                 heads: Sequence[int] = (1, 2, 4, 8), ffn_expansion_factor: float = 2.66, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if dim != 48:
            raise ValueError("RestormerEngine: dim must be 48 (per-head width 48 / 96 is what the attention kernels take)")
        self.args = dict(dim=int(dim), num_blocks=tuple(num_blocks), num_refinement_blocks=int(num_refinement_blocks),
                         heads=tuple(heads), ffn_expansion_factor=float(ffn_expansion_factor))
        for (_, _, c, h) in _stages(dim, num_blocks, num_refinement_blocks, heads):
            if c % h or c // h not in (48, 96):
                raise ValueError("RestormerEngine: channels per head must be 48 or 96")
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
        self.dtype, self.device_id = dtype, int(device_id)
        self._dev = torch.device("cuda", self.device_id)
        nb = (C.c_int * 4)(*self.args["num_blocks"])
        hd = (C.c_int * 4)(*self.args["heads"])
        h = C.c_void_p()
        _lib.check(self._lib.fw_restormer_create(self.device_id, int(dim), nb, int(num_refinement_blocks), hd,
                                                 float(ffn_expansion_factor), _lib.DTYPES[dtype], C.byref(h)))
        self._h = h
        self._state = None

    # ---- weights ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, state: Mapping[str, object]) -> None:
        if "params" in state:
            state = state["params"]  # type: ignore[assignment]
        elif "state_dict" in state:
            state = state["state_dict"]  # type: ignore[assignment]
        kept = {}
        for key, shape in restormer_tensor_shapes(**self.args):
            if key not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {key}")
            a = np.ascontiguousarray(_to_numpy(state[key]), dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{key}: expected shape {shape}, got {a.shape}")
            _lib.check(self._lib.fw_restormer_set_tensor(self._h, key.encode(), C.c_void_p(a.ctypes.data), a.size))
            kept[key] = a
        _lib.check(self._lib.fw_restormer_finalize(self._h))
        self._state = kept

    # ---- forward ----------------------------------------------------------------------------------------------------
    def denoise_device(self, frame, out=None, out_rgb_f32=None):
        """torch.uint8 CUDA tensor H x W x 3 (H, W multiples of 8, as the network's three PixelUnshuffles require) -> same
        shape; asynchronous on torch's current stream of the engine's device."""
        import torch
        if self._state is None:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "RestormerEngine: no weights loaded")
        if frame.dtype != torch.uint8 or not frame.is_cuda or frame.dim() != 3 or frame.shape[2] != 3 or not frame.is_contiguous():
            raise ValueError("denoise_device expects a contiguous uint8 CUDA tensor H x W x 3")
        if frame.device != self._dev:
            raise ValueError(f"tensor is on {frame.device}, engine on {self._dev}")
        H, Wd = int(frame.shape[0]), int(frame.shape[1])
        if H % 8 or Wd % 8:
            raise ValueError(f"Restormer needs frame sizes divisible by 8, got {Wd}x{H}")
        if out is None and out_rgb_f32 is None:
            out = torch.empty_like(frame)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self._lib.fw_restormer_denoise_u8(self._h, p(frame), _lib.FW_DEVICE, H, Wd, p(out), _lib.FW_DEVICE,
                                                     p(out_rgb_f32), st))
        return out if out is not None else out_rgb_f32

    def denoise(self, frame_bgr: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(frame_bgr)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR frame")
        if f.shape[0] % 8 or f.shape[1] % 8:
            raise ValueError(f"Restormer needs frame sizes divisible by 8, got {f.shape[1]}x{f.shape[0]}")
        if self._state is None:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "RestormerEngine: no weights loaded")
        out = np.empty_like(f)
        _lib.check(self._lib.fw_restormer_denoise_u8(self._h, C.c_void_p(f.ctypes.data), _lib.FW_HOST, f.shape[0], f.shape[1],
                                                     C.c_void_p(out.ctypes.data), _lib.FW_HOST, None, None))
        return out

    def clone(self) -> "RestormerEngine":
        """A second handle with the same weights and its own workspace (the tiled TAP path runs several tiles concurrently)."""
        if self._state is None:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "RestormerEngine.clone: no weights loaded")
        e = RestormerEngine(dtype=self.dtype, device_id=self.device_id, **self.args)
        e.load_state_dict(self._state)
        return e

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.fw_restormer_destroy(h)
        self._state = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
