"""The reference's in-tree ``AESRGAN`` — RRDB trunk with full-image self-attention blocks — on the MI355X kernels.

Reference: ``src/framewright/processors/aesrgan_face.py`` — ``AttentionBlock`` (:142-168), ``ResidualDenseBlock`` / ``RRDB``
(:171-204), ``AESRGAN`` (:205-269: no pixel-unshuffle front end, an AttentionBlock behind every ``num_block // num_attention``-th
RRDB, ``conv_up2`` only for scale >= 4), the network ``AESRGANFaceRestorer`` (:383+) runs on face crops.  SURVEY.md section
8f/4 ("other RRDB consumers").  Face detection and paste-back of that class are cv2 host code and stay out of scope.

One engine behind the C-ABI (``fw_aesrgan_*``, csrc/aesrgan.hip - round 1 sequenced the launches from here): every 3x3
convolution on the MFMA conv kernel (chunk-planar typed activations, the residual streams in fp32, `x5 * 0.2 + x` and the RRDB's
second residual in the conv5 epilogue), the 1x1 query / key / value projections on the pointwise GEMM, softmax(q^T k) over all
pixels as a row kernel, and the product with v as a GEMM over the pixel axis with the gamma residual epilogue.
Parity: oracle/rrdbnet_ref.py ``aesrgan_forward``, which is pinned on vectors the reference's own module produced
(tests/golden/aesrgan_attention.npz).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Mapping

import numpy as np

from . import _lib
from ._lib import FramewrightHipError
from .synth import aesrgan_attention_positions


def _np(t) -> np.ndarray:
    return t if isinstance(t, np.ndarray) else t.detach().cpu().float().numpy()


class AESRGANEngine:
    """``AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block, scale, num_attention)`` resident on one GPU: thin owner of an
    ``fw_aesrgan*``; the weights, the workspace arena and the launches of a forward live behind the C-ABI."""

    def __init__(self, num_block: int = 23, scale: int = 2, num_attention: int = 4, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if scale not in (2, 4) or num_block < 1 or num_attention < 1 or num_attention > num_block:
            raise ValueError("AESRGANEngine: scale 2 or 4, 1 <= num_attention <= num_block")
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
        self.num_block, self.scale, self.num_attention = int(num_block), int(scale), int(num_attention)
        self.attn_after = aesrgan_attention_positions(self.num_block, self.num_attention)
        self.dtype, self.device_id = dtype, int(device_id)
        self._dev = torch.device("cuda", self.device_id)
        self._mu = threading.Lock()
        h = C.c_void_p()
        _lib.check(self._lib.fw_aesrgan_create(self.device_id, self.num_block, self.scale, self.num_attention, _lib.DTYPES[dtype], C.byref(h)))
        self._h = h
        self._loaded = False

    def load_state_dict(self, state: Mapping[str, object], attention: Mapping[str, object]) -> None:
        """``state``: the RRDB trunk / tail under BasicSR's key names (``conv_first``, ``body.{i}.rdb{1,2,3}.conv{1..5}``,
        ``conv_body``, ``conv_up1`` [, ``conv_up2``], ``conv_hr``, ``conv_last``); ``attention``: ``attn.{i}.query|key|value.
        weight|bias`` and ``attn.{i}.gamma`` for the block behind RRDB ``i`` (synth.synthetic_attention_state has the layout)."""
        names = ["conv_first", "conv_body", "conv_up1", "conv_hr", "conv_last"] + (["conv_up2"] if self.scale >= 4 else [])
        for i in range(self.num_block):
            for r in (1, 2, 3):
                names += [f"body.{i}.rdb{r}.conv{c}" for c in range(1, 6)]
        items = []
        for k in names:
            if k + ".weight" not in state or k + ".bias" not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {k}")
            items += [(k + ".weight", state[k + ".weight"]), (k + ".bias", state[k + ".bias"])]
        for i in self.attn_after:
            for name in ("query", "key", "value"):
                items += [(f"attn.{i}.{name}.weight", attention[f"attn.{i}.{name}.weight"]), (f"attn.{i}.{name}.bias", attention[f"attn.{i}.{name}.bias"])]
            items.append((f"attn.{i}.gamma", attention[f"attn.{i}.gamma"]))
        for key, t in items:
            a = np.ascontiguousarray(_np(t), dtype=np.float32).reshape(-1)
            _lib.check(self._lib.fw_aesrgan_set_tensor(self._h, key.encode(), C.c_void_p(a.ctypes.data), a.size))
        _lib.check(self._lib.fw_aesrgan_finalize(self._h))
        self._loaded = True

    # ---- forward -------------------------------------------------------------------------------------------------------
    def forward_rgb(self, x):
        """x: float32 CUDA tensor H x W x 3, RGB in [0, 1].  Returns float32 (scale*H) x (scale*W) x 3, un-clamped — what
        ``AESRGAN.forward`` returns for a 1 x 3 x H x W input, NHWC (asynchronous on torch's current stream)."""
        import torch
        if not self._loaded:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "AESRGANEngine: no weights loaded")
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or x.shape[2] != 3:
            raise ValueError("forward_rgb expects a float32 CUDA tensor H x W x 3")
        if x.device != self._dev:
            raise ValueError(f"tensor is on {x.device}, engine on {self._dev}")
        x = x.contiguous()
        H, W = int(x.shape[0]), int(x.shape[1])
        out = torch.empty((H * self.scale, W * self.scale, 3), dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self._lib.fw_aesrgan_forward_rgb(self._h, C.c_void_p(x.data_ptr()), H, W, C.c_void_p(out.data_ptr()), st))
        return out

    def enhance(self, bgr: np.ndarray) -> np.ndarray:
        """uint8 BGR face crop -> uint8 BGR, scale x larger (the tensor round trip of AESRGANFaceRestorer: /255, RGB, network,
        clamp, x255)."""
        import torch
        if not isinstance(bgr, np.ndarray) or bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR image")
        # one forward at a time per instance: the launches of a forward are sequenced from this thread onto the device's current
        # stream, and a second thread on the same stream would interleave with them
        with self._mu, torch.cuda.device(self._dev):
            x = torch.from_numpy(np.ascontiguousarray(bgr[:, :, ::-1])).to(self._dev).float() / 255.0
            y = self.forward_rgb(x).clamp_(0, 1)
            out = (y * 255.0).round().to(torch.uint8).cpu().numpy()
        return np.ascontiguousarray(out[:, :, ::-1])

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.fw_aesrgan_destroy(h)
        self._loaded = False
