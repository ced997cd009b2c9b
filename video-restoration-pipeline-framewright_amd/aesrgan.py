"""The reference's in-tree ``AESRGAN`` — RRDB trunk with full-image self-attention blocks — on the MI355X kernels.

Reference: ``src/framewright/processors/aesrgan_face.py`` — ``AttentionBlock`` (:142-168), ``ResidualDenseBlock`` / ``RRDB``
(:171-204), ``AESRGAN`` (:205-269: no pixel-unshuffle front end, an AttentionBlock behind every ``num_block // num_attention``-th
RRDB, ``conv_up2`` only for scale >= 4), the network ``AESRGANFaceRestorer`` (:383+) runs on face crops.  SURVEY.md section
8f/4 ("other RRDB consumers").  Face detection and paste-back of that class are cv2 host code and stay out of scope.

Host-sequenced over the C-ABI building blocks, like restormer.py (face crops are small; the 1080p Real-ESRGAN path has its
own fused engine in csrc/rrdbnet.hip): ``fw_conv3x3_nhwc`` for every 3x3 convolution (chunk-planar typed activations, the
residual streams in fp32, `x5 * 0.2 + x` and the RRDB's second residual in the conv5 epilogue), ``fw_pointwise_nhwc`` for the
1x1 query / key / value projections, ``fw_attn_softmax_rows`` for softmax(q^T k) over all pixels, and the product with v as
a GEMM over the pixel axis (``fw_pack_pointwise_transposed`` + ``fw_pointwise_nhwc`` with the gamma residual epilogue).
Parity: oracle/rrdbnet_ref.py ``aesrgan_forward``, which is pinned on vectors the reference's own module produced
(tests/golden/aesrgan_attention.npz).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, List, Mapping, Optional

import numpy as np

from . import _lib
from ._lib import FramewrightHipError
from .synth import aesrgan_attention_positions


def _np(t) -> np.ndarray:
    return t if isinstance(t, np.ndarray) else t.detach().cpu().float().numpy()


class _Conv:
    """Packed 3x3 convolution: weight fragments + bias padded to the 32-channel output tiles."""

    def __init__(self, lib, dt, w: np.ndarray, b: np.ndarray, dev):
        import torch
        cout, cin = int(w.shape[0]), int(w.shape[1])
        self.ct, self.chunks = (cout + 31) // 32, (cin + 31) // 32
        w = np.ascontiguousarray(w, np.float32)
        n = lib.fw_pack_conv3x3(dt, None, cout, cin, self.ct, self.chunks, None)
        buf = np.zeros(n, np.uint16)
        if lib.fw_pack_conv3x3(dt, C.c_void_p(w.ctypes.data), cout, cin, self.ct, self.chunks, C.c_void_p(buf.ctypes.data)) != n:
            raise FramewrightHipError(_lib.FW_ERR_INTERNAL, "fw_pack_conv3x3 failed")
        self.w = torch.from_numpy(buf.view(np.int16)).to(dev)
        bp = np.zeros(32 * self.ct, np.float32)
        bp[:cout] = b
        self.b = torch.from_numpy(bp).to(dev)


class AESRGANEngine:
    """``AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block, scale, num_attention)`` resident on one GPU."""

    def __init__(self, num_block: int = 23, scale: int = 2, num_attention: int = 4, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if scale not in (2, 4) or num_block < 1 or num_attention < 1 or num_attention > num_block:
            raise ValueError("AESRGANEngine: scale 2 or 4, 1 <= num_attention <= num_block")
        self.num_block, self.scale, self.num_attention = int(num_block), int(scale), int(num_attention)
        self.attn_after = aesrgan_attention_positions(self.num_block, self.num_attention)
        self.dtype, self.device_id = dtype, int(device_id)
        self._dt = _lib.DTYPES[dtype]
        self._tdt = torch.float16 if self._dt == _lib.FW_DTYPE_F16 else torch.bfloat16
        self._dev = torch.device("cuda", self.device_id)
        self._mu = threading.Lock()
        self._c: Dict[str, _Conv] = {}
        self._a: Dict[int, dict] = {}

    # ---- weights -------------------------------------------------------------------------------------------------------
    def _pw(self, w2d: np.ndarray, bias: np.ndarray):
        """[cout][64] 1x1 weight -> fw_pack_pointwise fragments (cout padded to 32) + padded bias."""
        import torch
        cout = w2d.shape[0]
        cp = ((cout + 31) // 32) * 32
        wp = np.zeros((cp, 64), np.float32)
        wp[:cout] = w2d
        n = self._lib.fw_pack_pointwise(self._dt, None, cp, 64, None)
        buf = np.zeros(n, np.uint16)
        if self._lib.fw_pack_pointwise(self._dt, C.c_void_p(wp.ctypes.data), cp, 64, C.c_void_p(buf.ctypes.data)) != n:
            raise FramewrightHipError(_lib.FW_ERR_INTERNAL, "fw_pack_pointwise failed")
        bp = np.zeros(cp, np.float32)
        bp[:cout] = bias
        return torch.from_numpy(buf.view(np.int16)).to(self._dev), torch.from_numpy(bp).to(self._dev), cp // 32

    def load_state_dict(self, state: Mapping[str, object], attention: Mapping[str, object]) -> None:
        """``state``: the RRDB trunk / tail under BasicSR's key names (``conv_first``, ``body.{i}.rdb{1,2,3}.conv{1..5}``,
        ``conv_body``, ``conv_up1`` [, ``conv_up2``], ``conv_hr``, ``conv_last``); ``attention``: ``attn.{i}.query|key|value.
        weight|bias`` and ``attn.{i}.gamma`` for the block behind RRDB ``i`` (synth.synthetic_attention_state has the layout)."""
        import torch
        names = ["conv_first", "conv_body", "conv_up1", "conv_hr", "conv_last"] + (["conv_up2"] if self.scale >= 4 else [])
        for i in range(self.num_block):
            for r in (1, 2, 3):
                names += [f"body.{i}.rdb{r}.conv{c}" for c in range(1, 6)]
        for k in names:
            if k + ".weight" not in state or k + ".bias" not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {k}")
            self._c[k] = _Conv(self._lib, self._dt, _np(state[k + ".weight"]), _np(state[k + ".bias"]), self._dev)
        for i in self.attn_after:
            blk = {}
            for name in ("query", "key", "value"):
                w = _np(attention[f"attn.{i}.{name}.weight"]).reshape(-1, 64)
                blk[name] = self._pw(w, _np(attention[f"attn.{i}.{name}.bias"]))
            g = float(_np(attention[f"attn.{i}.gamma"]).reshape(-1)[0])
            blk["gamma"] = torch.full((64,), g, dtype=torch.float32, device=self._dev)
            self._a[i] = blk

    # ---- forward -------------------------------------------------------------------------------------------------------
    def forward_rgb(self, x):
        """x: float32 CUDA tensor H x W x 3, RGB in [0, 1].  Returns float32 (scale*H) x (scale*W) x 3, un-clamped — what
        ``AESRGAN.forward`` returns for a 1 x 3 x H x W input, NHWC."""
        import torch
        if not self._c:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "AESRGANEngine: no weights loaded")
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or x.shape[2] != 3:
            raise ValueError("forward_rgb expects a float32 CUDA tensor H x W x 3")
        lib, dt, dev = self._lib, self._dt, x.device
        H, W = int(x.shape[0]), int(x.shape[1])
        M = H * W
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t, off=0: C.c_void_p(t.data_ptr() + off) if t is not None else None
        esz = 2
        typed = lambda *s: torch.empty(s, dtype=self._tdt, device=dev)
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)

        def conv(name, src, src_cstride, src_pstride, h, w, out=None, out_off=0, out_pstride=0, act=0, ups=0, res1=None, s1=1.0,
                 res2=None, s2=1.0, out_f32=None):
            cv = self._c[name]
            _lib.check(lib.fw_conv3x3_nhwc(dt, p(src), src_cstride, src_pstride, cv.chunks, h, w, p(cv.w), p(cv.b), cv.ct, act, ups,
                                           p(res1), float(s1), p(res2), float(s2), p(out, out_off) if out is not None else None, 32,
                                           out_pstride, 0, p(out_f32), st))

        # input: typed NHWC, 3 of 32 channels used
        img = torch.zeros((M, 32), dtype=self._tdt, device=dev)
        img[:, :3] = x.reshape(M, 3).to(self._tdt)
        PL = M * 32                                             # elements per 32-channel plane
        cat = [typed(6, M, 32), typed(6, M, 32)]
        feat = f32(M, 64)
        pool = [f32(M, 64) for _ in range(4)]                   # rotating fp32 streams; `feat` is never overwritten

        def free_buf(*busy):
            return next(t for t in pool if all(t is not q for q in busy))

        conv("conv_first", img, 32, 32, H, W, out=cat[0], out_pstride=PL, out_f32=feat)
        cur_f, cur = feat, 0                                    # the trunk: fp32 stream + its typed copy in planes 0,1 of cat[cur]
        for i in range(self.num_block):
            rrdb_in = x_f = cur_f
            for r in (1, 2, 3):
                pre = f"body.{i}.rdb{r}."
                X = cat[cur]
                for cidx in range(1, 5):                        # x1..x4 -> planes 2..5 (aesrgan_face.py:184-187)
                    conv(pre + f"conv{cidx}", X, 32, PL, H, W, out=X, out_off=(1 + cidx) * PL * esz, act=1)
                y_f = free_buf(x_f, rrdb_in)
                if r < 3:                                       # x5 * 0.2 + x  (:188-189)
                    conv(pre + "conv5", X, 32, PL, H, W, out=cat[1 - cur], out_pstride=PL, res1=x_f, s1=0.2, out_f32=y_f)
                else:                                           # ... and the RRDB's own residual: * 0.2 + rrdb_in (:204)
                    conv(pre + "conv5", X, 32, PL, H, W, out=cat[1 - cur], out_pstride=PL, res1=x_f, s1=0.2, res2=rrdb_in, s2=0.2,
                         out_f32=y_f)
                x_f, cur = y_f, 1 - cur
            cur_f = x_f
            if i in self._a:                                    # AttentionBlock behind this RRDB (:229-233)
                cur_f = self._attention(self._a[i], cur_f, M, st, dev)
                _lib.check(lib.fw_f32_to_planar(dt, p(cur_f), M, 64, p(cat[cur]), st))
        # feat + conv_body(body)  (:256-257)
        body = typed(2, M, 32)
        conv("conv_body", cat[cur], 32, PL, H, W, out=body, out_pstride=PL, res1=feat, s1=1.0)
        h2, w2 = 2 * H, 2 * W
        u1 = typed(2, h2 * w2, 32)
        conv("conv_up1", body, 32, PL, h2, w2, out=u1, out_pstride=h2 * w2 * 32, act=1, ups=1)          # nearest x2 + conv + lrelu
        top, ht, wt = u1, h2, w2
        if self.scale >= 4:
            h4, w4 = 2 * h2, 2 * w2
            u2 = typed(2, h4 * w4, 32)
            conv("conv_up2", u1, 32, h2 * w2 * 32, h4, w4, out=u2, out_pstride=h4 * w4 * 32, act=1, ups=1)
            top, ht, wt = u2, h4, w4
        hr = typed(2, ht * wt, 32)
        conv("conv_hr", top, 32, ht * wt * 32, ht, wt, out=hr, out_pstride=ht * wt * 32, act=1)
        last = f32(ht * wt, 32)
        conv("conv_last", hr, 32, ht * wt * 32, ht, wt, out_f32=last)
        return last[:, :3].reshape(ht, wt, 3).contiguous()

    def _attention(self, blk, x_f, M, st, dev):
        """gamma * (v @ softmax(q^T k)^T) + x on the fp32 stream x_f [M][64]; returns a new fp32 tensor."""
        import torch
        lib, dt = self._lib, self._dt
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        outs = {}
        for name in ("query", "key", "value"):
            wpk, bias, tiles = blk[name]
            o = torch.empty((M, 32 * tiles), dtype=self._tdt, device=dev)
            _lib.check(lib.fw_pointwise_nhwc(dt, p(x_f), 1, 64, M, 64, p(wpk), p(bias), tiles, p(o), 32 * tiles, None, 0, None, None, st))
            outs[name] = o
        kp = ((M + 31) // 32) * 32
        P = torch.empty((M, kp), dtype=self._tdt, device=dev)
        _lib.check(lib.fw_attn_softmax_rows(dt, p(outs["query"]), 32, p(outs["key"]), 32, M, 8, p(P), kp, st))
        vt = torch.empty(int(lib.fw_pack_pointwise(dt, None, 64, kp, None)), dtype=torch.int16, device=dev)
        _lib.check(lib.fw_pack_pointwise_transposed(dt, p(outs["value"]), 64, M, 64, kp, p(vt), st))
        y = torch.empty_like(x_f)
        _lib.check(lib.fw_pointwise_nhwc(dt, p(P), 0, kp, M, kp, p(vt), None, 2, None, 0, p(y), 64, p(x_f), p(blk["gamma"]), st))
        return y

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
        self._c, self._a = {}, {}
