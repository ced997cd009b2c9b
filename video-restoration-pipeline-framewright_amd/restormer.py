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
from typing import Dict, List, Mapping, Sequence, Tuple

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


class _Conv3:
    """A bias-free 3x3 convolution on chunk-planar typed input, split into launches of 64 output channels, fp32 NHWC out."""

    def __init__(self, lib, dt, w: np.ndarray, dev, cin_pad: int = 0):
        import torch
        cout, cin = w.shape[:2]
        self.cin_pad, self.cout_pad = cin_pad or _pad(cin, 64), _pad(cout, 64)
        wp = np.zeros((self.cout_pad, self.cin_pad, 3, 3), np.float32)
        wp[:cout, :cin] = w
        self.groups = []
        chunks = self.cin_pad // 32
        for off in range(0, self.cout_pad, 64):
            sl = np.ascontiguousarray(wp[off:off + 64])
            n = lib.fw_pack_conv3x3(dt, None, 64, self.cin_pad, 2, chunks, None)
            buf = np.zeros(n, np.uint16)
            lib.fw_pack_conv3x3(dt, C.c_void_p(sl.ctypes.data), 64, self.cin_pad, 2, chunks, C.c_void_p(buf.ctypes.data))
            self.groups.append((torch.from_numpy(buf.view(np.int16)).to(dev), off))
        self.bias = torch.zeros(64, dtype=torch.float32, device=dev)


class RestormerEngine:
    """Restormer resident on one GPU; ``denoise_device`` mirrors NAFNetEngine (uint8 BGR in, uint8 BGR out)."""

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
        self.dtype, self.device_id = dtype, int(device_id)
        self._dt = _lib.DTYPES[dtype]
        self._tdt = torch.float16 if self._dt == _lib.FW_DTYPE_F16 else torch.bfloat16
        self._dev = torch.device("cuda", self.device_id)
        self._w: Dict[str, object] = {}

    # ---- weights ----------------------------------------------------------------------------------------------------
    def _pw(self, w2d: np.ndarray, k_pad: int):
        """[cout][k] fp32 -> packed pointwise fragments (cout padded to 64, k padded to k_pad); returns (tensor, cout_tiles)."""
        import torch
        cout, k = w2d.shape
        cp = _pad(cout, 64)
        wp = np.zeros((cp, k_pad), np.float32)
        wp[:cout, :k] = w2d
        n = self._lib.fw_pack_pointwise(self._dt, None, cp, k_pad, None)
        buf = np.zeros(n, np.uint16)
        if self._lib.fw_pack_pointwise(self._dt, C.c_void_p(wp.ctypes.data), cp, k_pad, C.c_void_p(buf.ctypes.data)) != n:
            raise FramewrightHipError(_lib.FW_ERR_INTERNAL, "fw_pack_pointwise failed")
        return torch.from_numpy(buf.view(np.int16)).to(self._dev), cp // 32

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        import torch
        if "params" in state:
            state = state["params"]  # type: ignore[assignment]
        elif "state_dict" in state:
            state = state["state_dict"]  # type: ignore[assignment]
        sd = {}
        for key, shape in restormer_tensor_shapes(**self.args):
            if key not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {key}")
            a = np.ascontiguousarray(_to_numpy(state[key]), dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{key}: expected shape {shape}, got {a.shape}")
            sd[key] = a
        dev = self._dev
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        W: Dict[str, object] = {}
        a = self.args
        for name, n, c, h in _stages(a["dim"], a["num_blocks"], a["num_refinement_blocks"], a["heads"]):
            cp, hid = _pad(c, 64), int(c * a["ffn_expansion_factor"])   # the GEMM writes whole pairs of 32-channel tiles
            hp = _pad(hid)
            for i in range(n):
                p = f"{name}.{i}."
                blk: Dict[str, object] = {"c": c, "cp": cp, "heads": h, "ch": c // h, "hp": hp}
                blk["n1w"], blk["n1b"] = f32(sd[p + "norm1.body.weight"]), f32(sd[p + "norm1.body.bias"])
                blk["n2w"], blk["n2b"] = f32(sd[p + "norm2.body.weight"]), f32(sd[p + "norm2.body.bias"])
                blk["temp"] = f32(sd[p + "attn.temperature"].reshape(-1))
                # qkv: rows re-laid to q @ 0, k @ cp, v @ 2cp
                wq = sd[p + "attn.qkv.weight"].reshape(3 * c, c)
                wqkv = np.zeros((3 * cp, c), np.float32)
                wdw = np.zeros((3 * cp, 9), np.float32)
                dws = sd[p + "attn.qkv_dwconv.weight"].reshape(3 * c, 9)
                for t in range(3):
                    wqkv[t * cp:t * cp + c] = wq[t * c:(t + 1) * c]
                    wdw[t * cp:t * cp + c] = dws[t * c:(t + 1) * c]
                blk["qkv"], blk["qkv_t"] = self._pw(wqkv, cp)
                blk["qkv_dw"] = f32(wdw)
                blk["proj"], blk["proj_t"] = self._pw(sd[p + "attn.project_out.weight"].reshape(c, c), cp)
                # GDFN: x1 rows @ 0, x2 rows @ hp
                wi = sd[p + "ffn.project_in.weight"].reshape(2 * hid, c)
                wi2 = np.zeros((2 * hp, c), np.float32)
                wi2[:hid], wi2[hp:hp + hid] = wi[:hid], wi[hid:]
                di = sd[p + "ffn.dwconv.weight"].reshape(2 * hid, 9)
                di2 = np.zeros((2 * hp, 9), np.float32)
                di2[:hid], di2[hp:hp + hid] = di[:hid], di[hid:]
                blk["pin"], blk["pin_t"] = self._pw(wi2, cp)
                blk["ffn_dw"] = f32(di2)
                blk["pout"], blk["pout_t"] = self._pw(sd[p + "ffn.project_out.weight"].reshape(c, hid), hp)
                W[p] = blk
        W["patch_embed.proj.weight"] = _Conv3(self._lib, self._dt, sd["patch_embed.proj.weight"], dev, cin_pad=32)
        for k in ("down1_2.body.0.weight", "down2_3.body.0.weight", "down3_4.body.0.weight",
                  "up4_3.body.0.weight", "up3_2.body.0.weight", "up2_1.body.0.weight", "output.weight"):
            W[k] = _Conv3(self._lib, self._dt, sd[k], dev)
        for k in ("reduce_chan_level3.weight", "reduce_chan_level2.weight"):
            w = sd[k].reshape(sd[k].shape[0], sd[k].shape[1])
            W[k] = self._pw(w, w.shape[1])      # K = 8*dim / 4*dim: multiples of 32
        W["ones"] = torch.ones(2048, dtype=torch.float32, device=dev)
        self._w = W

    # ---- forward ----------------------------------------------------------------------------------------------------
    @_lib.on_tensor_device
    def denoise_device(self, frame, out=None, out_rgb_f32=None):
        """torch.uint8 CUDA tensor H x W x 3 (H, W multiples of 8, as the network's three PixelUnshuffles require) -> same
        shape; asynchronous on torch's current stream."""
        import torch
        if not self._w:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "RestormerEngine: no weights loaded")
        if frame.dtype != torch.uint8 or not frame.is_cuda or frame.dim() != 3 or frame.shape[2] != 3 or not frame.is_contiguous():
            raise ValueError("denoise_device expects a contiguous uint8 CUDA tensor H x W x 3")
        H, Wd = int(frame.shape[0]), int(frame.shape[1])
        if H % 8 or Wd % 8:
            raise ValueError(f"Restormer needs frame sizes divisible by 8, got {Wd}x{H}")
        lib, dev, dt, W, a = self._lib, frame.device, self._dt, self._w, self.args
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        typ = lambda *s: torch.empty(s, dtype=self._tdt, device=dev)
        if out is None and out_rgb_f32 is None:
            out = torch.empty_like(frame)

        def conv3(cv: _Conv3, x_f32, h, w, cstride):
            """3x3 conv of the fp32 stream x_f32 [h*w][cstride] -> fp32 [h*w][cout_pad]."""
            M = h * w
            xp = typ(cv.cin_pad // 32, M, 32)
            if cstride != cv.cin_pad:
                raise FramewrightHipError(_lib.FW_ERR_INTERNAL, "conv3: stream stride does not match the padded input channels")
            _lib.check(lib.fw_f32_to_planar(dt, p(x_f32), M, cv.cin_pad, p(xp), st))
            y = f32(M, cv.cout_pad)
            for wpk, off in cv.groups:
                _lib.check(lib.fw_conv3x3_nhwc_ex(dt, p(xp), 32, M * 32, cv.cin_pad // 32, h, w, p(wpk), p(cv.bias), 2, 0, 0,
                                                  None, 1.0, None, 1.0, None, 0, cv.cout_pad, off, None, 32, 0, 0, p(y), st))
            return y

        def block(blk, x, h, w):
            """x: fp32 [h*w][cp], updated in place."""
            M, c, cp, hp, heads, ch = h * w, blk["c"], blk["cp"], blk["hp"], blk["heads"], blk["ch"]
            t = typ(M, cp)
            _lib.check(lib.fw_layernorm_nhwc(dt, p(x), cp, M, c, p(blk["n1w"]), p(blk["n1b"]), 1e-5, p(t), cp, cp, st))
            qkv = typ(M, 3 * cp)
            _lib.check(lib.fw_pointwise_nhwc(dt, p(t), 0, cp, M, cp, p(blk["qkv"]), None, blk["qkv_t"], p(qkv), 3 * cp, None, 0,
                                             None, None, st))
            qkv2 = typ(M, 3 * cp)
            _lib.check(lib.fw_dwconv3x3_nhwc(dt, p(qkv), 3 * cp, h, w, 3 * cp, p(blk["qkv_dw"]), 0, p(qkv2), 3 * cp, st))
            ws = f32(lib.fw_attn_workspace_floats(heads, ch))
            attn = f32(heads, ch, ch)
            scratch = typ(int(lib.fw_attn_qk_scratch_elems(M, heads, ch)))
            _lib.check(lib.fw_attn_matrix_mfma(dt, p(qkv2), 3 * cp, M, cp, heads, ch, p(blk["temp"]), p(ws), p(scratch), p(attn), st))
            # attn @ v as a 1x1 convolution with the block-diagonal attention matrix on the MFMA GEMM
            apk = torch.empty(int(lib.fw_pack_pointwise(dt, None, cp, cp, None)), dtype=torch.int16, device=dev)
            _lib.check(lib.fw_attn_pack(dt, p(attn), heads, ch, cp, p(apk), st))
            _lib.check(lib.fw_pointwise_nhwc(dt, C.c_void_p(qkv2.data_ptr() + 2 * cp * 2), 0, 3 * cp, M, cp, p(apk), None, cp // 32,
                                             p(t), cp, None, 0, None, None, st))
            _lib.check(lib.fw_pointwise_nhwc(dt, p(t), 0, cp, M, cp, p(blk["proj"]), None, blk["proj_t"], None, 0, p(x), cp, p(x),
                                             p(W["ones"]), st))
            _lib.check(lib.fw_layernorm_nhwc(dt, p(x), cp, M, c, p(blk["n2w"]), p(blk["n2b"]), 1e-5, p(t), cp, cp, st))
            g = typ(M, 2 * hp)
            _lib.check(lib.fw_pointwise_nhwc(dt, p(t), 0, cp, M, cp, p(blk["pin"]), None, blk["pin_t"], p(g), 2 * hp, None, 0, None,
                                             None, st))
            g2 = typ(M, hp)
            _lib.check(lib.fw_dwconv3x3_nhwc(dt, p(g), 2 * hp, h, w, 2 * hp, p(blk["ffn_dw"]), 1, p(g2), hp, st))
            _lib.check(lib.fw_pointwise_nhwc(dt, p(g2), 0, hp, M, hp, p(blk["pout"]), None, blk["pout_t"], None, 0, p(x), cp, p(x),
                                             p(W["ones"]), st))

        def stage(name, n, x, h, w):
            for i in range(n):
                block(W[f"{name}.{i}."], x, h, w)
            return x

        def down(key, x, h, w, c):
            """conv3x3 (c -> c/2) + PixelUnshuffle(2): fp32 [h*w][pad(c)] -> fp32 [(h/2)*(w/2)][pad(2c)]."""
            y = conv3(W[key], x, h, w, _pad(c, 64))
            o = torch.zeros((h // 2) * (w // 2), _pad(2 * c, 64), dtype=torch.float32, device=dev)
            _lib.check(lib.fw_pixel_shuffle2_f32(p(y), y.shape[1], h // 2, w // 2, c // 2, p(o), o.shape[1], 0, 1, st))
            return o

        def up_cat(key, x, h, w, c, skip):
            """cat([PixelShuffle(2)(conv3x3 (c -> 2c)(x)), skip], channels): fp32 [(2h)*(2w)][pad64(c)], c/2 + c/2 channels."""
            y = conv3(W[key], x, h, w, _pad(c, 64))
            o = torch.zeros(4 * h * w, _pad(c, 64), dtype=torch.float32, device=dev)
            _lib.check(lib.fw_pixel_shuffle2_f32(p(y), y.shape[1], h, w, c // 2, p(o), o.shape[1], 0, 0, st))
            _lib.check(lib.fw_copy_channels_f32(p(skip), skip.shape[1], 4 * h * w, c // 2, p(o), o.shape[1], c // 2, st))
            return o

        def reduce(key, x, M, cin, cout):
            wpk, tiles = W[key]
            o = f32(M, 32 * tiles)
            _lib.check(lib.fw_pointwise_nhwc(dt, p(x), 1, x.shape[1], M, cin, p(wpk), None, tiles, None, 0, p(o), 32 * tiles, None, None, st))
            return o

        d, nb, nr, hd = a["dim"], a["num_blocks"], a["num_refinement_blocks"], a["heads"]
        x0 = typ(H, Wd, 32)
        _lib.check(lib.fw_u8_to_nhwc(dt, p(frame), H, Wd, p(x0), 32, st))
        pe = W["patch_embed.proj.weight"]
        e1 = f32(H * Wd, pe.cout_pad)
        for wpk, off in pe.groups:
            _lib.check(lib.fw_conv3x3_nhwc_ex(dt, p(x0), 32, 0, 1, H, Wd, p(wpk), p(pe.bias), 2, 0, 0, None, 1.0, None, 1.0, None, 0,
                                              pe.cout_pad, off, None, 32, 0, 0, p(e1), st))
        e1 = stage("encoder_level1", nb[0], e1, H, Wd)
        e2 = stage("encoder_level2", nb[1], down("down1_2.body.0.weight", e1, H, Wd, d), H // 2, Wd // 2)
        e3 = stage("encoder_level3", nb[2], down("down2_3.body.0.weight", e2, H // 2, Wd // 2, 2 * d), H // 4, Wd // 4)
        lat = stage("latent", nb[3], down("down3_4.body.0.weight", e3, H // 4, Wd // 4, 4 * d), H // 8, Wd // 8)
        d3 = up_cat("up4_3.body.0.weight", lat, H // 8, Wd // 8, 8 * d, e3)
        d3 = stage("decoder_level3", nb[2], reduce("reduce_chan_level3.weight", d3, (H // 4) * (Wd // 4), 8 * d, 4 * d), H // 4, Wd // 4)
        d2 = up_cat("up3_2.body.0.weight", d3, H // 4, Wd // 4, 4 * d, e2)
        d2 = stage("decoder_level2", nb[1], reduce("reduce_chan_level2.weight", d2, (H // 2) * (Wd // 2), 4 * d, 2 * d), H // 2, Wd // 2)
        d1 = up_cat("up2_1.body.0.weight", d2, H // 2, Wd // 2, 2 * d, e1)
        d1 = stage("decoder_level1", nb[0], d1, H, Wd)
        d1 = stage("refinement", nr, d1, H, Wd)
        y = conv3(W["output.weight"], d1, H, Wd, _pad(2 * d, 64))
        _lib.check(lib.fw_tap_post_u8(p(frame), p(y), H, Wd, Wd, y.shape[1], p(out), p(out_rgb_f32), st))
        return out if out is not None else out_rgb_f32

    def denoise(self, frame_bgr: np.ndarray) -> np.ndarray:
        import torch
        f = np.ascontiguousarray(frame_bgr)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR frame")
        with torch.cuda.device(self._dev):
            o = self.denoise_device(torch.from_numpy(f).to(self._dev))
            torch.cuda.synchronize(self._dev)
        return o.cpu().numpy()

    def clone(self) -> "RestormerEngine":
        e = RestormerEngine(dtype=self.dtype, device_id=self.device_id, **self.args)
        e._w = self._w      # weights are read-only on the device; activations are per call
        return e

    def close(self) -> None:
        self._w = {}
