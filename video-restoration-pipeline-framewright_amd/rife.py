"""RIFE x2 frame interpolation on MI355X (kernel set K7).

The reference's ``FrameInterpolator`` (``src/framewright/processors/interpolation.py``) shells out to the external
binary ``rife-ncnn-vulkan`` for every pass (:628-650).  Here the IFNet v4.6 arithmetic runs in
libframewright_hip.so: every convolution of the four IFBlocks goes through the MFMA conv3x3 kernel
(``fw_conv3x3_nhwc_ex``: the stride-2 convs as 3x3 convs on a pixel-unshuffled tensor, the ConvTranspose2d(4,2,1) as a 3x3
conv with four output parities), resize / backward warp / mask blend through the HBM-bound kernels of
``csrc/ifnet_ops.hip``.  This module holds the weight transforms, the launch order (PyTorch is used for device memory and
streams only) and the directory-level bookkeeping of ``FrameInterpolator.interpolate`` / ``interpolate_to_fps``.
"""
from __future__ import annotations

import ctypes as C
import logging
import shutil
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib, policy
from ._lib import FramewrightHipError
from .realesrgan import _imread, _imwrite, _to_numpy
from .synth import IFNET_CHANNELS, IFNET_SCALES, ifnet_tensor_shapes

logger = logging.getLogger(__name__)


class InterpolationError(Exception):
    """Same name as the reference's exception (interpolation.py:45)."""


def _pad(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def stride2_as_unshuffled_3x3(w: np.ndarray) -> np.ndarray:
    """Conv2d(cin, cout, 3, stride 2, pad 1) == 3x3/s1/p1 conv on pixel_unshuffle(x, 2) with weights
    W'[co][ci*4 + dy*2 + dx][U][V]: tap ky -> (U, dy) = {0: (0, 1), 1: (1, 0), 2: (1, 1)}; the U = V = 2 taps are zero."""
    cout, cin = w.shape[:2]
    out = np.zeros((cout, cin * 4, 3, 3), np.float32)
    m = {0: (0, 1), 1: (1, 0), 2: (1, 1)}
    for ky in range(3):
        for kx in range(3):
            (U, dy), (V, dx) = m[ky], m[kx]
            out[:, dy * 2 + dx::4, U, V] = w[:, :, ky, kx]
    return out


def convtranspose_as_3x3(w: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """ConvTranspose2d(cin, cout, 4, stride 2, pad 1) == 3x3 conv producing cout*4 channels n = co*4 + py*2 + px (the output
    parity), out[co][2y+py][2x+px].  Tap dy (input row y+dy) uses kernel row ky = py + 1 - 2*dy when 0 <= ky <= 3."""
    cin, cout = w.shape[:2]
    out = np.zeros((cout * 4, cin, 3, 3), np.float32)
    for py in range(2):
        for px in range(2):
            for dy in (-1, 0, 1):
                ky = py + 1 - 2 * dy
                if not 0 <= ky <= 3:
                    continue
                for dx in (-1, 0, 1):
                    kx = px + 1 - 2 * dx
                    if not 0 <= kx <= 3:
                        continue
                    out[py * 2 + px::4, :, dy + 1, dx + 1] = w[:, :, ky, kx].T
    return out, np.repeat(b.astype(np.float32), 4)


@dataclass
class _Group:
    w: object          # packed weights (torch int16 CUDA)
    b: object          # fp32 bias padded to 32*ct
    ct: int
    off: int           # first output channel


class _Conv:
    """One 3x3 convolution split into launches of <= 64 output channels."""

    def __init__(self, lib, dtype_id: int, w: np.ndarray, b: np.ndarray, cin_pad: int, cout_pad: int, device, pairs_only: bool):
        import torch
        cout, cin = w.shape[:2]
        wp = np.zeros((cout_pad, cin_pad, 3, 3), np.float32)
        wp[:cout, :cin] = w
        bp = np.zeros((cout_pad,), np.float32)
        bp[:cout] = b
        self.cin_pad, self.cout_pad = cin_pad, cout_pad
        self.groups: List[_Group] = []
        off = 0
        while off < cout_pad:
            ct = 2 if cout_pad - off >= 64 else 1
            if pairs_only and ct != 2:
                raise ValueError("residual convolutions need output channels in multiples of 64")
            sl = np.ascontiguousarray(wp[off:off + 32 * ct])
            n = lib.fw_pack_conv3x3(dtype_id, None, 32 * ct, cin_pad, ct, cin_pad // 32, None)
            buf = np.zeros(n, np.uint16)
            lib.fw_pack_conv3x3(dtype_id, C.c_void_p(sl.ctypes.data), 32 * ct, cin_pad, ct, cin_pad // 32,
                                C.c_void_p(buf.ctypes.data))
            self.groups.append(_Group(torch.from_numpy(buf.view(np.int16)).to(device),
                                      torch.from_numpy(np.ascontiguousarray(bp[off:off + 32 * ct])).to(device), ct, off))
            off += 32 * ct


class IFNetEngine:
    """IFNet v4.6 resident on one GPU."""

    def __init__(self, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        self.dtype, self.device_id = dtype, int(device_id)
        self._dt = _lib.DTYPES[dtype]
        self._tdt = torch.float16 if self._dt == _lib.FW_DTYPE_F16 else torch.bfloat16
        self._dev = torch.device("cuda", self.device_id)
        self._blocks: List[Dict[str, object]] = []

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        import torch
        sd = {}
        for key, shape in ifnet_tensor_shapes():
            if key not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {key}")
            a = np.ascontiguousarray(_to_numpy(state[key]), dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"{key}: expected shape {shape}, got {a.shape}")
            sd[key] = a
        self._blocks = []
        for i, c in enumerate(IFNET_CHANNELS):
            p = f"block{i}."
            cin = 7 if i == 0 else 12
            c2p, cp = _pad(c // 2, 32), _pad(c, 64)
            blk: Dict[str, object] = {"c": c, "cin": cin, "c2p": c2p, "cp": cp}
            blk["conv0_0"] = _Conv(self._lib, self._dt, stride2_as_unshuffled_3x3(sd[p + "conv0.0.0.weight"]),
                                   sd[p + "conv0.0.0.bias"], _pad(4 * cin, 32), c2p, self._dev, False)
            w1 = np.zeros((c, c2p, 3, 3), np.float32)
            w1[:, :c // 2] = sd[p + "conv0.1.0.weight"]
            blk["conv0_1"] = _Conv(self._lib, self._dt, stride2_as_unshuffled_3x3(w1), sd[p + "conv0.1.0.bias"], 4 * c2p, cp,
                                   self._dev, False)
            res = []
            for j in range(8):
                q = f"{p}convblock.{j}."
                beta = np.zeros((cp,), np.float32)
                beta[:c] = sd[q + "beta"].reshape(-1)
                res.append((_Conv(self._lib, self._dt, sd[q + "conv.weight"], sd[q + "conv.bias"], cp, cp, self._dev, True),
                            torch.from_numpy(beta).to(self._dev)))
            blk["res"] = res
            w3, b3 = convtranspose_as_3x3(sd[p + "lastconv.0.weight"], sd[p + "lastconv.0.bias"])
            blk["last"] = _Conv(self._lib, self._dt, w3, b3, cp, 96, self._dev, False)
            self._blocks.append(blk)

    # -- launch helpers ---------------------------------------------------------------------------------------
    def _conv(self, conv: _Conv, x, h: int, w: int, st, out=None, out_f32=None, act=0, res=None, beta=None, post_act=0):
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        for g in conv.groups:
            _lib.check(self._lib.fw_conv3x3_nhwc_ex(
                self._dt, p(x), conv.cin_pad, 0, conv.cin_pad // 32, h, w, p(g.w), p(g.b), g.ct, act, 0,
                p(res), 1.0, None, 1.0, C.c_void_p(beta.data_ptr() + 4 * g.off) if beta is not None else None, post_act,
                conv.cout_pad, g.off, p(out), conv.cout_pad, 0, g.off, p(out_f32), st))

    def interpolate_device(self, img0, img1, timestep: float = 0.5, out=None, out_rgb_f32=None):
        """img0/img1: uint8 CUDA tensors H x W x 3 (BGR).  Returns the uint8 mid frame (asynchronous on torch's current
        stream)."""
        import torch
        if not self._blocks:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "IFNetEngine: no weights loaded")
        for t in (img0, img1):
            if t.dtype != torch.uint8 or not t.is_cuda or t.dim() != 3 or t.shape[2] != 3 or not t.is_contiguous():
                raise ValueError("interpolate_device expects contiguous uint8 CUDA tensors H x W x 3")
        if img0.shape != img1.shape:
            raise ValueError("frame sizes differ")
        lib, dev = self._lib, img0.device
        H, W = int(img0.shape[0]), int(img0.shape[1])
        Hp, Wp = _pad(H, 32), _pad(W, 32)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        typ = lambda *s: torch.empty(s, dtype=self._tdt, device=dev)
        I0, I1 = f32(Hp, Wp, 3), f32(Hp, Wp, 3)
        _lib.check(lib.fw_u8_to_rgb_f32(p(img0), H, W, Hp, Wp, p(I0), st))
        _lib.check(lib.fw_u8_to_rgb_f32(p(img1), H, W, Hp, Wp, p(I1), st))
        flow, mask = f32(Hp, Wp, 4), f32(Hp, Wp, 1)
        for i, (blk, s) in enumerate(zip(self._blocks, IFNET_SCALES)):
            first = i == 0
            cin, c2p, cp = blk["cin"], blk["c2p"], blk["cp"]
            X = f32(Hp, Wp, 7 if first else 8)
            _lib.check(lib.fw_ifnet_build_x(p(I0), p(I1), None if first else p(flow), None if first else p(mask), Hp, Wp,
                                            float(timestep), p(X), st))
            hs, ws = Hp // s, Wp // s
            xin = f32(hs, ws, cin)
            _lib.check(lib.fw_resize_bilinear_f32(p(X), Hp, Wp, X.shape[2], p(xin), hs, ws, cin, 0, 1.0 / s, 1.0, st))
            if not first:
                _lib.check(lib.fw_resize_bilinear_f32(p(flow), Hp, Wp, 4, p(xin), hs, ws, cin, 8, 1.0 / s, 1.0 / s, st))
            # conv0: two stride-2 convs (+LeakyReLU) as 3x3 convs on pixel-unshuffled tensors
            c00, c01 = blk["conv0_0"], blk["conv0_1"]
            u0 = typ(hs // 2, ws // 2, c00.cin_pad)
            _lib.check(lib.fw_unshuffle2_cast(self._dt, p(xin), 1, hs, ws, cin, cin, p(u0), c00.cin_pad, st))
            a0 = typ(hs // 2, ws // 2, c2p)
            self._conv(c00, u0, hs // 2, ws // 2, st, out=a0, act=1)
            u1 = typ(hs // 4, ws // 4, c01.cin_pad)
            _lib.check(lib.fw_unshuffle2_cast(self._dt, p(a0), 0, hs // 2, ws // 2, c2p, c2p, p(u1), c01.cin_pad, st))
            hf, wf = hs // 4, ws // 4
            feat, feat32 = typ(hf, wf, cp), f32(hf, wf, cp)
            self._conv(c01, u1, hf, wf, st, out=feat, out_f32=feat32, act=1)
            nxt, nxt32 = typ(hf, wf, cp), f32(hf, wf, cp)
            for conv, beta in blk["res"]:   # ResConv: lrelu(conv(x) * beta + x)
                self._conv(conv, feat, hf, wf, st, out=nxt, out_f32=nxt32, res=feat32, beta=beta, post_act=1)
                feat, nxt = nxt, feat
                feat32, nxt32 = nxt32, feat32
            # lastconv: ConvTranspose2d(c, 24, 4, 2, 1) + PixelShuffle(2) -> 6 channels at (hs, ws)
            t96 = f32(hf, wf, 96)
            self._conv(blk["last"], feat, hf, wf, st, out_f32=t96)
            tmp = f32(hs, ws, 6)
            _lib.check(lib.fw_depth_to_space4_f32(p(t96), hf, wf, 96, p(tmp), st))
            _lib.check(lib.fw_ifnet_accumulate(p(tmp), hs, ws, Hp, Wp, float(s), p(flow), p(mask), 1 if first else 0, st))
        if out is None and out_rgb_f32 is None:
            out = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        _lib.check(lib.fw_ifnet_blend(p(I0), p(I1), p(flow), p(mask), Hp, Wp, H, W, p(out), p(out_rgb_f32), st))
        self._last_flow = flow
        return out if out is not None else out_rgb_f32

    def interpolate(self, img0: np.ndarray, img1: np.ndarray, timestep: float = 0.5) -> np.ndarray:
        import torch
        a = torch.from_numpy(np.ascontiguousarray(img0)).to(self._dev)
        b = torch.from_numpy(np.ascontiguousarray(img1)).to(self._dev)
        out = self.interpolate_device(a, b, timestep)
        torch.cuda.synchronize(self._dev)
        return out.cpu().numpy()


# ---- directory-level driver (FrameInterpolator) ---------------------------------------------------------------------
class FrameInterpolator:
    """Mirror of the reference class for the paths in scope (interpolation.py:530-809): ``interpolate`` doubles the
    frame count ``n`` passes in a row (x2 / x4 / x8 by factor), ``interpolate_to_fps`` adds the decimation loop.  Output
    naming ``frame_%08d.png`` as the reference passes to the binary (:634)."""

    def __init__(self, model: str = "rife-v4.6", gpu_id: int = 0, engine: Optional[IFNetEngine] = None, dtype: str = "f16"):
        self.model, self.gpu_id, self._engine, self._dtype = model, gpu_id, engine, dtype

    def _get_engine(self) -> IFNetEngine:
        if self._engine is None:
            import os
            from .synth import synthetic_ifnet_state
            eng = IFNetEngine(self._dtype, self.gpu_id)
            path = Path(os.environ.get("FRAMEWRIGHT_MODEL_DIR", str(Path.home() / ".framewright" / "models"))) / self.model / "flownet.pkl"
            if path.exists():
                import torch
                sd = torch.load(str(path), map_location="cpu", weights_only=True)
                eng.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})
            elif os.environ.get("FRAMEWRIGHT_AMD_SYNTHETIC_WEIGHTS") == "1":
                eng.load_state_dict(synthetic_ifnet_state())
            else:
                raise InterpolationError(f"IFNet weights not found: {path}")
            self._engine = eng
        return self._engine

    def double(self, frames: Sequence[np.ndarray]) -> List[np.ndarray]:
        """One x2 pass over an in-memory clip: [f0, mid01, f1, mid12, ..., f_{n-1}] (2n-1 frames)."""
        eng = self._get_engine()
        out: List[np.ndarray] = []
        for i, f in enumerate(frames):
            out.append(f)
            if i + 1 < len(frames):
                out.append(eng.interpolate(f, frames[i + 1]))
        return out

    def interpolate(self, input_dir: Path, output_dir: Path, source_fps: float, target_fps: float,
                    progress_callback: Optional[Callable[[float], None]] = None) -> Path:
        input_dir, output_dir = Path(input_dir), Path(output_dir)
        files = sorted(input_dir.glob("*.png"))
        if not files:
            raise InterpolationError(f"No frames found in {input_dir}")
        n_pass = policy.interpolation_exponent(target_fps / source_fps)
        frames = [_imread(f)[:, :, :3] for f in files]
        for k in range(n_pass):
            frames = self.double(frames)
            if progress_callback:
                progress_callback((k + 1) / n_pass)
        output_dir.mkdir(parents=True, exist_ok=True)
        for i, f in enumerate(frames):
            _imwrite(output_dir / f"frame_{i + 1:08d}.png", f)
        return output_dir

    def interpolate_to_fps(self, input_dir: Path, output_dir: Path, source_fps: float, target_fps: float,
                           progress_callback: Optional[Callable[[float], None]] = None) -> Tuple[Path, float]:
        input_dir, output_dir = Path(input_dir), Path(output_dir)
        if target_fps / source_fps <= 1.0:      # interpolation.py:752-758: copy through
            output_dir.mkdir(parents=True, exist_ok=True)
            for i, f in enumerate(sorted(input_dir.glob("*.png"))):
                shutil.copy(f, output_dir / f"frame_{i:08d}.png")
            return output_dir, source_fps
        interp_fps = policy.interp_fps_for_target(source_fps, target_fps)
        tmp = output_dir.parent / f"{output_dir.name}_temp"
        self.interpolate(input_dir, tmp, source_fps, interp_fps,
                         (lambda p: progress_callback(p * 0.7)) if progress_callback else None)
        if abs(interp_fps - target_fps) > 0.5:
            output_dir.mkdir(parents=True, exist_ok=True)
            files = sorted(tmp.glob("*.png"))
            for out_idx, i in enumerate(policy.decimation_indices(len(files), interp_fps, target_fps)):
                shutil.copy(files[i], output_dir / f"frame_{out_idx:08d}.png")
            shutil.rmtree(tmp)
            final = target_fps
        else:
            if output_dir.exists():
                shutil.rmtree(output_dir)
            tmp.rename(output_dir)
            final = interp_fps
        if progress_callback:
            progress_callback(1.0)
        return output_dir, final

    calculate_interpolation_factor = staticmethod(policy.calculate_interpolation_factor)

    def _detect_scene_by_histogram(self, img1: np.ndarray, img2: np.ndarray, scene_threshold: float = 0.3) -> bool:
        return policy.scene_change_by_histogram(img1, img2, scene_threshold)
