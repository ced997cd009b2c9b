"""Device-resident hand-off between the three stages of the hot path.

The reference chains `tap_denoise -> enhance -> interpolate` through three PNG directories
(core/restorer.py:3217-3329: each stage reads every frame back from disk, decodes it, uploads it, downloads the result and
encodes it again).  Here a clip stays in HBM from the first upload to the last download: the stages exchange uint8 CUDA
tensors, and every stage is exactly the engine call its directory driver makes, so the result is bit-identical to running
the three drivers one after the other.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import policy


class DeviceRestorationPipeline:
    """denoise (TAPDenoiser) -> upscale (RRDBNetEngine / SRVGGNetEngine) -> interpolate (IFNetEngine), any stage optional.

    ``interp_passes`` x2 passes of frame interpolation (1 -> 2n-1 frames, 2 -> 4n-3, ...), as
    `FrameInterpolator.interpolate` runs them for a fps ratio (`policy.interpolation_exponent`).
    """

    def __init__(self, denoiser=None, upscaler=None, interpolator=None, interp_passes: int = 1):
        self.denoiser, self.upscaler, self.interpolator = denoiser, upscaler, interpolator
        self.interp_passes = int(interp_passes)

    @classmethod
    def for_fps(cls, denoiser, upscaler, interpolator, source_fps: float, target_fps: float) -> "DeviceRestorationPipeline":
        return cls(denoiser, upscaler, interpolator, policy.interpolation_exponent(target_fps / source_fps))

    def run_device(self, frames: Sequence) -> List:
        """frames: uint8 BGR H x W x 3, numpy arrays or CUDA tensors.  Returns uint8 CUDA tensors (still on the device; the
        work is queued on torch's current stream)."""
        import torch
        dev = None
        for e in (self.upscaler, self.interpolator):
            if e is not None:
                dev = torch.device("cuda", e.device_id)
        if dev is None and self.denoiser is not None:
            dev = torch.device("cuda", self.denoiser.config.gpu_id)
        if dev is None:
            raise ValueError("DeviceRestorationPipeline: no stage configured")

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev) if isinstance(a, np.ndarray) else a.contiguous()

        with torch.cuda.device(dev):
            cur = [up(f) for f in frames]
            if self.denoiser is not None:
                cur = self.denoiser.denoise_clip_device(cur)
            if self.upscaler is not None:
                cur = [self.upscaler.upscale_device(f) for f in cur]
            if self.interpolator is not None:
                for _ in range(self.interp_passes):
                    nxt = []
                    for i, f in enumerate(cur):
                        nxt.append(f)
                        if i + 1 < len(cur):
                            nxt.append(self.interpolator.interpolate_device(f, cur[i + 1]))
                    cur = nxt
        return cur

    def run(self, frames: Sequence) -> List[np.ndarray]:
        import torch
        out = self.run_device(frames)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in out]
