"""Device-resident hand-off between the three stages of the hot path.

The reference chains `tap_denoise -> enhance -> interpolate` through three PNG directories
(core/restorer.py:3217-3329: each stage reads every frame back from disk, decodes it, uploads it, downloads the result and
encodes it again).  Here a clip stays in HBM from the first upload to the last download: the stages exchange uint8 CUDA
tensors, and every stage is exactly the engine call its directory driver makes, so the result is bit-identical to running
the three drivers one after the other.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence

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

    # ---- streaming form: the codec edge (codec.py) ---------------------------------------------------------------------------
    def _device(self):
        import torch
        for e in (self.upscaler, self.interpolator):
            if e is not None:
                return torch.device("cuda", e.device_id)
        if self.denoiser is not None:
            return torch.device("cuda", self.denoiser.config.gpu_id)
        raise ValueError("DeviceRestorationPipeline: no stage configured")

    def _gen_denoise(self, frames: Iterator, block: int):
        """The temporal-window stage over a stream: every frame goes through the network once (`denoise_only_device`), and a block's
        windows are formed as soon as the `window // 2` frames behind it have been denoised too - the neighbours of a block are
        handed to `denoise_clip_device` as already-denoised halos, exactly what the multi-GPU block partition does between ranks
        (sharding.py), so the frames equal the whole-clip call bit for bit and the stage holds `block + window` frames, not the clip."""
        half = self.denoiser.config.temporal_window // 2
        tail: List = []                      # the last `half` denoised frames in front of the block
        raw: List = []
        den: List = []
        eof = False
        it = iter(frames)
        while True:
            while not eof and len(raw) < block + half:
                chunk = []
                for f in it:
                    chunk.append(f)
                    if len(raw) + len(chunk) >= block + half:
                        break
                else:
                    eof = True
                if chunk:
                    den += self.denoiser.denoise_only_device(chunk)
                    raw += chunk
            if not raw:
                return
            n_out = len(raw) if eof else block
            yield from self.denoiser.denoise_clip_device(raw[:n_out], halo_before=tail, halo_after=den[n_out:n_out + half], denoised=den[:n_out])
            tail = (tail + den[:n_out])[-half:] if half else []
            raw, den = raw[n_out:], den[n_out:]

    def _gen_interp(self, frames: Iterator):
        prev = None
        for f in frames:
            if prev is not None:
                yield prev
                yield self.interpolator.interpolate_device(prev, f)
            prev = f
        if prev is not None:
            yield prev

    def stream_device(self, frames: Iterable, block: int = 8):
        """Generator form of `run_device` for clips that do not fit (or have not arrived) in memory: ``frames`` is any iterator of
        uint8 BGR frames (numpy - e.g. `codec.RawVideoReader` - or CUDA tensors); yields the output frames, uint8 CUDA tensors, in
        order, as soon as their inputs allow.  Identical frames to `run_device` on the whole clip."""
        import torch
        dev = self._device()
        if block < 1:
            raise ValueError("block must be >= 1")
        upload = torch.cuda.Stream(device=dev)

        def gen_up():
            for a in frames:
                if isinstance(a, np.ndarray):
                    # on a stream of its own and waited for on the host (a 6 MB copy): the reader reuses its slot two frames later,
                    # while the compute stream may still be tens of frames behind the host
                    with torch.cuda.stream(upload):
                        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(upload)
                    ev.synchronize()
                    t.record_stream(torch.cuda.current_stream(dev))
                    yield t
                else:
                    yield a.contiguous()

        with torch.cuda.device(dev):
            g = gen_up()
            if self.denoiser is not None:
                g = self._gen_denoise(g, block)
            if self.upscaler is not None:
                g = (self.upscaler.upscale_device(f) for f in g)
            if self.interpolator is not None:
                for _ in range(self.interp_passes):
                    g = self._gen_interp(g)
            yield from g

    def run_stream(self, frames: Iterable, writer, block: int = 8, slots: int = 3) -> int:
        """``frames`` (e.g. a `codec.RawVideoReader` on the decoder's pipe) through the stages into ``writer`` (a
        `codec.RawVideoWriter` on the encoder's pipe): decode, GPU stages, download and encode overlap - the reader thread runs
        ahead of this thread, finished frames leave through a ring of ``slots`` pinned buffers on a download stream, and the writer
        thread waits for each download before it writes.  The writer's bounded queue is the back-pressure.  Returns the number of
        frames written (the caller closes reader and writer)."""
        import queue as _queue

        import torch
        dev = self._device()
        down = torch.cuda.Stream(device=dev)
        pins: Optional[list] = None
        free: "_queue.Queue[int]" = _queue.Queue()
        n = 0
        with torch.cuda.device(dev):
            for t in self.stream_device(frames, block):
                if pins is None:
                    pins = [torch.empty(tuple(t.shape), dtype=torch.uint8).pin_memory() for _ in range(max(2, int(slots)))]
                    for k in range(len(pins)):
                        free.put(k)
                k = free.get()                       # every slot with the writer: wait for it (back-pressure)
                ev_c = torch.cuda.Event()
                ev_c.record(torch.cuda.current_stream(dev))
                with torch.cuda.stream(down):
                    down.wait_event(ev_c)
                    pins[k].copy_(t, non_blocking=True)
                    ev_d = torch.cuda.Event()
                    ev_d.record(down)
                t.record_stream(down)
                writer.write(pins[k].numpy(), ready=ev_d.synchronize, release=lambda k=k: free.put(k))
                n += 1
        return n
