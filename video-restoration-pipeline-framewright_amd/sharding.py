"""Frame sharding over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* Real-ESRGAN: frames are independent -> round-robin ``frame_idx % world`` exactly like the reference's planners
  (``infrastructure/gpu/distributor.py:287-304``, ``utils/multi_gpu.py:796-800``); NO data-path collective.
* TAP temporal denoise (window 2r+1): contiguous block partition; each rank denoises its own frames once, then ONE
  exchange step sends the first/last r DENOISED uint8 frames to the neighbouring ranks (point-to-point
  isend/irecv — ncclSend/ncclRecv over the direct xGMI link; there is no ring collective anywhere), then the local
  weighted average.  Clip ends clamp as ``tap_denoise.py:510-511``.
* RIFE x2 (pairs i, i+1): block partition + a 1-frame halo of INPUT frames from rank r+1.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np


def round_robin_assignment(n_frames: int, world: int) -> Dict[int, List[int]]:
    """rank -> frame indices; ``distributor.py:298-302``."""
    if world < 1:
        raise ValueError("world must be >= 1")
    out: Dict[int, List[int]] = {r: [] for r in range(world)}
    for i in range(n_frames):
        out[i % world].append(i)
    return out


def block_partition(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """[(start, end)) per rank: rank r owns frames [r*N/world, (r+1)*N/world)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    return [(r * n_frames // world, (r + 1) * n_frames // world) for r in range(world)]


def _exchange(send_prev: Optional[np.ndarray], send_next: Optional[np.ndarray], shape_prev, shape_next, device):
    """One neighbour exchange step: returns (from_prev, from_next) as uint8 arrays (or None at the clip ends)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    ops, bufs = [], {}

    def to_t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    if rank > 0:
        if send_prev is not None and send_prev.size:
            ops.append(dist.P2POp(dist.isend, to_t(send_prev), rank - 1))
        if shape_prev is not None and int(np.prod(shape_prev)):
            bufs["prev"] = torch.empty(shape_prev, dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs["prev"], rank - 1))
    if rank < world - 1:
        if send_next is not None and send_next.size:
            ops.append(dist.P2POp(dist.isend, to_t(send_next), rank + 1))
        if shape_next is not None and int(np.prod(shape_next)):
            bufs["next"] = torch.empty(shape_next, dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, bufs["next"], rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    g = lambda k: bufs[k].cpu().numpy() if k in bufs else None
    return g("prev"), g("next")


def sharded_upscale(frames: Sequence[np.ndarray], upscale_fn: Callable[[np.ndarray], np.ndarray]) -> Dict[int, np.ndarray]:
    """This rank's share of a clip through ``upscale_fn`` (round-robin, no communication).  Returns {frame_idx: output}."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    return {i: upscale_fn(frames[i]) for i in round_robin_assignment(len(frames), world)[rank]}


def sharded_temporal_denoise(frames: Sequence[np.ndarray], radius: int,
                             denoise_one: Callable[[np.ndarray], np.ndarray],
                             combine: Callable[[List[np.ndarray], int, int, int], np.ndarray],
                             device="cpu") -> Dict[int, np.ndarray]:
    """Block-partitioned temporal stage with a ``radius``-frame halo of already-denoised neighbours.

    ``denoise_one(frame) -> uint8 frame``; ``combine(window_frames, window_start, center, n_total) -> uint8 frame`` gets the
    denoised frames of the clamped window [window_start, window_start + len) and the global centre index.
    Every rank passes the FULL list of input frames (or at least its own block; other entries may be None).
    """
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = len(frames)
    parts = block_partition(n, world)
    lo, hi = parts[rank]
    own = [denoise_one(frames[i]) for i in range(lo, hi)]
    prev_halo = next_halo = None
    if world > 1:
        shape = tuple(own[0].shape) if own else tuple(np.asarray(frames[0]).shape)
        cnt = lambda r: max(0, parts[r][1] - parts[r][0])
        # what the neighbours own limits what they can send (blocks shorter than the radius are not supported)
        for r in range(world):
            if 0 < cnt(r) < radius and world > 1:
                raise ValueError("block partition needs at least `radius` frames per rank")
        k_prev = min(radius, cnt(rank - 1)) if rank > 0 else 0
        k_next = min(radius, cnt(rank + 1)) if rank < world - 1 else 0
        send_prev = np.stack(own[:radius]) if own and rank > 0 else None
        send_next = np.stack(own[-radius:]) if own and rank < world - 1 else None
        prev_halo, next_halo = _exchange(send_prev, send_next, (k_prev,) + shape if k_prev else None,
                                         (k_next,) + shape if k_next else None, device)
    den: Dict[int, np.ndarray] = {lo + j: f for j, f in enumerate(own)}
    if prev_halo is not None:
        for j in range(prev_halo.shape[0]):
            den[lo - prev_halo.shape[0] + j] = prev_halo[j]
    if next_halo is not None:
        for j in range(next_halo.shape[0]):
            den[hi + j] = next_halo[j]
    out: Dict[int, np.ndarray] = {}
    for i in range(lo, hi):
        s, e = max(0, i - radius), min(n, i + radius + 1)
        out[i] = combine([den[j] for j in range(s, e)], s, i, n)
    return out


def sharded_pairs(frames: Sequence[np.ndarray], interp_pair: Callable[[np.ndarray, np.ndarray], np.ndarray],
                  device="cpu") -> Dict[int, np.ndarray]:
    """RIFE x2: mid-frames of the pairs (i, i+1) owned by this rank; the right neighbour's first INPUT frame is the halo."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = len(frames)
    lo, hi = block_partition(n, world)[rank]
    halo = None
    if world > 1:
        shape = tuple(np.asarray(frames[lo if hi > lo else 0]).shape)
        send_prev = np.stack([frames[lo]]) if hi > lo and rank > 0 else None
        _, nxt = _exchange(send_prev, None, None, (1,) + shape if rank < world - 1 and hi < n else None, device)
        halo = nxt[0] if nxt is not None else None
    out: Dict[int, np.ndarray] = {}
    for i in range(lo, hi):
        if i + 1 < hi:
            out[i] = interp_pair(frames[i], frames[i + 1])
        elif i + 1 < n:
            out[i] = interp_pair(frames[i], halo if halo is not None else frames[i + 1])
    return out
