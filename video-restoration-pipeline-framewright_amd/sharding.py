"""Frame sharding over the GPUs of one node: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* Real-ESRGAN: frames are independent -> round-robin ``frame_idx % world`` exactly like the reference's planners
  (``infrastructure/gpu/distributor.py:287-304``, ``utils/multi_gpu.py:796-800``); NO data-path collective.
* TAP temporal denoise (window 2r+1): contiguous block partition; each rank denoises its own frames once, then ONE
  exchange step sends the first/last r DENOISED uint8 frames to the neighbouring ranks (point-to-point
  isend/irecv — ncclSend/ncclRecv over the direct xGMI link; there is no ring collective anywhere), then the local
  weighted average.  Clip ends clamp as ``tap_denoise.py:510-511``.
* RIFE x2 (pairs i, i+1): block partition + a 1-frame halo of INPUT frames from rank r+1.

Halos travel as uint8 tensors on the device the frames live on (CUDA tensors over RCCL; CPU tensors over gloo in the
tests): nothing is staged through host numpy.  Every rank derives who sends what from the SAME partition table, so a clip
shorter than the world (ranks without frames) cannot leave a receive posted that nobody answers.
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


def block_partition(n_frames: int, world: int, min_block: int = 1) -> List[Tuple[int, int]]:
    """[(start, end)) per rank.  The first ``active = min(world, n_frames // min_block)`` ranks (at least one) share the
    clip in contiguous blocks of at least ``min_block`` frames, rank r of them owning [r*N/active, (r+1)*N/active); the
    remaining ranks own nothing.  With N >= world * min_block this is the plain [r*N/world, (r+1)*N/world) split."""
    if world < 1 or min_block < 1:
        raise ValueError("world and min_block must be >= 1")
    active = max(1, min(world, n_frames // min_block)) if n_frames > 0 else 0
    parts = [(r * n_frames // active, (r + 1) * n_frames // active) for r in range(active)]
    return parts + [(n_frames, n_frames)] * (world - active)


def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def exchange_with_neighbours(send_prev, send_next, recv_prev_shape, recv_next_shape, device):
    """One neighbour exchange step on uint8 tensors: sends ``send_prev`` to rank - 1 and ``send_next`` to rank + 1 (None =
    nothing), receives tensors of the given shapes (None = nothing) from them.  Returns (from_prev, from_next).  The caller
    computes the four arguments from the partition table, identically on both ends of every link."""
    import torch
    import torch.distributed as dist
    rank, world = _dist_info()
    ops, got = [], {}
    if send_prev is not None:
        ops.append(dist.P2POp(dist.isend, send_prev.contiguous(), rank - 1))
    if recv_prev_shape is not None:
        got["prev"] = torch.empty(tuple(recv_prev_shape), dtype=torch.uint8, device=device)
        ops.append(dist.P2POp(dist.irecv, got["prev"], rank - 1))
    if send_next is not None:
        ops.append(dist.P2POp(dist.isend, send_next.contiguous(), rank + 1))
    if recv_next_shape is not None:
        got["next"] = torch.empty(tuple(recv_next_shape), dtype=torch.uint8, device=device)
        ops.append(dist.P2POp(dist.irecv, got["next"], rank + 1))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return got.get("prev"), got.get("next")


def _as_tensor(a, device):
    import torch
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return a.to(device)


def sharded_upscale(frames: Sequence[np.ndarray], upscale_fn: Callable[[np.ndarray], np.ndarray]) -> Dict[int, np.ndarray]:
    """This rank's share of a clip through ``upscale_fn`` (round-robin, no communication).  Returns {frame_idx: output}."""
    rank, world = _dist_info()
    return {i: upscale_fn(frames[i]) for i in round_robin_assignment(len(frames), world)[rank]}


def temporal_halo_plan(n_frames: int, world: int, rank: int, radius: int):
    """(own block, frames to send to / receive from rank - 1, frames to send to / receive from rank + 1) for the temporal
    stage: a rank sends its first ``radius`` frames to its predecessor and its last ``radius`` to its successor, if that
    neighbour owns frames at all.  Blocks hold at least ``radius`` frames (block_partition's ``min_block``)."""
    parts = block_partition(n_frames, world, max(1, radius))
    cnt = [e - s for s, e in parts]
    lo, hi = parts[rank]
    has = lambda r: 0 <= r < world and cnt[r] > 0
    mine = cnt[rank] > 0
    k = lambda r: min(radius, cnt[r])
    send_prev = k(rank) if mine and has(rank - 1) else 0
    recv_prev = k(rank - 1) if mine and has(rank - 1) else 0
    send_next = k(rank) if mine and has(rank + 1) else 0
    recv_next = k(rank + 1) if mine and has(rank + 1) else 0
    return (lo, hi), (send_prev, recv_prev), (send_next, recv_next)


def sharded_temporal_denoise(frames: Sequence, radius: int, denoise_one: Callable, combine: Callable, device="cpu") -> Dict[int, object]:
    """Block-partitioned temporal stage with a ``radius``-frame halo of already-denoised neighbours.

    ``denoise_one(frame) -> uint8 frame``; ``combine(window_frames, window_start, center, n_total) -> uint8 frame`` gets the
    denoised frames of the clamped window [window_start, window_start + len) and the global centre index.  Frames may be
    numpy arrays or tensors; the halo frames handed to ``combine`` are tensors on ``device`` when the inputs are tensors and
    numpy arrays when they are numpy.  Every rank passes the FULL list of input frames (or at least its own block; other
    entries may be None)."""
    import torch
    rank, world = _dist_info()
    n = len(frames)
    (lo, hi), (sp, rp), (sn, rn) = temporal_halo_plan(n, world, rank, radius)
    own = [denoise_one(frames[i]) for i in range(lo, hi)]
    as_numpy = bool(own) and isinstance(own[0], np.ndarray)
    den: Dict[int, object] = {lo + j: f for j, f in enumerate(own)}
    if world > 1 and own:
        shape = tuple(own[0].shape)
        t = [_as_tensor(f, device) for f in own]
        prev_halo, next_halo = exchange_with_neighbours(
            torch.stack(t[:sp]) if sp else None, torch.stack(t[-sn:]) if sn else None,
            (rp,) + shape if rp else None, (rn,) + shape if rn else None, device)
        conv = (lambda x: x.cpu().numpy()) if as_numpy else (lambda x: x)
        for j in range(rp):
            den[lo - rp + j] = conv(prev_halo[j])
        for j in range(rn):
            den[hi + j] = conv(next_halo[j])
    out: Dict[int, object] = {}
    for i in range(lo, hi):
        s, e = max(0, i - radius), min(n, i + radius + 1)
        out[i] = combine([den[j] for j in range(s, e)], s, i, n)
    return out


def sharded_pairs(frames: Sequence, interp_pair: Callable, device="cpu", interp_many: Optional[Callable] = None) -> Dict[int, object]:
    """RIFE x2: mid-frames of the pairs (i, i+1) whose left frame this rank owns; the right neighbour's first INPUT frame is
    the halo (a tensor on ``device`` when the inputs are tensors).  ``interp_many([(a, b), ...]) -> [mid, ...]``, when given, gets
    all of this rank's pairs in one call (IFNetEngine.interpolate_pairs_device overlaps them on streams)."""
    import torch
    rank, world = _dist_info()
    n = len(frames)
    parts = block_partition(n, world)
    cnt = [e - s for s, e in parts]
    lo, hi = parts[rank]
    mine = cnt[rank] > 0
    halo = None
    if world > 1 and mine:
        f0 = frames[lo]
        as_numpy = isinstance(f0, np.ndarray)
        send_prev = _as_tensor(f0, device).unsqueeze(0) if rank > 0 and cnt[rank - 1] > 0 else None
        want_next = rank + 1 < world and cnt[rank + 1] > 0
        _, nxt = exchange_with_neighbours(send_prev, None, None, (1,) + tuple(f0.shape) if want_next else None, device)
        if nxt is not None:
            halo = nxt[0].cpu().numpy() if as_numpy else nxt[0]
    todo = []
    for i in range(lo, hi):
        if i + 1 < hi:
            todo.append((i, frames[i], frames[i + 1]))
        elif i + 1 < n:
            todo.append((i, frames[i], halo if halo is not None else frames[i + 1]))
    if interp_many is not None:
        return dict(zip([i for i, _, _ in todo], interp_many([(a, b) for _, a, b in todo])))
    return {i: interp_pair(a, b) for i, a, b in todo}


def sharded_tap_denoise_device(tap, frames: Sequence, halo_device=None) -> Dict[int, object]:
    """BASELINE configs[3] on the real engine, device-resident: this rank's block of ``frames`` (numpy arrays or uint8 CUDA
    tensors; entries outside the block may be None) through ``tap`` (a TAPDenoiser).  Every frame is denoised ONCE by the rank
    that owns it; the first / last ``temporal_window // 2`` denoised frames go to the neighbours as CUDA tensors (ncclSend /
    ncclRecv); the weighted window average, strength blend and grain add-back then run locally with those halos
    (``denoise_clip_device(halo_before=, halo_after=, denoised=)``).  Returns {frame_idx: uint8 CUDA tensor}; bit-identical to
    the single-process clip."""
    import torch
    rank, world = _dist_info()
    n = len(frames)
    radius = tap.config.temporal_window // 2
    dev = torch.device("cuda", tap.config.gpu_id)
    (lo, hi), (sp, rp), (sn, rn) = temporal_halo_plan(n, world, rank, radius)
    block = [_as_tensor(frames[i], dev) for i in range(lo, hi)]
    if not block:
        return {}
    den = tap.denoise_only_device(block)
    prev_halo = next_halo = None
    if world > 1 and radius > 0:
        shape = tuple(den[0].shape)
        hd = dev if halo_device is None else torch.device(halo_device)
        prev_halo, next_halo = exchange_with_neighbours(
            torch.stack(den[:sp]).to(hd) if sp else None, torch.stack(den[-sn:]).to(hd) if sn else None,
            (rp,) + shape if rp else None, (rn,) + shape if rn else None, hd)
        prev_halo = prev_halo.to(dev) if prev_halo is not None else None
        next_halo = next_halo.to(dev) if next_halo is not None else None
    out = tap.denoise_clip_device(block, halo_before=[prev_halo[j] for j in range(rp)] if prev_halo is not None else (),
                                  halo_after=[next_halo[j] for j in range(rn)] if next_halo is not None else (), denoised=den)
    return {lo + j: o for j, o in enumerate(out)}


def sharded_interpolate_device(engine, frames: Sequence, halo_device=None) -> Dict[int, object]:
    """BASELINE configs[2] sharded: mid-frames of this rank's pairs on an IFNetEngine, frames and halo as uint8 CUDA tensors
    (``halo_device="cpu"``: the halo crosses a gloo process group as a CPU tensor)."""
    import torch
    dev = torch.device("cuda", engine.device_id)
    rank, world = _dist_info()
    lo, hi = block_partition(len(frames), world)[rank]
    fr = [_as_tensor(f, dev) if (lo <= i < hi or (world == 1 and f is not None)) and f is not None else None for i, f in enumerate(frames)]
    hd = dev if halo_device is None else torch.device(halo_device)
    return sharded_pairs(fr, lambda a, b: engine.interpolate_device(a.to(dev), b.to(dev), 0.5), device=hd,
                         interp_many=lambda ps: engine.interpolate_pairs_device([(a.to(dev), b.to(dev)) for a, b in ps], 0.5))
