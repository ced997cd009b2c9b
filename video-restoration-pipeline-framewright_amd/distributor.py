"""Frame distribution across the GPUs of one node behind the reference's ``MultiGPUDistributor`` interface
(``src/framewright/utils/multi_gpu.py:511-870``).

Same surface: ``MultiGPUDistributor(gpu_manager, strategy, workers_per_gpu, max_retries, enable_work_stealing)
.distribute_frames(frames, process_fn, output_dir, progress_callback) -> DistributionResult`` with
``process_fn(input_path, output_dir, gpu_id) -> (output_path, ok, error)`` (``backends.make_shard_process_fn`` is one), the
four ``LoadBalanceStrategy`` assignment rules (``_assign_frames``, :780-870, restated in ``assign_frames`` and pinned on
reference-run vectors: tests/golden/assign_frames.json), retry of a failed frame on a GPU it has not failed on yet
(:676-700), ``DistributionResult`` with the reference's derived fields (:96-138, :745-757).

Not a translation of its machinery: the reference spins ``num_gpus * workers_per_gpu`` threads over per-worker queues with
work stealing; here every GPU has ONE deque served by ``workers_per_gpu`` threads (the HIP engines serialise per GPU, so
the extra workers only overlap PNG decode / encode with the kernels) and a frame that fails moves to the deque of the next
GPU it has not tried.  GPUs come from ``torch.cuda`` (the reference shells out to nvidia-smi, :212-271, which finds no
MI355X).
"""
from __future__ import annotations

import threading
import time
from collections import deque
from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path
from typing import Callable, Deque, Dict, List, Optional, Sequence, Tuple


class LoadBalanceStrategy(Enum):  # multi_gpu.py:30-36
    ROUND_ROBIN = "round_robin"
    LEAST_LOADED = "least_loaded"
    VRAM_AWARE = "vram_aware"
    WEIGHTED = "weighted"


@dataclass
class GPUInfo:  # multi_gpu.py:40-93
    id: int
    name: str
    total_vram_mb: int
    free_vram_mb: int
    utilization_pct: float
    temperature_c: Optional[float] = None
    pcie_bandwidth_gbps: Optional[float] = None
    compute_capability: Optional[str] = None

    @property
    def used_vram_mb(self) -> int:
        return self.total_vram_mb - self.free_vram_mb

    @property
    def vram_usage_pct(self) -> float:
        return 0.0 if self.total_vram_mb == 0 else (self.used_vram_mb / self.total_vram_mb) * 100

    @property
    def is_healthy(self) -> bool:
        return not (self.temperature_c is not None and self.temperature_c > 90)

    @property
    def effective_capacity(self) -> float:
        vram_score = self.free_vram_mb / max(self.total_vram_mb, 1)
        util_score = 1.0 - (self.utilization_pct / 100.0)
        return (vram_score * 0.7) + (util_score * 0.3)


@dataclass
class DistributionResult:  # multi_gpu.py:96-138
    frames_per_gpu: Dict[int, List[Path]] = field(default_factory=dict)
    total_time: float = 0.0
    speedup_factor: float = 1.0
    gpu_utilization: Dict[int, float] = field(default_factory=dict)
    errors: Dict[str, str] = field(default_factory=dict)
    retried_frames: List[Path] = field(default_factory=list)

    @property
    def total_frames(self) -> int:
        return sum(len(v) for v in self.frames_per_gpu.values())

    @property
    def success_rate(self) -> float:
        total = self.total_frames + len(self.errors)
        return 100.0 if total == 0 else (self.total_frames / total) * 100

    def summary(self) -> str:
        counts = ", ".join(f"GPU{g}: {len(v)}" for g, v in self.frames_per_gpu.items())
        return (f"Processed {self.total_frames} frames across {len(self.frames_per_gpu)} GPUs ({counts}) in "
                f"{self.total_time:.1f}s (speedup: {self.speedup_factor:.2f}x, success: {self.success_rate:.1f}%)")


def assign_frames(frames: Sequence, gpus: Sequence[GPUInfo], strategy: LoadBalanceStrategy) -> Dict[int, list]:
    """``MultiGPUDistributor._assign_frames`` (multi_gpu.py:780-870): GPU id -> frames, in frame order."""
    out: Dict[int, list] = {g.id: [] for g in gpus}
    n = len(frames)

    def round_robin(order: Sequence[GPUInfo], start: int = 0) -> None:
        for i in range(start, n):
            out[order[i % len(order)].id].append(frames[i])

    if strategy == LoadBalanceStrategy.ROUND_ROBIN:
        round_robin(gpus)
    elif strategy == LoadBalanceStrategy.LEAST_LOADED:
        round_robin(sorted(gpus, key=lambda g: g.utilization_pct))
    else:
        if strategy == LoadBalanceStrategy.VRAM_AWARE:
            weights = {g.id: g.free_vram_mb for g in gpus}
        elif strategy == LoadBalanceStrategy.WEIGHTED:
            weights = {g.id: g.effective_capacity for g in gpus}
        else:
            raise ValueError(strategy)
        total = sum(weights.values())
        if total == 0:
            round_robin(gpus)
            return out
        idx = 0
        for gid, w in weights.items():            # contiguous runs, int() truncation of the share
            for _ in range(int(n * (w / total))):
                if idx < n:
                    out[gid].append(frames[idx])
                    idx += 1
        if strategy == LoadBalanceStrategy.VRAM_AWARE:
            round_robin(gpus, idx)                 # the rest: gpus[frame_idx % len(gpus)]
        else:
            best = max(weights, key=lambda k: weights[k])   # the rest: the GPU with the highest capacity
            out[best].extend(frames[idx:])
    return out


class GPUManager:
    """The part of the reference's ``GPUManager`` (multi_gpu.py:166-427) the distributor uses, from ``torch.cuda``."""

    def __init__(self, gpu_ids: Optional[Sequence[int]] = None):
        self._ids = list(gpu_ids) if gpu_ids is not None else None

    def detect_gpus(self) -> List[GPUInfo]:
        import torch
        infos = []
        count = torch.cuda.device_count() if torch.cuda.is_available() else 0
        for i in (self._ids if self._ids is not None else range(count)):
            if i >= count:
                continue
            free, total = torch.cuda.mem_get_info(i)
            infos.append(GPUInfo(id=i, name=torch.cuda.get_device_name(i), total_vram_mb=total >> 20, free_vram_mb=free >> 20,
                                 utilization_pct=0.0))
        return infos

    get_all_gpu_info = detect_gpus

    def get_healthy_gpus(self) -> List[GPUInfo]:
        return [g for g in self.detect_gpus() if g.is_healthy]

    @property
    def gpu_ids(self) -> List[int]:
        return [g.id for g in self.detect_gpus()]

    @property
    def gpu_count(self) -> int:
        return len(self.gpu_ids)

    @property
    def is_multi_gpu(self) -> bool:
        return self.gpu_count > 1


ProcessFn = Callable[[Path, Path, int], Tuple[Path, bool, Optional[str]]]


class MultiGPUDistributor:
    def __init__(self, gpu_manager: Optional[GPUManager] = None, strategy: LoadBalanceStrategy = LoadBalanceStrategy.VRAM_AWARE,
                 workers_per_gpu: int = 2, max_retries: int = 2, enable_work_stealing: bool = True):
        self.gpu_manager = gpu_manager or GPUManager()
        self.strategy, self.workers_per_gpu, self.max_retries = strategy, int(workers_per_gpu), int(max_retries)
        self.enable_work_stealing = enable_work_stealing   # accepted for compatibility; a failed frame always moves on
        self._result: Optional[DistributionResult] = None
        self._stop_event = threading.Event()

    def _assign_frames(self, frames, gpus):
        return assign_frames(frames, gpus, self.strategy)

    def distribute_frames(self, frames: List[Path], process_fn: ProcessFn, output_dir: Path,
                          progress_callback: Optional[Callable[[float, str], None]] = None) -> DistributionResult:
        if not frames:
            return DistributionResult()
        t0 = time.time()
        self._stop_event.clear()
        gpus = self.gpu_manager.get_healthy_gpus()
        if not gpus:
            return DistributionResult(errors={str(f): "No GPUs available" for f in frames})
        gpu_ids = [g.id for g in gpus]
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        res = DistributionResult(frames_per_gpu={g: [] for g in gpu_ids})
        queues: Dict[int, Deque] = {g: deque() for g in gpu_ids}
        for gid, fs in self._assign_frames(list(frames), gpus).items():
            for f in fs:
                queues[gid].append((f, []))          # (frame, GPUs it has failed on)
        lock = threading.Condition()
        state = {"done": 0}
        total = len(frames)

        def finish(msg: str) -> None:                # under lock
            state["done"] += 1
            if progress_callback:
                progress_callback(state["done"] / total, msg)
            lock.notify_all()

        def worker(gid: int) -> None:
            while not self._stop_event.is_set():
                with lock:
                    while not queues[gid] and state["done"] < total and not self._stop_event.is_set():
                        lock.wait(0.2)
                    if not queues[gid]:
                        return
                    frame, failed = queues[gid].popleft()
                try:
                    out_path, ok, err = process_fn(frame, output_dir, gid)
                except Exception as e:  # noqa: BLE001 - multi_gpu.py:720-728: an exception fails the frame, no retry
                    with lock:
                        res.errors[str(frame)] = str(e)
                        finish(f"Error: {Path(frame).name}")
                    continue
                with lock:
                    if ok:
                        res.frames_per_gpu[gid].append(out_path)
                        finish(f"Processed {state['done'] + 1}/{total} frames")
                        continue
                    failed = failed + [gid]
                    untried = [g for g in gpu_ids if g not in failed]
                    # WorkItem.can_retry (multi_gpu.py:160-163) is `attempts < 3`, whatever max_retries says - kept
                    if len(failed) < 3 and untried:
                        res.retried_frames.append(frame)
                        queues[untried[0]].append((frame, failed))
                        lock.notify_all()
                    else:
                        res.errors[str(frame)] = err or "Unknown error"
                        finish(f"Error: {Path(frame).name}")

        threads = [threading.Thread(target=worker, args=(g,), daemon=True) for g in gpu_ids for _ in range(max(1, self.workers_per_gpu))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        res.total_time = time.time() - t0
        if len(gpu_ids) > 1 and res.total_frames > 0:                    # multi_gpu.py:745-753
            most = max(len(v) for v in res.frames_per_gpu.values())
            if most > 0:
                res.speedup_factor = len(gpu_ids) * ((res.total_frames / len(gpu_ids)) / most)
        for g in gpus:
            res.gpu_utilization[g.id] = g.utilization_pct
        self._result = res
        return res

    def stop(self) -> None:
        self._stop_event.set()

    def get_result(self) -> Optional[DistributionResult]:
        return self._result
