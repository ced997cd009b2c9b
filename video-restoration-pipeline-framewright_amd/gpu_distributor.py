"""Drop-in for the reference's second multi-GPU front end, ``framewright.infrastructure.gpu.distributor``
(``GPUDistributor`` :128-580, ``MultiGPUProcessor`` :583-834, ``detect_multi_gpu_support`` :841+): the B4 boundary
``process_func(frame: ndarray, device_id: int) -> ndarray`` driven by per-GPU thread pools.

The five planners (round-robin, load-balanced, memory-aware, speed-aware, priority) are restated from :287-470 and pinned on
plans produced by the reference's own class (tests/golden/gpu_distributor_reference.json, oracle/gen_golden.py).  Detection
enumerates the MI355X devices through the HIP library (the reference's detector classifies a non-NVIDIA card as CPU,
enhancement/denoising.py:222-277 - a defect listed in SURVEY.md section 8(f)); the per-device compute backend is
``backends.HipRocmBackend`` (``BackendType.ROCM``).  ``Config.enable_multi_gpu / gpu_ids / gpu_load_balance_strategy /
workers_per_gpu`` (config.py:349-352) map onto ``from_config``.
"""
from __future__ import annotations

import gc
import logging
import threading
import time
from concurrent.futures import Future, ThreadPoolExecutor, as_completed
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Callable, Dict, List, Optional

import numpy as np

logger = logging.getLogger(__name__)


class GPUVendor(Enum):
    """detector.py:26-32."""
    NVIDIA = "nvidia"
    AMD = "amd"
    INTEL = "intel"
    APPLE = "apple"
    UNKNOWN = "unknown"


class DistributionStrategy(Enum):
    """distributor.py:41-55."""
    ROUND_ROBIN = "round_robin"
    LOAD_BALANCED = "load_balanced"
    MEMORY_AWARE = "memory_aware"
    SPEED_AWARE = "speed_aware"
    PRIORITY = "priority"


@dataclass
class DeviceInfo:
    """The fields of detector.DeviceInfo (:65-89) the distributor reads."""
    index: int
    name: str
    vendor: GPUVendor
    total_memory_mb: int
    free_memory_mb: int = 0
    driver_version: str = ""
    compute_capability: str = ""
    is_dedicated: bool = True
    supports_fp16: bool = True
    supports_int8: bool = False
    max_threads: int = 0

    @property
    def used_memory_mb(self) -> int:
        return self.total_memory_mb - self.free_memory_mb

    @property
    def memory_usage_percent(self) -> float:
        return 0.0 if self.total_memory_mb == 0 else (self.used_memory_mb / self.total_memory_mb) * 100


@dataclass
class GPUStats:
    """distributor.py:58-89."""
    device_id: int
    vendor: GPUVendor
    name: str
    total_memory_mb: int = 0
    used_memory_mb: int = 0
    frames_processed: int = 0
    total_time_seconds: float = 0.0
    avg_time_per_frame: float = 0.0
    errors: int = 0
    is_healthy: bool = True

    def update_timing(self, elapsed: float):
        self.frames_processed += 1
        self.total_time_seconds += elapsed
        self.avg_time_per_frame = self.total_time_seconds / self.frames_processed


@dataclass
class DistributionPlan:
    """distributor.py:92-105."""
    frame_assignments: Dict[int, int] = field(default_factory=dict)
    gpu_workloads: Dict[int, List[int]] = field(default_factory=dict)
    total_frames: int = 0
    estimated_time_seconds: float = 0.0


@dataclass
class ProcessingResult:
    """distributor.py:108-125."""
    frame_index: int
    device_id: int
    success: bool
    output: Optional[Any] = None
    error: Optional[str] = None
    elapsed_seconds: float = 0.0


def detect_devices() -> List[DeviceInfo]:
    """Every HIP device visible to the library, with torch's memory figures when a context can be queried."""
    from . import _lib
    try:
        n = _lib.load().fw_device_count()
    except _lib.FramewrightHipError:
        return []
    out: List[DeviceInfo] = []
    for i in range(n):
        name, total, free = f"HIP device {i}", 0, 0
        try:
            import torch
            p = torch.cuda.get_device_properties(i)
            name, total = p.name, int(p.total_memory // (1 << 20))
            f, _ = torch.cuda.mem_get_info(i)
            free = int(f // (1 << 20))
        except Exception:  # noqa: BLE001 - detection must not raise (reference: get_hardware_info never does)
            free = total
        out.append(DeviceInfo(index=i, name=name, vendor=GPUVendor.AMD, total_memory_mb=total, free_memory_mb=free))
    return out


def _weighted(frame_count: int, order: List[int], weights: Dict[int, float], plan: DistributionPlan) -> DistributionPlan:
    """The loop the weighted planners share (:349-361): int(frame_count * weight) consecutive frames per device in dict
    order, the last device takes what is left."""
    assigned = 0
    for device_id in order:
        count = int(frame_count * weights[device_id])
        if device_id == order[-1]:
            count = frame_count - assigned
        for _ in range(count):
            if assigned < frame_count:
                plan.frame_assignments[assigned] = device_id
                plan.gpu_workloads[device_id].append(assigned)
                assigned += 1
    return plan


class GPUDistributor:
    """distributor.py:128-580."""

    def __init__(self, strategy: DistributionStrategy = DistributionStrategy.LOAD_BALANCED,
                 excluded_devices: Optional[List[int]] = None):
        self.strategy = strategy
        self.excluded_devices = set(excluded_devices or [])
        self._devices: List[DeviceInfo] = []
        self._stats: Dict[int, GPUStats] = {}
        self._backends: Dict[int, Any] = {}
        self._lock = threading.Lock()
        self._initialized = False

    def detect_all_gpus(self, force_refresh: bool = False) -> List[DeviceInfo]:
        if self._devices and not force_refresh:
            return self._devices
        with self._lock:
            self._devices = [d for d in detect_devices() if d.index not in self.excluded_devices]
            for d in self._devices:
                if d.index not in self._stats:
                    self._stats[d.index] = GPUStats(device_id=d.index, vendor=d.vendor, name=d.name, total_memory_mb=d.total_memory_mb,
                                                    used_memory_mb=d.total_memory_mb - d.free_memory_mb)
            logger.info(f"Detected {len(self._devices)} GPUs: {[d.name for d in self._devices]}")
            return self._devices

    def get_device_count(self) -> int:
        if not self._devices:
            self.detect_all_gpus()
        return len(self._devices)

    def get_device_info(self, device_id: int) -> Optional[DeviceInfo]:
        if not self._devices:
            self.detect_all_gpus()
        for d in self._devices:
            if d.index == device_id:
                return d
        return None

    def get_gpu_stats(self, device_id: int) -> Optional[GPUStats]:
        return self._stats.get(device_id)

    def get_all_stats(self) -> Dict[int, GPUStats]:
        return self._stats.copy()

    def distribute_frames(self, frame_count: int, strategy: Optional[DistributionStrategy] = None) -> DistributionPlan:
        if not self._devices:
            self.detect_all_gpus()
        strategy = strategy or self.strategy
        plan = DistributionPlan(total_frames=frame_count)
        if not self._devices:
            logger.warning("No GPUs available for distribution")
            return plan
        healthy = [d for d in self._devices if self._stats.get(d.index, GPUStats(0, GPUVendor.UNKNOWN, "")).is_healthy]
        if not healthy:
            healthy = self._devices
        fn = {DistributionStrategy.ROUND_ROBIN: self._distribute_round_robin,
              DistributionStrategy.LOAD_BALANCED: self._distribute_load_balanced,
              DistributionStrategy.MEMORY_AWARE: self._distribute_memory_aware,
              DistributionStrategy.SPEED_AWARE: self._distribute_speed_aware,
              DistributionStrategy.PRIORITY: self._distribute_priority}.get(strategy)
        return fn(frame_count, healthy) if fn else plan

    @staticmethod
    def _empty_plan(frame_count: int, devices) -> DistributionPlan:
        plan = DistributionPlan(total_frames=frame_count)
        for d in devices:
            plan.gpu_workloads[d.index] = []
        return plan

    def _distribute_round_robin(self, frame_count: int, devices) -> DistributionPlan:
        """:287-304."""
        plan = self._empty_plan(frame_count, devices)
        for i in range(frame_count):
            dev = devices[i % len(devices)].index
            plan.frame_assignments[i] = dev
            plan.gpu_workloads[dev].append(i)
        return plan

    def _distribute_load_balanced(self, frame_count: int, devices) -> DistributionPlan:
        """:306-363: weight = speed factor 1 / (avg time + 1 ms) x free-memory fraction."""
        plan = self._empty_plan(frame_count, devices)
        scores: Dict[int, float] = {}
        for d in devices:
            st = self._stats.get(d.index)
            scores[d.index] = (1.0 / (st.avg_time_per_frame + 0.001)) * (d.free_memory_mb / max(d.total_memory_mb, 1)) if st else 1.0
        total = sum(scores.values())
        return _weighted(frame_count, list(scores), {k: v / total for k, v in scores.items()}, plan)

    def _distribute_memory_aware(self, frame_count: int, devices) -> DistributionPlan:
        """:365-399."""
        total_free = sum(d.free_memory_mb for d in devices)
        if total_free == 0:
            return self._distribute_round_robin(frame_count, devices)
        plan = self._empty_plan(frame_count, devices)
        weights = {d.index: d.free_memory_mb / total_free for d in devices}
        return _weighted(frame_count, list(weights), weights, plan)

    def _distribute_speed_aware(self, frame_count: int, devices) -> DistributionPlan:
        """:401-443: measured 1 / avg time, else a tier default by VRAM."""
        plan = self._empty_plan(frame_count, devices)
        speeds: Dict[int, float] = {}
        for d in devices:
            st = self._stats.get(d.index)
            if st and st.avg_time_per_frame > 0:
                speeds[d.index] = 1.0 / st.avg_time_per_frame
            else:
                speeds[d.index] = 100.0 if d.total_memory_mb >= 16384 else 60.0 if d.total_memory_mb >= 8192 else 30.0
        total = sum(speeds.values())
        return _weighted(frame_count, list(speeds), {k: v / total for k, v in speeds.items()}, plan)

    def _distribute_priority(self, frame_count: int, devices) -> DistributionPlan:
        """:445-490: devices by VRAM (larger first), 60 % of what remains to each in turn."""
        order = sorted(devices, key=lambda d: d.total_memory_mb, reverse=True)
        weights, remaining = [], 1.0
        for i, _ in enumerate(order):
            if i == len(order) - 1:
                weights.append(remaining)
            else:
                w = remaining * 0.6
                weights.append(w)
                remaining -= w
        plan = self._empty_plan(frame_count, order)
        assigned = 0
        for d, w in zip(order, weights):
            count = int(frame_count * w)
            if d == order[-1]:
                count = frame_count - assigned
            for _ in range(count):
                if assigned < frame_count:
                    plan.frame_assignments[assigned] = d.index
                    plan.gpu_workloads[d.index].append(assigned)
                    assigned += 1
        return plan

    def get_optimal_distribution(self, frame_count: int, frame_memory_mb: float = 100.0) -> DistributionPlan:
        """:472-507."""
        if not self._devices:
            self.detect_all_gpus()
        if any(s.frames_processed > 10 for s in self._stats.values()):
            return self.distribute_frames(frame_count, DistributionStrategy.SPEED_AWARE)
        if any(d.free_memory_mb < frame_memory_mb * 10 for d in self._devices):
            return self.distribute_frames(frame_count, DistributionStrategy.MEMORY_AWARE)
        return self.distribute_frames(frame_count, DistributionStrategy.LOAD_BALANCED)

    def collect_results(self, futures: List[Future], timeout: Optional[float] = None) -> List[ProcessingResult]:
        """:509-548."""
        results = []
        for fut in as_completed(futures, timeout=timeout):
            try:
                r = fut.result()
                results.append(r)
                if r.device_id in self._stats:
                    st = self._stats[r.device_id]
                    if r.success:
                        st.update_timing(r.elapsed_seconds)
                    else:
                        st.errors += 1
                        if st.errors > 5:
                            st.is_healthy = False
            except Exception as e:  # noqa: BLE001
                logger.error(f"Error collecting result: {e}")
                results.append(ProcessingResult(frame_index=-1, device_id=-1, success=False, error=str(e)))
        return results

    def mark_device_unhealthy(self, device_id: int) -> None:
        if device_id in self._stats:
            self._stats[device_id].is_healthy = False
            logger.warning(f"GPU {device_id} marked as unhealthy")

    def mark_device_healthy(self, device_id: int) -> None:
        if device_id in self._stats:
            self._stats[device_id].is_healthy = True
            self._stats[device_id].errors = 0

    def reset_stats(self) -> None:
        for st in self._stats.values():
            st.frames_processed = 0
            st.total_time_seconds = 0.0
            st.avg_time_per_frame = 0.0
            st.errors = 0
            st.is_healthy = True


class MultiGPUProcessor:
    """distributor.py:583-834: one thread pool per GPU, ``process_func(frame, device_id)`` per frame, results in frame order."""

    def __init__(self, strategy: DistributionStrategy = DistributionStrategy.LOAD_BALANCED, max_workers_per_gpu: int = 1,
                 excluded_devices: Optional[List[int]] = None):
        self.distributor = GPUDistributor(strategy, excluded_devices)
        self.max_workers_per_gpu = max_workers_per_gpu
        self._backends: Dict[int, Any] = {}
        self._executors: Dict[int, ThreadPoolExecutor] = {}
        self._initialized = False
        self._lock = threading.Lock()

    @classmethod
    def from_config(cls, config) -> "MultiGPUProcessor":
        """The four multi-GPU fields of the reference's Config (config.py:349-352): ``gpu_ids`` selects the devices (every other
        index is excluded), ``gpu_load_balance_strategy`` names a DistributionStrategy (the names of utils/multi_gpu's
        LoadBalanceStrategy map onto it), ``workers_per_gpu`` sizes the per-GPU pools."""
        names = {"round_robin": DistributionStrategy.ROUND_ROBIN, "least_loaded": DistributionStrategy.LOAD_BALANCED,
                 "load_balanced": DistributionStrategy.LOAD_BALANCED, "vram_aware": DistributionStrategy.MEMORY_AWARE,
                 "memory_aware": DistributionStrategy.MEMORY_AWARE, "weighted": DistributionStrategy.SPEED_AWARE,
                 "speed_aware": DistributionStrategy.SPEED_AWARE, "priority": DistributionStrategy.PRIORITY}
        strat = names.get(str(getattr(config, "gpu_load_balance_strategy", "load_balanced")).lower(), DistributionStrategy.LOAD_BALANCED)
        ids = getattr(config, "gpu_ids", None)
        excluded = None
        if ids:
            excluded = [d.index for d in detect_devices() if d.index not in set(ids)]
        return cls(strat, int(getattr(config, "workers_per_gpu", 1) or 1), excluded)

    def initialize(self) -> bool:
        if self._initialized:
            return True
        with self._lock:
            devices = self.distributor.detect_all_gpus()
            if not devices:
                logger.warning("No GPUs detected")
                return False
            from . import backends as B
            for d in devices:
                try:
                    backend = B.HipRocmBackend(device_id=d.index)
                    if backend.initialize():
                        self._backends[d.index] = backend
                        self._executors[d.index] = ThreadPoolExecutor(max_workers=self.max_workers_per_gpu,
                                                                      thread_name_prefix=f"gpu_{d.index}_")
                        logger.info(f"Initialized GPU {d.index}: {d.name}")
                    else:
                        logger.warning(f"Failed to initialize GPU {d.index}")
                except Exception as e:  # noqa: BLE001 - a device that fails to come up is skipped (:651-652)
                    logger.error(f"Error initializing GPU {d.index}: {e}")
            self._initialized = len(self._backends) > 0
            return self._initialized

    def cleanup(self) -> None:
        with self._lock:
            for ex in self._executors.values():
                ex.shutdown(wait=True)
            self._executors.clear()
            for b in self._backends.values():
                b.cleanup()
            self._backends.clear()
            self._initialized = False
            gc.collect()

    def process_frames(self, frames: List[np.ndarray], process_func: Callable[[np.ndarray, int], np.ndarray],
                       callback: Optional[Callable[[ProcessingResult], None]] = None,
                       timeout_per_frame: float = 60.0) -> List[ProcessingResult]:
        """:687-752."""
        if not self._initialized and not self.initialize():
            raise RuntimeError("Failed to initialize multi-GPU processor")
        plan = self.distributor.get_optimal_distribution(len(frames))
        futures = []
        for idx, dev in plan.frame_assignments.items():
            if dev not in self._executors:
                continue
            futures.append(self._executors[dev].submit(self._process_single_frame, frames[idx], idx, dev, process_func))
        results: List[ProcessingResult] = []
        for fut in as_completed(futures, timeout=timeout_per_frame * max(len(frames), 1)):
            try:
                r = fut.result(timeout=timeout_per_frame)
                results.append(r)
                st = self.distributor._stats.get(r.device_id)
                if st is not None:
                    if r.success:
                        st.update_timing(r.elapsed_seconds)
                    else:
                        st.errors += 1
                if callback:
                    callback(r)
            except Exception as e:  # noqa: BLE001
                logger.error(f"Frame processing error: {e}")
        results.sort(key=lambda r: r.frame_index)
        return results

    def _process_single_frame(self, frame: np.ndarray, frame_idx: int, device_id: int, process_func: Callable) -> ProcessingResult:
        t0 = time.time()
        try:
            out = process_func(frame, device_id)
            return ProcessingResult(frame_index=frame_idx, device_id=device_id, success=True, output=out, elapsed_seconds=time.time() - t0)
        except Exception as e:  # noqa: BLE001 - reported per frame, never raised (:766-777)
            logger.error(f"Error processing frame {frame_idx} on GPU {device_id}: {e}")
            return ProcessingResult(frame_index=frame_idx, device_id=device_id, success=False, error=str(e), elapsed_seconds=time.time() - t0)

    def process_batch(self, batch: np.ndarray, process_func: Callable[[np.ndarray, int], np.ndarray],
                      device_id: Optional[int] = None) -> np.ndarray:
        """:779-819: the device with the most free memory when none is named."""
        if not self._initialized and not self.initialize():
            raise RuntimeError("Failed to initialize")
        if device_id is None:
            best, best_mem = None, 0
            for d, b in self._backends.items():
                free = b.get_memory_info()["free_mb"]
                if free > best_mem:
                    best, best_mem = d, free
            device_id = best or list(self._backends.keys())[0]
        return process_func(batch, device_id)

    def get_available_gpus(self) -> List[int]:
        return list(self._backends.keys())

    def get_gpu_count(self) -> int:
        return len(self._backends)

    def get_backend(self, device_id: int):
        return self._backends.get(device_id)

    def get_stats(self) -> Dict[int, GPUStats]:
        return self.distributor.get_all_stats()


def detect_multi_gpu_support() -> Dict[str, Any]:
    """distributor.py:841-875."""
    devices = GPUDistributor().detect_all_gpus()
    vendors = list(set(d.vendor for d in devices))
    return {"gpu_count": len(devices), "vendors": [v.value for v in vendors], "total_memory_mb": sum(d.total_memory_mb for d in devices),
            "mixed_vendors": len(vendors) > 1,
            "gpus": [{"device_id": d.index, "name": d.name, "vendor": d.vendor.value, "memory_mb": d.total_memory_mb} for d in devices]}


def upscale_process_func(config=None) -> Callable[[np.ndarray, int], np.ndarray]:
    """A B4 ``process_func(frame, device_id)`` for ``MultiGPUProcessor.process_frames``: Real-ESRGAN on the named device (one
    cached upsampler per device through ``get_upsampler``)."""
    import dataclasses
    from . import realesrgan as R
    base = config or R.PyTorchESRGANConfig()

    def fn(frame: np.ndarray, device_id: int) -> np.ndarray:
        up = R.get_upsampler(dataclasses.replace(base, gpu_id=int(device_id)))
        return up.enhance(frame, outscale=base.scale_factor)[0]
    return fn
