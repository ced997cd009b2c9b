"""Processor-facade boundary: MI355X backends behind the reference's ABCs.

The reference's ``SRBackend`` (``src/framewright/processors/enhancement/super_resolution.py:237-311``) and
``DenoiserBackend`` (``processors/enhancement/denoising.py:349-386``) cannot be imported on a machine without the
reference, so their interfaces are mirrored here method for method (same names, arguments and result fields);
``register_with_reference()`` plugs the HIP backends into the reference's ``SuperResolution.BACKENDS`` /
``Denoiser.BACKENDS`` tables when the reference package is importable (INTEGRATION.md shows the two lines).
"""
from __future__ import annotations

import time
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, List, Optional, Tuple

import numpy as np

from . import realesrgan as R
from . import tap_denoise as T


# ---- result / ABC mirrors -----------------------------------------------------------------------------------------
@dataclass
class SRResult:
    """super_resolution.py:207-230."""
    frames_processed: int = 0
    frames_failed: int = 0
    output_dir: Optional[Path] = None
    backend_used: str = "unknown"
    processing_time_seconds: float = 0.0
    avg_fps: float = 0.0
    peak_vram_mb: int = 0
    scale_factor: int = 4
    warnings: List[str] = field(default_factory=list)


@dataclass
class DenoiseResult:
    """denoising.py:321-343."""
    frames_processed: int = 0
    frames_failed: int = 0
    output_dir: Optional[Path] = None
    backend_used: str = "unknown"
    processing_time_seconds: float = 0.0
    avg_noise_reduction: float = 0.0
    peak_vram_mb: int = 0
    warnings: List[str] = field(default_factory=list)


class SRBackend(ABC):
    """super_resolution.py:237-311."""

    @property
    @abstractmethod
    def name(self) -> str: ...

    @property
    @abstractmethod
    def supported_scales(self) -> List[int]: ...

    @abstractmethod
    def is_available(self) -> bool: ...

    @abstractmethod
    def estimate_vram_usage(self, width: int, height: int, scale: int) -> int: ...

    @abstractmethod
    def upscale_frame(self, frame: np.ndarray, scale: int = 4) -> np.ndarray: ...

    @abstractmethod
    def upscale_frames(self, input_dir: Path, output_dir: Path, scale: int = 4,
                       progress_callback: Optional[Callable[[float], None]] = None) -> SRResult: ...

    def clear_cache(self) -> None:
        pass


class DenoiserBackend(ABC):
    """denoising.py:349-386."""

    @property
    @abstractmethod
    def name(self) -> str: ...

    @abstractmethod
    def is_available(self) -> bool: ...

    @abstractmethod
    def process(self, input_dir: Path, output_dir: Path, config, progress_callback=None) -> DenoiseResult: ...


# ---- HIP backends -------------------------------------------------------------------------------------------------
_VARIANTS = {"x4plus": ("RealESRGAN_x4plus", 4), "x2plus": ("RealESRGAN_x2plus", 2),
             "anime": ("RealESRGAN_x4plus_anime_6B", 4)}


class HipRealESRGANBackend(SRBackend):
    """Counterpart of ``RealESRGANBackend`` (super_resolution.py:441-601) on the HIP engine."""

    def __init__(self, config=None, hardware=None, model_variant: str = "x4plus", gpu_id: int = 0, dtype: str = "f16"):
        if model_variant not in _VARIANTS:
            raise ValueError(f"model_variant must be one of {sorted(_VARIANTS)}")
        self.config, self.hardware, self.model_variant = config, hardware, model_variant
        self.gpu_id = getattr(config, "gpu_id", gpu_id)
        self.dtype = dtype
        self._cfg: Optional[R.PyTorchESRGANConfig] = None

    @property
    def name(self) -> str:
        return f"realesrgan_hip_{self.model_variant}"

    @property
    def supported_scales(self) -> List[int]:
        return [_VARIANTS[self.model_variant][1]]

    def is_available(self) -> bool:
        return R.is_pytorch_esrgan_available()

    def estimate_vram_usage(self, width: int, height: int, scale: int) -> int:
        """MB of device workspace — computed from the engine's real buffer plan, not the reference's 450 MB/out-MP
        heuristic (utils/gpu.py:421-426)."""
        s = _VARIANTS[self.model_variant][1]
        ht, wt = (height, width) if s == 4 else ((height + 1) // 2, (width + 1) // 2)
        px = ht * wt
        pad = -(-ht // 16) * 16 * (-(-wt // 32) * 32)
        b = px * (32 * 2 + 2 * 192 * 2 + 4 * 64 * 2 + 2 * 16 * 64 * 2) + 4 * pad * 64 * 4 + width * height * 3 * (1 + s * s)
        return int(b // (1024 * 1024)) + 70  # + weights

    def _ensure_config(self) -> R.PyTorchESRGANConfig:
        if self._cfg is None:
            model, s = _VARIANTS[self.model_variant]
            self._cfg = R.PyTorchESRGANConfig(model_name=model, scale_factor=s, gpu_id=self.gpu_id, dtype=self.dtype,
                                              tile_size=getattr(self.config, "tile_size", 0) or 0)
        return self._cfg

    def upscale_frame(self, frame: np.ndarray, scale: int = 4) -> np.ndarray:
        cfg = self._ensure_config()
        out, _ = R.get_upsampler(cfg).enhance(frame, outscale=cfg.scale_factor)
        return out

    def upscale_frames(self, input_dir: Path, output_dir: Path, scale: int = 4,
                       progress_callback: Optional[Callable[[float], None]] = None) -> SRResult:
        res = SRResult(backend_used=self.name, scale_factor=scale)
        t0 = time.time()
        cfg = self._ensure_config()
        input_dir, output_dir = Path(input_dir), Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        res.output_dir = output_dir
        frames = sorted(input_dir.glob("*.png")) or sorted(input_dir.glob("*.jpg"))
        if not frames:
            res.warnings.append("No frames found")
            return res
        for i, fp in enumerate(frames):
            ok, err = R.enhance_frame_pytorch(fp, output_dir / fp.name, cfg)
            if ok:
                res.frames_processed += 1
            else:
                res.frames_failed += 1
                res.warnings.append(f"Frame {fp.name}: {err}")
            if progress_callback:
                progress_callback((i + 1) / len(frames))
        res.processing_time_seconds = time.time() - t0
        if res.processing_time_seconds > 0 and res.frames_processed > 0:
            res.avg_fps = res.frames_processed / res.processing_time_seconds
        return res

    def clear_cache(self) -> None:
        R.clear_upsampler_cache()


class HipTAPDenoiserBackend(DenoiserBackend):
    """Counterpart of ``TAPDenoiserBackend`` (denoising.py:636-775).  The reference hard-codes ``TAPModel.RESTORMER``
    (:729); the accelerated model is NAFNet."""

    @property
    def name(self) -> str:
        return "tap_hip"

    def is_available(self) -> bool:
        return T.TAPDenoiser(T.TAPDenoiseConfig()).is_available()

    def _get_denoiser(self, config) -> T.TAPDenoiser:
        tile = getattr(config, "tile_size", 512)
        return T.TAPDenoiser(T.TAPDenoiseConfig(
            model=T.TAPModel.NAFNET, temporal_window=getattr(config, "temporal_radius", 2) * 2 + 1,
            strength=getattr(config, "strength", 1.0), preserve_grain=getattr(config, "preserve_grain", False),
            half_precision=getattr(config, "half_precision", True), tile_size=tile if tile and tile > 0 else 512,
            gpu_id=getattr(config, "gpu_id", 0)))

    def process(self, input_dir: Path, output_dir: Path, config, progress_callback=None) -> DenoiseResult:
        res = DenoiseResult(backend_used=self.name)
        t0 = time.time()
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        res.output_dir = output_dir
        try:
            tr = self._get_denoiser(config).denoise_frames(input_dir, output_dir, progress_callback)
            res.frames_processed, res.frames_failed = tr.frames_processed, tr.frames_failed
            res.avg_noise_reduction = tr.avg_psnr_improvement / 20.0
            res.peak_vram_mb = tr.peak_vram_mb
            if tr.model_used:
                res.backend_used = f"tap_hip_{tr.model_used}"
        except Exception as e:  # noqa: BLE001 — reference contract (denoising.py:768-771)
            res.warnings.append(f"TAP denoising error: {e}")
            res.frames_failed = len(list(Path(input_dir).glob("*.png")))
        res.processing_time_seconds = time.time() - t0
        return res


class RealESRGANProcessor:
    """The class ``distributed/worker.py:298-347`` of the reference imports but the reference never defines
    (SURVEY.md §8b): ``RealESRGANProcessor(model_name, scale_factor, gpu_device).process_frame(in_path, out_path)``."""

    def __init__(self, model_name: str = "realesrgan-x4plus", scale_factor: int = 4, gpu_device: int = 0):
        self.config = R.PyTorchESRGANConfig(model_name=R.convert_ncnn_model_name(model_name), scale_factor=scale_factor,
                                            gpu_id=gpu_device)

    def process_frame(self, input_path: str, output_path: str) -> bool:
        ok, err = R.enhance_frame_pytorch(Path(input_path), Path(output_path), self.config)
        if not ok:
            raise RuntimeError(err)
        return True


def make_shard_process_fn(config: Optional[R.PyTorchESRGANConfig] = None):
    """``process_fn(input_path, output_dir, gpu_id) -> (output_path, ok, err)`` for the reference's
    ``MultiGPUDistributor.distribute_frames`` (utils/multi_gpu.py:552-561)."""
    base = config or R.PyTorchESRGANConfig()

    def process_fn(input_path: Path, output_dir: Path, gpu_id: int) -> Tuple[Path, bool, Optional[str]]:
        cfg = R.PyTorchESRGANConfig(**{**base.__dict__, "gpu_id": gpu_id})
        out = Path(output_dir) / Path(input_path).name
        ok, err = R.enhance_frame_pytorch(Path(input_path), out, cfg)
        return out, ok, err

    return process_fn


def make_device_process_func(config: Optional[R.PyTorchESRGANConfig] = None):
    """``process_func(frame, device_id) -> frame`` for ``MultiGPUProcessor.process_frames``
    (infrastructure/gpu/distributor.py:690-699)."""
    base = config or R.PyTorchESRGANConfig()

    def process_func(frame: np.ndarray, device_id: int) -> np.ndarray:
        cfg = R.PyTorchESRGANConfig(**{**base.__dict__, "gpu_id": device_id})
        return R.get_upsampler(cfg).enhance(frame, outscale=cfg.scale_factor)[0]

    return process_func


# ---- compute-backend ABC (infrastructure/gpu/backends/base.py:65-215) ----------------------------------------------
@dataclass
class BackendCapabilities:
    """infrastructure/gpu/backends/base.py:28-62 (backend_type / vendor as their string values: the enums live in the
    reference's detector module)."""
    name: str
    backend_type: str = "rocm"
    vendor: str = "amd"
    supports_fp16: bool = True
    supports_fp32: bool = True
    supports_int8: bool = False
    supports_dynamic_shapes: bool = True
    supports_batching: bool = True
    max_memory_mb: int = 0
    recommended_memory_mb: int = 0
    max_batch_size: int = 32
    max_tile_size: int = 1024
    supported_models: List[str] = field(default_factory=list)

    def to_dict(self) -> dict:
        return {"name": self.name, "backend_type": self.backend_type, "vendor": self.vendor,
                "supports_fp16": self.supports_fp16, "supports_int8": self.supports_int8,
                "max_memory_mb": self.max_memory_mb, "max_batch_size": self.max_batch_size}


class HipRocmBackend:
    """The reference's `Backend` interface (initialize / cleanup / get_capabilities / allocate_memory / free_memory /
    get_memory_info / load_model / unload_model / run_inference, context manager) for MI355X.  The reference registers no
    ROCm backend (`_backend_registry`, base.py:665-670, has CUDA / Metal / Vulkan / CPU) and its CUDA one is a stub whose
    `run_inference` returns its input; here `load_model` builds the engine for the named model and `run_inference` runs it:

      Real-ESRGAN names (`RealESRGAN_x4plus`, ...): inputs = BGR uint8 frame           -> upscaled BGR uint8 frame
      "nafnet" / "restormer":                       inputs = BGR uint8 frame           -> denoised BGR uint8 frame
      "rife-v4.6":                                  inputs = (frame0, frame1[, t=0.5]) -> interpolated BGR uint8 frame
    """

    backend_type = "rocm"
    name = "ROCm (MI355X HIP kernels)"

    def __init__(self, device_id: int = 0):
        self.device_id = int(device_id)
        self._initialized = False
        self._loaded_models: dict = {}

    @property
    def is_initialized(self) -> bool:
        return self._initialized

    def initialize(self) -> bool:
        from . import _lib
        try:
            self._initialized = _lib.load().fw_device_count() > self.device_id
        except _lib.FramewrightHipError:
            self._initialized = False
        return self._initialized

    def cleanup(self) -> None:
        for name in list(self._loaded_models):
            self.unload_model(name)
        self._initialized = False

    def get_memory_info(self) -> dict:
        import torch
        free, total = torch.cuda.mem_get_info(self.device_id)
        return {"total_mb": total / 2 ** 20, "used_mb": (total - free) / 2 ** 20, "free_mb": free / 2 ** 20}

    def get_capabilities(self) -> BackendCapabilities:
        mem = self.get_memory_info() if self._initialized else {"total_mb": 0.0}
        return BackendCapabilities(name=self.name, max_memory_mb=int(mem["total_mb"]), recommended_memory_mb=int(mem["total_mb"] * 0.9),
                                   max_tile_size=0, supported_models=list(R.RRDB_MODELS) + ["nafnet", "restormer", "rife-v4.6"])

    def allocate_memory(self, size_mb: float) -> bool:
        return self._initialized and self.get_memory_info()["free_mb"] >= size_mb   # engines size their own workspaces

    def free_memory(self) -> None:
        import torch
        torch.cuda.empty_cache()

    def load_model(self, model_name: str, model_path: Optional[Path] = None, **kwargs) -> bool:
        if not self._initialized:
            return False
        dtype = kwargs.get("dtype", "f16")
        if model_name in R.RRDB_MODELS:
            cfg = R.PyTorchESRGANConfig(model_name=model_name, gpu_id=self.device_id, dtype=dtype,
                                        model_path=str(model_path) if model_path else None)
            self._loaded_models[model_name] = ("sr", R.get_upsampler(cfg))
        elif model_name in ("nafnet", "restormer"):
            dn = T.TAPDenoiser(T.TAPDenoiseConfig(model=model_name, gpu_id=self.device_id, tile_size=kwargs.get("tile_size", 0),
                                                  temporal_window=1, dtype=kwargs.get("dtype", "f16")),
                               model_dir=Path(model_path).parent if model_path else None)
            dn._load_model()
            self._loaded_models[model_name] = ("denoise", dn)
        elif model_name == "rife-v4.6":
            from . import rife as RF
            fi = RF.FrameInterpolator(model=model_name, gpu_id=self.device_id, dtype=kwargs.get("dtype", "f16"))
            fi._get_engine()
            self._loaded_models[model_name] = ("interp", fi)
        else:
            return False
        return True

    def unload_model(self, model_name: str) -> None:
        kind, obj = self._loaded_models.pop(model_name, (None, None))
        if kind == "denoise":
            obj.clear_cache()
        self.free_memory()

    def run_inference(self, model_name: str, inputs, **kwargs):
        if model_name not in self._loaded_models:
            raise ValueError(f"Model {model_name} not loaded")           # base.py:346-347
        kind, obj = self._loaded_models[model_name]
        if kind == "sr":
            return obj.enhance(inputs, outscale=kwargs.get("outscale"))[0]
        if kind == "denoise":
            return obj.denoise_clip([inputs])[0]
        f0, f1 = inputs[0], inputs[1]
        return obj._get_engine().interpolate(f0, f1, inputs[2] if len(inputs) > 2 else kwargs.get("timestep", 0.5))

    def __enter__(self):
        self.initialize()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.cleanup()
        return False


def register_with_reference() -> List[str]:
    """Add the HIP backends to the reference's selector tables if ``framewright`` is importable.  On MI355X the
    reference's own auto-detection classifies any non-NVIDIA GPU as CPU/NCNN (denoising.py:222-277), so
    the backend has to be requested by name: ``SuperResolution(backend="realesrgan_hip")``."""
    done: List[str] = []
    try:
        from framewright.processors.enhancement import super_resolution as sr  # type: ignore
        sr.SuperResolution.BACKENDS["realesrgan_hip"] = HipRealESRGANBackend
        done.append("SuperResolution.BACKENDS['realesrgan_hip']")
    except Exception:  # noqa: BLE001 - reference absent or not importable (its package __init__ is broken in the snapshot)
        pass
    try:
        from framewright.processors.enhancement import denoising as dn  # type: ignore
        dn.Denoiser.BACKENDS["tap_hip"] = HipTAPDenoiserBackend
        done.append("Denoiser.BACKENDS['tap_hip']")
    except Exception:  # noqa: BLE001
        pass
    try:   # the reference has a ROCM BackendType but registers no class for it (base.py:665-670)
        from framewright.infrastructure.gpu.backends import base as gb  # type: ignore
        from framewright.infrastructure.gpu.detector import BackendType  # type: ignore
        cls = type("HipRocmBackend", (HipRocmBackend, gb.Backend), {"backend_type": property(lambda self: BackendType.ROCM),
                                                                      "name": property(lambda self: HipRocmBackend.name)})
        gb.register_backend(BackendType.ROCM, cls)
        done.append("register_backend(BackendType.ROCM)")
    except Exception:  # noqa: BLE001
        pass
    return done
