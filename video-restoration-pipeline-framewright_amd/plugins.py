"""Plugin boundary: ``ProcessorPlugin`` implementations for the reference's plugin manager.

Mirrors ``src/framewright/plugins/base.py``: ``PluginCapability`` (:13-46, the members used here), ``PluginMetadata``
(:49-96), ``PluginBase`` (:99-183), ``ProcessorPlugin`` (:186-250) and the ``@plugin`` decorator (:328-355).  When the
reference is importable its own classes are used, so ``PluginManager`` (plugins/manager.py:155-170) discovers these
plugins if this file is dropped into ``~/.framewright/plugins``; otherwise local mirrors keep the module usable.
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from enum import Enum, auto
from typing import Any, Callable, Dict, List, Optional, Set, Type

import numpy as np

logger = logging.getLogger(__name__)

try:  # the reference's own classes, when present
    from framewright.plugins.base import PluginCapability, PluginMetadata, ProcessorPlugin, plugin  # type: ignore
    USING_REFERENCE_BASE = True
except Exception:  # noqa: BLE001
    USING_REFERENCE_BASE = False

    class PluginCapability(Enum):  # plugins/base.py:13-46 (subset; names match)
        DENOISE = auto()
        UPSCALE = auto()
        INTERPOLATE = auto()
        TEMPORAL_FILTER = auto()

    @dataclass
    class PluginMetadata:  # plugins/base.py:49-96
        name: str
        version: str
        description: str
        author: str = ""
        website: str = ""
        license: str = ""
        capabilities: Set[PluginCapability] = field(default_factory=set)
        dependencies: List[str] = field(default_factory=list)
        python_packages: List[str] = field(default_factory=list)
        min_vram_mb: int = 0
        recommended_vram_mb: int = 0
        supports_cpu: bool = True
        supports_cuda: bool = True
        supports_mps: bool = False
        min_framewright_version: str = "1.0.0"
        max_framewright_version: Optional[str] = None
        settings_schema: Dict[str, Any] = field(default_factory=dict)

        def to_dict(self) -> Dict[str, Any]:  # plugins/base.py:79-96
            return {"name": self.name, "version": self.version, "description": self.description, "author": self.author,
                    "website": self.website, "license": self.license, "capabilities": [c.name for c in self.capabilities],
                    "dependencies": self.dependencies, "python_packages": self.python_packages, "min_vram_mb": self.min_vram_mb,
                    "recommended_vram_mb": self.recommended_vram_mb, "supports_cpu": self.supports_cpu,
                    "supports_cuda": self.supports_cuda, "supports_mps": self.supports_mps}

    class ProcessorPlugin(ABC):  # plugins/base.py:99-250 (PluginBase + ProcessorPlugin)
        def __init__(self):
            self._initialized = False
            self._settings: Dict[str, Any] = {}
            self._device: str = "cpu"

        @classmethod
        @abstractmethod
        def get_metadata(cls) -> PluginMetadata: ...

        def initialize(self, device: str = "cpu", settings: Optional[Dict[str, Any]] = None) -> None:
            self._device = device
            if settings:
                self._settings.update(settings)
            self._on_initialize()
            self._initialized = True

        def cleanup(self) -> None:
            self._on_cleanup()
            self._initialized = False

        def _on_initialize(self) -> None: ...

        def _on_cleanup(self) -> None: ...

        @property
        def is_initialized(self) -> bool:
            return self._initialized

        @property
        def device(self) -> str:
            return self._device

        @property
        def settings(self) -> Dict[str, Any]:
            return self._settings

        def update_settings(self, settings: Dict[str, Any]) -> None:  # plugins/base.py:151-158
            self._settings.update(settings)
            self._on_settings_changed(settings)

        def _on_settings_changed(self, settings: Dict[str, Any]) -> None:
            pass

        def validate_requirements(self) -> List[str]:  # plugins/base.py:159-183: missing packages, then the VRAM floor
            meta = self.get_metadata()
            issues = []
            for package in meta.python_packages:
                try:
                    __import__(package)
                except ImportError:
                    issues.append(f"Missing Python package: {package}")
            if self._device.startswith("cuda") and meta.min_vram_mb > 0:
                import torch
                if torch.cuda.is_available():
                    vram = torch.cuda.get_device_properties(_gpu_id(self._device)).total_memory / (1024 * 1024)
                    if vram < meta.min_vram_mb:
                        issues.append(f"Insufficient VRAM: {vram:.0f}MB < {meta.min_vram_mb}MB required")
            return issues

        @abstractmethod
        def process_frame(self, frame: np.ndarray, frame_number: int, context: Optional[Dict[str, Any]] = None) -> np.ndarray: ...

        def process_batch(self, frames: List[np.ndarray], start_frame: int, context: Optional[Dict[str, Any]] = None):
            return [self.process_frame(f, start_frame + i, context) for i, f in enumerate(frames)]

        def get_temporal_radius(self) -> int:
            return 0

        def supports_batch(self) -> bool:
            return False

        def estimate_output_size(self, input_size: tuple) -> tuple:
            return input_size

        def get_progress_weight(self) -> float:
            return 1.0

    def plugin(name: str, version: str, description: str, capabilities: Optional[Set[PluginCapability]] = None,
               **metadata_kwargs) -> Callable[[Type], Type]:  # plugins/base.py:328-355
        def decorator(cls):
            meta = PluginMetadata(name=name, version=version, description=description, capabilities=capabilities or set(),
                                  **metadata_kwargs)
            cls.get_metadata = classmethod(lambda c: meta)
            cls.__abstractmethods__ = frozenset(getattr(cls, "__abstractmethods__", frozenset()) - {"get_metadata"})
            return cls
        return decorator


def _gpu_id(device: str) -> int:
    """"cuda" / "cuda:3" (plugins/base.py:113-116,173) -> ordinal; "cpu" is refused: there is no CPU path."""
    if not device.startswith("cuda"):
        raise RuntimeError(f"framewright_amd plugins need a GPU device string ('cuda[:i]'), got '{device}'")
    return int(device.split(":")[1]) if ":" in device else 0


# NOTE: the reference's @plugin decorator (plugins/base.py:328-355) assigns get_metadata AFTER the class is created, when
# ABCMeta has already frozen __abstractmethods__, so a class that relies on it alone cannot be instantiated.  The
# plugins below therefore define get_metadata themselves (the decorator stays available for third parties).
_SR_META = PluginMetadata(name="realesrgan_mi355x", version="0.1.0",
                          description="Real-ESRGAN x2/x4 on MI355X (hand-written HIP kernels)",
                          capabilities={PluginCapability.UPSCALE}, supports_cpu=False, min_vram_mb=2000)
_TAP_META = PluginMetadata(name="tap_denoise_mi355x", version="0.1.0",
                           description="NAFNet temporal denoise on MI355X (hand-written HIP kernels)",
                           capabilities={PluginCapability.DENOISE}, supports_cpu=False, min_vram_mb=2000)


_RIFE_META = PluginMetadata(name="rife_mi355x", version="0.1.0",
                            description="RIFE v4.6 frame interpolation on MI355X (hand-written HIP kernels)",
                            capabilities={PluginCapability.INTERPOLATE}, supports_cpu=False, min_vram_mb=2000)


class RealESRGANPlugin(ProcessorPlugin):
    @classmethod
    def get_metadata(cls) -> PluginMetadata:
        return _SR_META

    def _on_initialize(self) -> None:
        from . import realesrgan as R
        self._cfg = R.PyTorchESRGANConfig(model_name=self._settings.get("model_name", "RealESRGAN_x4plus"),
                                          scale_factor=int(self._settings.get("scale_factor", 4)),
                                          tile_size=int(self._settings.get("tile_size", 0)), gpu_id=_gpu_id(self._device),
                                          dtype=self._settings.get("dtype", "f16"))
        self._cfg.validate()
        self._up = R.get_upsampler(self._cfg)

    def process_frame(self, frame, frame_number, context=None):
        return self._up.enhance(frame, outscale=self._cfg.scale_factor)[0]

    def estimate_output_size(self, input_size):
        s = int(self._settings.get("scale_factor", 4))
        return (input_size[0] * s, input_size[1] * s)

    def get_progress_weight(self) -> float:
        return 10.0


class TAPDenoisePlugin(ProcessorPlugin):
    @classmethod
    def get_metadata(cls) -> PluginMetadata:
        return _TAP_META

    def _on_initialize(self) -> None:
        from . import tap_denoise as T
        self._dn = T.TAPDenoiser(T.TAPDenoiseConfig(model=self._settings.get("model", T.TAPModel.RESTORMER),
                                                    temporal_window=int(self._settings.get("temporal_window", 5)),
                                                    strength=float(self._settings.get("strength", 1.0)),
                                                    tile_size=int(self._settings.get("tile_size", 512)),
                                                    gpu_id=_gpu_id(self._device)))

    def _on_cleanup(self) -> None:
        self._dn.clear_cache()

    def get_temporal_radius(self) -> int:
        return self._dn.config.temporal_window // 2 if getattr(self, "_dn", None) else 2

    def supports_batch(self) -> bool:
        return True

    def process_frame(self, frame, frame_number, context=None):
        """``context`` may carry ``{"frames": [...], "index": i}`` (the temporal neighbourhood); without it the frame is
        denoised with a window of one, as ``_denoise_with_temporal_window`` does at clip ends of length 1."""
        if context and "frames" in context:
            return self._dn.denoise_clip(context["frames"], only=[int(context.get("index", 0))])[0]
        return self._dn.denoise_clip([frame])[0]

    def process_batch(self, frames, start_frame, context=None):
        return self._dn.denoise_clip(list(frames))


class RIFEInterpolatePlugin(ProcessorPlugin):
    """IFNet v4.6 behind the plugin surface (temporal radius 1).  A frame-in / frame-out
    interface cannot change the frame count, so ``process_frame`` returns the frame half-way between ``frame`` and
    ``context["next_frame"]`` (the frame itself at the end of the clip) and ``process_batch`` the x2 sequence
    ``[f0, mid01, f1, ..., f_{n-1}]`` the reference's ``rife -n 2n`` pass produces (interpolation.py:628-650)."""

    @classmethod
    def get_metadata(cls) -> PluginMetadata:
        return _RIFE_META

    def _on_initialize(self) -> None:
        from . import rife as RF
        self._fi = RF.FrameInterpolator(model=self._settings.get("model", "rife-v4.6"), gpu_id=_gpu_id(self._device),
                                        dtype=self._settings.get("dtype", "f16"))

    def get_temporal_radius(self) -> int:
        return 1

    def supports_batch(self) -> bool:
        return True

    def process_frame(self, frame, frame_number, context=None):
        nxt = (context or {}).get("next_frame")
        if nxt is None:
            return frame
        return self._fi._get_engine().interpolate(frame, nxt)

    def process_batch(self, frames, start_frame, context=None):
        return self._fi.double(list(frames))
