"""The reference's in-tree ``AESRGAN`` — RRDB trunk with full-image self-attention blocks — on the MI355X kernels.

Reference: ``src/framewright/processors/aesrgan_face.py`` — ``AttentionBlock`` (:142-168), ``ResidualDenseBlock`` / ``RRDB``
(:171-204), ``AESRGAN`` (:205-269: no pixel-unshuffle front end, an AttentionBlock behind every ``num_block // num_attention``-th
RRDB, ``conv_up2`` only for scale >= 4), the network ``AESRGANFaceRestorer`` (:383+) runs on face crops.  SURVEY.md section
8f/4 ("other RRDB consumers").  Round 3: ``AESRGANFaceRestorer`` itself (config, crop, tensor round trip, ``cv2.resize`` + feathered
paste-back, directory driver) is mirrored below; face DETECTION stays a host input - the reference's detectors are RetinaFace and
OpenCV's Haar cascade, neither of which exists in this image - through an injectable ``detect(frame) -> [FaceBox]``.

One engine behind the C-ABI (``fw_aesrgan_*``, csrc/aesrgan.hip - round 1 sequenced the launches from here): every 3x3
convolution on the MFMA conv kernel (chunk-planar typed activations, the residual streams in fp32, `x5 * 0.2 + x` and the RRDB's
second residual in the conv5 epilogue), the 1x1 query / key / value projections on the pointwise GEMM, softmax(q^T k) over all
pixels as a row kernel, and the product with v as a GEMM over the pixel axis with the gamma residual epilogue.
Parity: oracle/rrdbnet_ref.py ``aesrgan_forward``, which is pinned on vectors the reference's own module produced
(tests/golden/aesrgan_attention.npz).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import shutil
import threading
import time
from dataclasses import dataclass
from enum import Enum
from pathlib import Path
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FramewrightHipError
from .synth import aesrgan_attention_positions

logger = logging.getLogger(__name__)


def _np(t) -> np.ndarray:
    return t if isinstance(t, np.ndarray) else t.detach().cpu().float().numpy()


class AESRGANEngine:
    """``AESRGAN(num_in_ch=3, num_out_ch=3, num_feat=64, num_block, scale, num_attention)`` resident on one GPU: thin owner of an
    ``fw_aesrgan*``; the weights, the workspace arena and the launches of a forward live behind the C-ABI."""

    def __init__(self, num_block: int = 23, scale: int = 2, num_attention: int = 4, dtype: str = "f16", device_id: int = 0):
        import torch
        self._lib = _lib.load()
        _lib.require_gpu()
        if scale not in (2, 4) or num_block < 1 or num_attention < 1 or num_attention > num_block:
            raise ValueError("AESRGANEngine: scale 2 or 4, 1 <= num_attention <= num_block")
        if dtype not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_lib.DTYPES)}")
        self.num_block, self.scale, self.num_attention = int(num_block), int(scale), int(num_attention)
        self.attn_after = aesrgan_attention_positions(self.num_block, self.num_attention)
        self.dtype, self.device_id = dtype, int(device_id)
        self._dev = torch.device("cuda", self.device_id)
        self._mu = threading.Lock()
        h = C.c_void_p()
        _lib.check(self._lib.fw_aesrgan_create(self.device_id, self.num_block, self.scale, self.num_attention, _lib.DTYPES[dtype], C.byref(h)))
        self._h = h
        self._loaded = False

    def load_state_dict(self, state: Mapping[str, object], attention: Mapping[str, object]) -> None:
        """``state``: the RRDB trunk / tail under BasicSR's key names (``conv_first``, ``body.{i}.rdb{1,2,3}.conv{1..5}``,
        ``conv_body``, ``conv_up1`` [, ``conv_up2``], ``conv_hr``, ``conv_last``); ``attention``: ``attn.{i}.query|key|value.
        weight|bias`` and ``attn.{i}.gamma`` for the block behind RRDB ``i`` (synth.synthetic_attention_state has the layout)."""
        names = ["conv_first", "conv_body", "conv_up1", "conv_hr", "conv_last"] + (["conv_up2"] if self.scale >= 4 else [])
        for i in range(self.num_block):
            for r in (1, 2, 3):
                names += [f"body.{i}.rdb{r}.conv{c}" for c in range(1, 6)]
        items = []
        for k in names:
            if k + ".weight" not in state or k + ".bias" not in state:
                raise FramewrightHipError(_lib.FW_ERR_INVALID, f"state dict is missing {k}")
            items += [(k + ".weight", state[k + ".weight"]), (k + ".bias", state[k + ".bias"])]
        for i in self.attn_after:
            for name in ("query", "key", "value"):
                items += [(f"attn.{i}.{name}.weight", attention[f"attn.{i}.{name}.weight"]), (f"attn.{i}.{name}.bias", attention[f"attn.{i}.{name}.bias"])]
            items.append((f"attn.{i}.gamma", attention[f"attn.{i}.gamma"]))
        for key, t in items:
            a = np.ascontiguousarray(_np(t), dtype=np.float32).reshape(-1)
            _lib.check(self._lib.fw_aesrgan_set_tensor(self._h, key.encode(), C.c_void_p(a.ctypes.data), a.size))
        _lib.check(self._lib.fw_aesrgan_finalize(self._h))
        self._loaded = True

    # ---- forward -------------------------------------------------------------------------------------------------------
    def forward_rgb(self, x):
        """x: float32 CUDA tensor H x W x 3, RGB in [0, 1].  Returns float32 (scale*H) x (scale*W) x 3, un-clamped — what
        ``AESRGAN.forward`` returns for a 1 x 3 x H x W input, NHWC (asynchronous on torch's current stream)."""
        import torch
        if not self._loaded:
            raise FramewrightHipError(_lib.FW_ERR_INVALID, "AESRGANEngine: no weights loaded")
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or x.shape[2] != 3:
            raise ValueError("forward_rgb expects a float32 CUDA tensor H x W x 3")
        if x.device != self._dev:
            raise ValueError(f"tensor is on {x.device}, engine on {self._dev}")
        x = x.contiguous()
        H, W = int(x.shape[0]), int(x.shape[1])
        out = torch.empty((H * self.scale, W * self.scale, 3), dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self._lib.fw_aesrgan_forward_rgb(self._h, C.c_void_p(x.data_ptr()), H, W, C.c_void_p(out.data_ptr()), st))
        return out

    def enhance(self, bgr: np.ndarray) -> np.ndarray:
        """uint8 BGR face crop -> uint8 BGR, scale x larger (the tensor round trip of AESRGANFaceRestorer: /255, RGB, network,
        clamp, x255)."""
        import torch
        if not isinstance(bgr, np.ndarray) or bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3:
            raise ValueError("expected an H x W x 3 uint8 BGR image")
        # one forward at a time per instance: the launches of a forward are sequenced from this thread onto the device's current
        # stream, and a second thread on the same stream would interleave with them
        with self._mu, torch.cuda.device(self._dev):
            x = torch.from_numpy(np.ascontiguousarray(bgr[:, :, ::-1])).to(self._dev).float() / 255.0
            y = self.forward_rgb(x).clamp_(0, 1)
            out = (y * 255.0).round().to(torch.uint8).cpu().numpy()
        return np.ascontiguousarray(out[:, :, ::-1])

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.fw_aesrgan_destroy(h)
        self._loaded = False


# =====================================================================================================================================
# AESRGANFaceRestorer (reference src/framewright/processors/aesrgan_face.py:51-136, 270-760): same names, fields, defaults, validation
# messages and result bookkeeping; the model is the engine above, the resize and the feathered blend are device kernels
# (fw_resize_linear_u8, fw_face_paste_u8; oracle/face_ref.py restates them).
# =====================================================================================================================================
class FaceDetectorType(Enum):
    """aesrgan_face.py:51-56."""
    RETINAFACE = "retinaface"
    MTCNN = "mtcnn"
    DLIB = "dlib"
    OPENCV = "opencv"


@dataclass
class AESRGANFaceConfig:
    """aesrgan_face.py:59-92 (field for field; ``half_precision`` selects f16 operands, which the engine uses either way)."""
    detection_threshold: float = 0.7
    enhancement_strength: float = 0.8
    preserve_identity: bool = True
    attention_scale: float = 1.0
    upscale_factor: int = 2
    paste_back: bool = True
    face_detector: FaceDetectorType = FaceDetectorType.RETINAFACE
    gpu_id: int = 0
    half_precision: bool = True

    def __post_init__(self) -> None:
        if isinstance(self.face_detector, str):
            self.face_detector = FaceDetectorType(self.face_detector)
        if not 0.0 <= self.detection_threshold <= 1.0:
            raise ValueError(f"detection_threshold must be 0-1, got {self.detection_threshold}")
        if not 0.0 <= self.enhancement_strength <= 1.0:
            raise ValueError(f"enhancement_strength must be 0-1, got {self.enhancement_strength}")
        if self.upscale_factor not in [1, 2, 4]:
            raise ValueError(f"upscale_factor must be 1, 2, or 4, got {self.upscale_factor}")


@dataclass
class FaceBox:
    """aesrgan_face.py:95-114."""
    x1: int
    y1: int
    x2: int
    y2: int
    confidence: float
    landmarks: Optional[np.ndarray] = None

    @property
    def width(self) -> int:
        return self.x2 - self.x1

    @property
    def height(self) -> int:
        return self.y2 - self.y1

    @property
    def center(self) -> Tuple[int, int]:
        return ((self.x1 + self.x2) // 2, (self.y1 + self.y2) // 2)


@dataclass
class AESRGANFaceResult:
    """aesrgan_face.py:117-136."""
    frames_processed: int = 0
    frames_failed: int = 0
    faces_enhanced: int = 0
    output_dir: Optional[Path] = None
    processing_time_seconds: float = 0.0
    peak_vram_mb: int = 0


class FaceDetector:
    """aesrgan_face.py:270-381.  The reference's backends (the ``retinaface`` package, OpenCV's Haar cascade) are third-party code
    that is absent here; ``detect_fn(frame) -> list of FaceBox`` (or of ``(x1, y1, x2, y2[, confidence])``) takes their place - without
    one, like the reference without a backend, no faces are found."""

    def __init__(self, detector_type: FaceDetectorType, gpu_id: int = 0, detect_fn: Optional[Callable] = None):
        self.detector_type = detector_type
        self.gpu_id = gpu_id
        self._detect_fn = detect_fn
        self._backend = self._detect_backend()

    def _detect_backend(self) -> Optional[str]:
        """The reference probes for `retinaface` / cv2 here (aesrgan_face.py:279-297); this mirror has the injected callable or nothing."""
        return "callable" if self._detect_fn is not None else None

    def detect(self, frame: np.ndarray) -> List[FaceBox]:
        if self._detect_fn is None:
            return []
        out = []
        for b in self._detect_fn(frame):
            if isinstance(b, FaceBox):
                out.append(b)
            else:
                out.append(FaceBox(int(b[0]), int(b[1]), int(b[2]), int(b[3]), float(b[4]) if len(b) > 4 else 1.0))
        return out


def split_aesrgan_checkpoint(state: Mapping[str, object], num_block: int, num_attention: int):
    """``AESRGAN.state_dict()`` keys -> (trunk, attention) in the engine's naming.  The reference keeps RRDBs and AttentionBlocks in ONE
    ``nn.ModuleList`` (aesrgan_face.py:228-235), so module index m of ``body.{m}.`` counts both: RRDB i sits at i + (attention blocks
    before it), its AttentionBlock right behind."""
    positions = set(aesrgan_attention_positions(num_block, num_attention))
    trunk: Dict[str, object] = {k: v for k, v in state.items() if not k.startswith("body.")}
    attn: Dict[str, object] = {}
    m = 0
    for i in range(num_block):
        pre = f"body.{m}."
        for k, v in state.items():
            if k.startswith(pre):
                trunk[f"body.{i}." + k[len(pre):]] = v
        m += 1
        if i in positions:
            pre = f"body.{m}."
            for k, v in state.items():
                if k.startswith(pre):
                    attn[f"attn.{i}." + k[len(pre):]] = v
            m += 1
    return trunk, attn


class AESRGANFaceRestorer:
    """aesrgan_face.py:383-728 on the HIP engine.  ``detect_fn`` supplies the face boxes (see FaceDetector)."""

    DEFAULT_MODEL_DIR = Path.home() / ".framewright" / "models" / "aesrgan"
    MODEL_FILE = "aesrgan_face.pth"
    NUM_BLOCK, NUM_ATTENTION = 23, 4

    def __init__(self, config: Optional[AESRGANFaceConfig] = None, model_dir: Optional[Path] = None, detect_fn: Optional[Callable] = None,
                 engine: Optional[AESRGANEngine] = None):
        self.config = config or AESRGANFaceConfig()
        env = os.environ.get("FRAMEWRIGHT_MODEL_DIR")
        self.model_dir = Path(model_dir) if model_dir else (Path(env) / "aesrgan" if env else self.DEFAULT_MODEL_DIR)
        self._model = engine
        self._device = None
        self._detect_fn = detect_fn
        self._face_detector = FaceDetector(self.config.face_detector, self.config.gpu_id, detect_fn) if engine is not None else None
        self._backend = self._detect_backend()

    def _detect_backend(self) -> Optional[str]:
        try:
            _lib.load()
            _lib.require_gpu()
        except Exception as e:   # noqa: BLE001 - the reference reports "disabled", it does not raise here
            logger.warning("HIP engine not available - AESRGAN disabled: %s", e)
            return None
        if (self.model_dir / self.MODEL_FILE).exists():
            return "aesrgan_weights"
        logger.warning("AESRGAN weights not found at %s. Will use random initialization (quality will be limited).", self.model_dir / self.MODEL_FILE)
        return "aesrgan_random"

    def is_available(self) -> bool:
        return self._backend is not None

    def _load_model(self) -> None:
        if self._model is not None and self._face_detector is not None:
            return
        if self.config.upscale_factor not in (2, 4):
            raise RuntimeError("AESRGANFaceRestorer: the HIP engine builds the x2 and x4 networks (upscale_factor 1 has no up-conv tail)")
        if self._model is None:
            from .synth import synthetic_attention_state, synthetic_rrdbnet_state
            eng = AESRGANEngine(self.NUM_BLOCK, self.config.upscale_factor, self.NUM_ATTENTION, "f16", self.config.gpu_id)
            path = self.model_dir / self.MODEL_FILE
            trunk = attn = None
            if path.exists():
                try:
                    import torch
                    ckpt = torch.load(path, map_location="cpu")
                    sd = ckpt["params"] if "params" in ckpt else (ckpt["state_dict"] if "state_dict" in ckpt else ckpt)
                    trunk, attn = split_aesrgan_checkpoint(sd, self.NUM_BLOCK, self.NUM_ATTENTION)
                    eng.load_state_dict(trunk, attn)
                    logger.info("Loaded AESRGAN weights from %s", path)
                except Exception as e:   # noqa: BLE001 - the reference warns and keeps its random initialisation
                    logger.warning("Failed to load AESRGAN weights: %s", e)
                    trunk = None
            if trunk is None:
                # the reference runs torch's default initialisation here ("quality will be limited"); seeded weights of the same
                # shapes take its place (there is no torch.nn module on this path to initialise)
                eng.load_state_dict(synthetic_rrdbnet_state(self.NUM_BLOCK, 4, seed=1234),   # (AESRGAN has no pixel-unshuffle front end: 3-channel conv_first at every scale)
                                    synthetic_attention_state(self.NUM_BLOCK, self.NUM_ATTENTION))
            self._model = eng
        import torch
        self._device = torch.device("cuda", self.config.gpu_id)
        self._face_detector = FaceDetector(self.config.face_detector, self.config.gpu_id, self._detect_fn)

    # ---- the three steps of restore_frame ---------------------------------------------------------------------------------------------
    def _extract_face(self, frame: np.ndarray, face_box: FaceBox, padding: float = 0.3):
        h, w = frame.shape[:2]
        pad_w = int(face_box.width * padding)
        pad_h = int(face_box.height * padding)
        x1 = max(0, face_box.x1 - pad_w)
        y1 = max(0, face_box.y1 - pad_h)
        x2 = min(w, face_box.x2 + pad_w)
        y2 = min(h, face_box.y2 + pad_h)
        return frame[y1:y2, x1:x2].copy(), (x1, y1, x2, y2)

    def _enhance_face_device(self, face_crop):
        """uint8 BGR crop (numpy array or CUDA tensor) -> uint8 BGR CUDA tensor, upscale_factor times larger: BGR -> RGB, / 255, the
        network, ``clip(y * 255, 0, 255)`` and a TRUNCATING cast (aesrgan_face.py:519-541)."""
        import torch
        eng = self._model
        with eng._mu, torch.cuda.device(eng._dev):
            t = face_crop if isinstance(face_crop, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(face_crop)).to(eng._dev)
            x = t.flip(2).float() / 255.0
            y = eng.forward_rgb(x)
            return (y * 255.0).clamp_(0, 255).to(torch.uint8).flip(2).contiguous()

    def _enhance_face(self, face_crop: np.ndarray) -> np.ndarray:
        return self._enhance_face_device(face_crop).cpu().numpy()

    def _paste_device(self, d, enhanced, region: Tuple[int, int, int, int]) -> None:
        """``cv2.resize(enhanced, (w, h))`` + the feathered float32 blend (aesrgan_face.py:543-584), in place on the CUDA frame ``d``."""
        import torch
        lib, eng = _lib.load(), self._model
        x1, y1, x2, y2 = region
        th, tw = y2 - y1, x2 - x1
        st = C.c_void_p(torch.cuda.current_stream(eng._dev).cuda_stream)
        resized = torch.empty((th, tw, 3), dtype=torch.uint8, device=eng._dev)
        _lib.check(lib.fw_resize_linear_u8(C.c_void_p(enhanced.data_ptr()), int(enhanced.shape[0]), int(enhanced.shape[1]), 3, C.c_void_p(resized.data_ptr()), th, tw, st))
        _lib.check(lib.fw_face_paste_u8(C.c_void_p(d.data_ptr()), int(d.shape[0]), int(d.shape[1]), x1, y1, x2, y2, C.c_void_p(resized.data_ptr()),
                                        float(self.config.enhancement_strength), st))

    def _paste_face_back(self, frame: np.ndarray, enhanced_face, region: Tuple[int, int, int, int]) -> np.ndarray:
        """The reference's method on host arrays: a new frame with the enhanced face blended into ``region``."""
        import torch
        eng = self._model
        with torch.cuda.device(eng._dev):
            enh = enhanced_face if isinstance(enhanced_face, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(enhanced_face)).to(eng._dev)
            d = torch.from_numpy(np.ascontiguousarray(frame)).to(eng._dev)
            self._paste_device(d, enh.contiguous(), region)
            return d.cpu().numpy()

    def restore_frame(self, frame: np.ndarray) -> Tuple[np.ndarray, int]:
        """aesrgan_face.py:586-626.  The frame goes to the device once: every face is cropped from the running result there (a later
        face sees the earlier ones pasted in, as in the reference), enhanced, resized and blended in place, and the result comes back
        once."""
        import torch
        if self._model is None or self._face_detector is None:
            self._load_model()
        faces = self._face_detector.detect(frame)
        faces = [f for f in faces if f.confidence >= self.config.detection_threshold]
        if not faces:
            return frame, 0
        eng = self._model
        with torch.cuda.device(eng._dev):
            d = torch.from_numpy(np.ascontiguousarray(frame)).to(eng._dev)
            for face_box in faces:
                try:
                    _, (x1, y1, x2, y2) = self._extract_face(frame, face_box)          # the region: index arithmetic only
                    enhanced_face = self._enhance_face_device(d[y1:y2, x1:x2].contiguous())
                    if self.config.paste_back:
                        self._paste_device(d, enhanced_face, (x1, y1, x2, y2))
                except Exception as e:   # noqa: BLE001 - like the reference: a face that fails is skipped
                    logger.warning("Failed to enhance face: %s", e)
                    continue
            result = d.cpu().numpy() if self.config.paste_back else frame.copy()
        return result, len(faces)

    def restore_faces(self, input_dir: Path, output_dir: Path, progress_callback: Optional[Callable[[float], None]] = None) -> AESRGANFaceResult:
        """aesrgan_face.py:628-720: every ``*.png`` (else ``*.jpg``) of ``input_dir`` -> the same name in ``output_dir``; a frame that
        fails is counted and copied through."""
        from PIL import Image
        result = AESRGANFaceResult()
        start_time = time.time()
        if not self.is_available():
            logger.error("AESRGAN face restoration not available")
            return result
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        result.output_dir = output_dir
        input_dir = Path(input_dir)
        frame_files = sorted(input_dir.glob("*.png")) or sorted(input_dir.glob("*.jpg"))
        if not frame_files:
            logger.warning("No frames found in %s", input_dir)
            return result
        total_frames = len(frame_files)
        try:
            self._load_model()
        except Exception as e:   # noqa: BLE001
            logger.error("Failed to load AESRGAN model: %s", e)
            return result
        for i, frame_file in enumerate(frame_files):
            try:
                frame = np.ascontiguousarray(np.asarray(Image.open(frame_file).convert("RGB"))[:, :, ::-1])   # BGR, as cv2.imread
                enhanced, num_faces = self.restore_frame(frame)
                result.faces_enhanced += num_faces
                Image.fromarray(np.ascontiguousarray(enhanced[:, :, ::-1])).save(output_dir / frame_file.name)
                result.frames_processed += 1
            except Exception as e:   # noqa: BLE001
                logger.error("Failed to process %s: %s", frame_file, e)
                result.frames_failed += 1
                try:
                    shutil.copy2(frame_file, output_dir / frame_file.name)
                except Exception:   # noqa: BLE001
                    pass
            if progress_callback:
                progress_callback((i + 1) / total_frames)
        result.processing_time_seconds = time.time() - start_time
        try:
            import torch
            if torch.cuda.is_available():
                result.peak_vram_mb = torch.cuda.max_memory_allocated(self.config.gpu_id) // (1024 * 1024)
                torch.cuda.reset_peak_memory_stats(self.config.gpu_id)
        except Exception:   # noqa: BLE001
            pass
        return result

    def clear_cache(self) -> None:
        if self._model is not None:
            self._model.close()
            self._model = None


def create_aesrgan_restorer(enhancement_strength: float = 0.8, preserve_identity: bool = True, gpu_id: int = 0) -> AESRGANFaceRestorer:
    """aesrgan_face.py:731-752."""
    return AESRGANFaceRestorer(AESRGANFaceConfig(enhancement_strength=enhancement_strength, preserve_identity=preserve_identity, gpu_id=gpu_id))
