"""framewright_amd — MI355X-native engine for FrameWright's per-frame conv-net hot path.

Host code is Python (the reference is Python); all arithmetic runs in hand-written gfx950 HIP kernels reached
through the C-ABI of ``lib/libframewright_hip.so`` (``include/framewright_hip.h``).  There is no CPU fallback:
importing the operator modules works anywhere, but creating an engine without the library or without a GPU
raises ``FramewrightHipError``.
"""
__version__ = "0.1.0"
