"""Importable alias for the ``video-restoration-pipeline-framewright_amd/`` source directory.

The build contract fixes that directory name, which is not a valid Python identifier; this shim makes its
contents importable as ``framewright_amd`` (``framewright_amd.realesrgan``, ``framewright_amd.synth`` ...).
"""
from pathlib import Path as _Path

_IMPL = _Path(__file__).resolve().parent.parent / "video-restoration-pipeline-framewright_amd"
if not (_IMPL / "__init__.py").exists():  # pragma: no cover
    raise ImportError(f"framewright_amd: implementation directory missing: {_IMPL}")
__path__ = [str(_IMPL)]
__file__ = str(_IMPL / "__init__.py")
exec(compile((_IMPL / "__init__.py").read_text(), __file__, "exec"))
