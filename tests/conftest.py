import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from framewright_amd import build as fw_build
    fw_build.build()
    from framewright_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
