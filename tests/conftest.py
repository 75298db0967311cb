import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from dolfinx_materials_amd import _lib

        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu_available():
    return _gpu_count() > 0
