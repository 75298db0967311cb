import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # traced / custom hardening laws compile into private copies of the library: keep those of the test runs inside the tree
    # (git- and gpurun-ignored `_jit/`), not under the user's ~/.cache (the product's default, _lib.jit_cache_dir)
    os.environ.setdefault("DXM_JIT_CACHE", os.path.join(ROOT, "_jit"))


def pytest_sessionstart(session):
    """The suites need the in-tree native pieces (`__graft_entry__.build()` makes them).  If one is missing --
    a fresh clone where build() has not run -- build it here; existing ones are left alone (no rebuild on
    the GPU box, where the built files travel with the snapshot)."""
    import subprocess

    for lib, mk in ((os.path.join(ROOT, "dolfinx_materials_amd", "libdxmat.so"), os.path.join(ROOT, "dolfinx_materials_amd", "csrc")),
                    (os.path.join(ROOT, "oracle", "liboracle_dxmat.so"), os.path.join(ROOT, "oracle"))):
        if not os.path.exists(lib):
            subprocess.run(["make", "-C", mk], check=True)


def _gpu_count():
    try:
        from dolfinx_materials_amd import _lib

        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu_available():
    return _gpu_count() > 0
