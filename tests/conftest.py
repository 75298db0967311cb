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


_hip = None


@pytest.fixture(autouse=True)
def no_hip_error_left_behind(request):
    """A GPU test must not leave the HIP runtime's "last error" set: the host application shares the runtime with the library
     and its next launch check (torch's, for one) would report an error that is not its own.  Checked --
    and cleared -- after every ``-m gpu`` test: an expected failure inside the library (a refused page-lock, a pointer query of
    ordinary memory) has to be consumed where it happens."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    global _hip
    if _hip is None:
        import ctypes

        # the runtime the process ALREADY uses (torch's, preloaded into the global symbol scope by _lib.load; else the one libdxmat.so
        # linked): never a second copy by file name -- its "last error" would be another runtime's and always 0
        _hip = False
        mapped = []
        try:   # the copy that is mapped into this process (the one libdxmat.so / torch resolved), by its path
            with open("/proc/self/maps") as maps:
                mapped = sorted({line.split()[-1] for line in maps if "libamdhip64.so" in line})
        except OSError:
            pass
        for candidate in mapped + ["libamdhip64.so"]:
            try:
                lib = ctypes.CDLL(candidate)
                lib.hipGetErrorName.restype = ctypes.c_char_p
                lib.hipGetLastError
                _hip = lib
                break
            except (OSError, AttributeError):
                continue
    if _hip:
        err = _hip.hipGetLastError()
        assert err == 0, f"the test left HIP error {err} ({_hip.hipGetErrorName(err).decode()}) in the runtime's last-error state"
