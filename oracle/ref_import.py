"""Import the importable half of the Python reference (build container only).

TEST INFRASTRUCTURE ONLY.  ``/root/reference`` does not exist on the GPU box and the
reference never travels; this module is used (a) by ``tests/golden/make_golden.py`` to
generate the committed fixtures and (b) by CPU tests that are skipped when the reference
tree is absent.

Only ``dolfinx_materials.generic`` and ``dolfinx_materials.python_materials`` can be
imported here (jax, jaxmat, equinox, dolfinx are missing).  ``generic.py:2`` imports
``dolfinx.common.Timer`` without using it, so a stub module is injected.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("DXMAT_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "dolfinx_materials", "generic.py"))


def import_reference():
    """Returns (generic_module, python_materials_module) of the reference."""
    if not reference_available():
        raise ImportError(f"reference tree not found under {REFERENCE_ROOT}")
    if "dolfinx" not in sys.modules:
        dolfinx = types.ModuleType("dolfinx")
        common = types.ModuleType("dolfinx.common")

        class Timer:  # stand-in for the unused import at generic.py:2
            def __init__(self, *a, **k):
                pass

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

        common.Timer = Timer
        dolfinx.common = common
        sys.modules["dolfinx"] = dolfinx
        sys.modules["dolfinx.common"] = common
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import dolfinx_materials.generic as generic
        import dolfinx_materials.python_materials as python_materials
    return generic, python_materials
