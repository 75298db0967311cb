"""dolfinx_materials_amd -- MI355X-native batched constitutive-update engine.

Drop-in for ONE path of bleyerj/dolfinx_materials: ``QuadratureMap.update()`` ->
``material.integrate(gradients)`` (reference ``dolfinx_materials/quadrature_map.py:297-334``,
``dolfinx_materials/jaxmat.py:208-234``).  The Python classes here implement the reference's
duck-typed ``Material`` protocol; all arithmetic runs in hand-written HIP kernels for gfx950
behind the C ABI of ``include/dxmat.h`` (``libdxmat.so``).  There is no CPU fallback.
"""
__version__ = "0.1.0"


class PerformanceWarning(UserWarning):
    """Same role as ``dolfinx_materials.PerformanceWarning`` (reference ``__init__.py:12-15``)."""


from .hip_material import HIPMaterial, DataManager  # noqa: E402,F401
from . import materials  # noqa: E402,F401
