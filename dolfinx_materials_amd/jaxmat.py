"""Import-compatible stand-in for ``dolfinx_materials.jaxmat`` (reference ``jaxmat.py:141-234``).

A user script changes two imports and nothing else::

    from dolfinx_materials_amd.jaxmat import JAXMaterial
    import dolfinx_materials_amd.materials as jm

``JAXMaterial(behavior)`` is :class:`dolfinx_materials_amd.HIPMaterial`: same constructor
argument, same protocol, fused HIP kernels instead of ``jit(vmap(jacfwd(...)))``.
"""
from .hip_material import DataManager, HIPMaterial  # noqa: F401
from .materials import FiniteStrainBehavior, SmallStrainBehavior  # noqa: F401

JAXMaterial = HIPMaterial
