"""Import-compatible stand-in for ``dolfinx_materials.python_materials``
(reference ``python_materials/elasticity.py:5-24``): capitalised field names ``Strain`` /
``Stress`` as in ``generic.py:134-139``."""
from .hip_material import HIPMaterial
from . import materials as _m


class LinearElasticIsotropic(HIPMaterial):
    def __init__(self, E, nu, device=0, **engine_options):
        """``LinearElasticIsotropic(E, nu)`` as in the reference; ``device`` / ``devices=[...]`` / ``lazy_isv`` ... are
        the engine's own keyword arguments (:class:`HIPMaterial`)."""
        super().__init__(
            _m.ElasticBehavior(_m.LinearElasticIsotropic(E=E, nu=nu)),
            device=device,
            gradient_name="Strain",
            flux_name="Stress",
            **engine_options,
        )
        self.E = E
        self.nu = nu

    @property
    def name(self):
        return self.__class__.__name__

    def get_Lame_parameters(self, E, nu):
        return E * nu / (1 + nu) / (1 - 2 * nu), E / 2 / (1 + nu)
