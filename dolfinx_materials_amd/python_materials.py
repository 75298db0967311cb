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
        self.C = self.compute_C(E, nu)

    @property
    def name(self):
        return self.__class__.__name__

    def get_Lame_parameters(self, E, nu):
        return E * nu / (1 + nu) / (1 - 2 * nu), E / 2 / (1 + nu)

    def compute_C(self, E, nu):
        """``elasticity.py:15-19``: ``C = 2 mu I6; C[:3, :3] += lambda`` (Mandel basis)."""
        import numpy as np

        lmbda, mu = self.get_Lame_parameters(E, nu)
        C = 2 * mu * np.eye(6)
        C[:3, :3] += lmbda
        return C

    def constitutive_update(self, eps, state, dt=0):
        """The per-point callable in the convention of ``generic.Material`` (``elasticity.py:21-24``): returns ``(C, state)`` -- the
        TANGENT, not the stress -- with ``state["Stress"]`` set in the dictionary that was passed.  (``JAXMaterial.constitutive_update``
        returns ``(stress, new_state)``, ``jaxmat.py:158-164``; ``batched_constitutive_update`` returns ``(Ct, new_state)`` in both.)"""
        import numpy as np

        known = {k: np.asarray(v, dtype=np.float64).reshape(1, -1) for k, v in state.items() if k in self.variables}
        Ct, new = self.batched_constitutive_update(np.asarray(eps, dtype=np.float64).reshape(1, -1), known, dt)
        state["Stress"] = new["Stress"][0]
        return Ct[0], state
