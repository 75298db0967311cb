"""TEST INFRASTRUCTURE: the Material protocol backed by the numpy oracle, so that the host-side
drivers (QuadratureFieldMap, the stand-in FE loop, sharding) can be tested on CPU and so that GPU
runs can be compared call by call.  Never imported by the product."""
import numpy as np

from oracle import constitutive_np as onp


class _DM:
    def __init__(self, m):
        self._m = m

    def update(self):
        self._m.s0 = {k: v.copy() for k, v in self._m.s1.items()}

    def revert(self):
        self._m.s1 = {k: v.copy() for k, v in self._m.s0.items()}


class OracleJ2Material:
    def __init__(self, E, nu, hard):
        self.E, self.nu, self.hard = E, nu, hard
        self.last_stats = None

    gradients = {"strain": 6}
    fluxes = {"stress": 6}
    internal_state_variables = {"p": 1, "epsp": 6}
    tangent_blocks = {("stress", "strain"): (6, 6)}
    rotation_matrix = None
    material_properties = {}     # iterated by the reference's QuadratureMap.__init__ (quadrature_map.py:160-172)

    def update_material_property(self, name, value):
        raise AssertionError("no material property to update")

    @property
    def variables(self):
        return {**self.gradients, **self.fluxes, **self.internal_state_variables}

    def set_data_manager(self, n):
        self.n = n
        z = lambda d: np.zeros((n, d))  # noqa: E731
        self.s0 = dict(strain=z(6), stress=z(6), p=z(1), epsp=z(6))
        self.s1 = {k: v.copy() for k, v in self.s0.items()}
        self.data_manager = _DM(self)

    def set_initial_state_dict(self, state):
        for k, v in state.items():
            assert k in self.s0
            self.s0[k] = np.asarray(v, dtype=np.float64).reshape(self.n, -1).copy()

    def get_initial_state_dict(self):
        return self.s0

    def get_final_state_dict(self):
        return self.s1

    def integrate(self, g, dt=0):
        r = onp.j2_update(g, self.s0["epsp"], self.s0["p"][:, 0], self.E, self.nu, self.hard)
        self.s1 = dict(strain=np.array(g), stress=r["sig"], p=r["p"][:, None], epsp=r["epsp"])
        self.last_stats = dict(n_nan=int(np.isnan(r["sig"]).any()), n_plastic=int(r["plastic"].sum()))
        return r["sig"], np.hstack([r["p"][:, None], r["epsp"]]), r["Ct"]


class PackedOracleJ2Material(OracleJ2Material):
    """The same law handing its tangent out packed, like ``HIPMaterial(tangent_layout=...)``: ``"sym"`` the 21 upper-triangle
    entries, ``"coef"`` (c1, c2, c3, n[6]), ``"pack4"`` (c1, c2, c3, w) with n = dev(stress) w."""

    def __init__(self, E, nu, hard, tangent_layout):
        super().__init__(E, nu, hard)
        self.tangent_layout = tangent_layout
        self.tangent_size = {"full": 36, "sym": 21, "coef": 9, "pack4": 4}[tangent_layout]

    def integrate(self, g, dt=0):
        from dolfinx_materials_amd.conventions import pack_sym_tangent

        r = onp.j2_update(g, self.s0["epsp"], self.s0["p"][:, 0], self.E, self.nu, self.hard)
        self.s1 = dict(strain=np.array(g), stress=r["sig"], p=r["p"][:, None], epsp=r["epsp"])
        self.last_stats = dict(n_nan=int(np.isnan(r["sig"]).any()), n_plastic=int(r["plastic"].sum()))
        ct = {"full": lambda: r["Ct"], "sym": lambda: pack_sym_tangent(r["Ct"]), "coef": lambda: np.hstack([r["coef"], r["n"]]),
              "pack4": lambda: np.hstack([r["coef"], r["w"][:, None]])}[self.tangent_layout]()
        self.full_tangent = r["Ct"]
        return r["sig"], np.hstack([r["p"][:, None], r["epsp"]]), ct
