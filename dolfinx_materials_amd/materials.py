"""Behaviour descriptors with the constructor surface of ``jaxmat.materials`` as the reference
uses it (``tests/test_FeFp_jax.py:17-19``, ``demos/jax/elastoplasticity/plane_elastoplasticity.py:67-71``,
``demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:165-169``).

A descriptor only carries parameters and names; it selects one fused HIP kernel (a law id of
``include/dxmat.h``).  Hardening is one of the closed-form laws below, a :class:`CustomHardening`
(C expressions) or -- as in the reference, ``tests/test_FeFp_jax.py:14-19`` -- any Python callable
``yield_stress(p)`` written with arithmetic operators and numpy ufuncs: it is traced once
(``tracing.py``), differentiated symbolically and compiled into the kernel on first use.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import _lib


@dataclass
class LinearElasticIsotropic:
    """``jm.LinearElasticIsotropic(E=, nu=)``; Lame constants as ``python_materials/elasticity.py:12-13``."""

    E: float
    nu: float

    @property
    def mu(self):
        return self.E / 2 / (1 + self.nu)

    @property
    def lmbda(self):
        return self.E * self.nu / (1 + self.nu) / (1 - 2 * self.nu)

    @property
    def kappa(self):
        return self.lmbda + 2 * self.mu / 3


@dataclass
class LinearHardening:
    """R(p) = sig0 + H p (``tests/mfront/IsotropicPlasticMisesFlow.mfront:7-11``)."""

    sig0: float
    H: float


@dataclass
class VoceHardening:
    """R(p) = sig0 + (sigu - sig0)(1 - exp(-b p)) (``tests/test_FeFp_jax.py:14-15``)."""

    sig0: float
    sigu: float
    b: float


class CustomHardening:
    """An arbitrary isotropic hardening law given as C expressions, compiled into the fused kernels
    on first use (the role a Python ``yield_stress(p)`` callable plays for jaxmat,
    ``tests/test_FeFp_jax.py:14-19``)::

        jm.CustomHardening("sig0 + K * pow(p + 1e-12, n)", "K * n * pow(p + 1e-12, n - 1)",
                           sig0=250.0, K=600.0, n=0.3)

    ``R`` is R(p), ``dR`` its derivative, both in the variable ``p``, ``sig0`` and up to six named
    parameters; R(0) must equal ``sig0`` and R must be non-decreasing (the local Newton relies on
    it, as with the built-in laws)."""

    #: names that mean something else inside the generated C expression
    _RESERVED = {"p", "sig0", "c", "exp", "expm1", "log", "log1p", "log2", "log10", "pow", "sqrt", "cbrt", "tanh", "sinh",
                 "cosh", "sin", "cos", "tan", "atan", "asin", "acos", "fabs", "fmax", "fmin", "fma", "erf", "double", "float",
                 "int", "const", "return", "if", "else", "for", "while", "prm"}

    def __init__(self, R: str, dR: str, sig0: float, **params):
        import re

        if len(params) > 6:
            raise ValueError("at most 6 named parameters")
        bad = self._RESERVED & set(params)
        if bad:
            raise ValueError(f"parameter names {sorted(bad)} are reserved")
        self.sig0 = float(sig0)
        self.names = list(params)
        self.values = [float(v) for v in params.values()]
        self.R_source, self.dR_source = R, dR

        def sub(expr):
            for i, name in enumerate(self.names):
                expr = re.sub(rf"\b{re.escape(name)}\b", f"c[{i}]", expr)
            return "(" + expr + ")"

        self.expr_R, self.expr_dR = sub(R), sub(dR)

    @classmethod
    def from_callable(cls, func):
        """Trace a Python ``yield_stress(p)`` (operators + numpy ufuncs) into the two C expressions; numbers the
        callable closes over become literals, ``sig0 = yield_stress(0)``."""
        from .tracing import TracedLaw

        law = TracedLaw(func)
        self = cls(law.expr_R, law.expr_dR, sig0=law.sig0)
        self.traced = law
        return self

    def coefficients(self):
        return self.values + [0.0] * (6 - len(self.values))

    # named parameters read and written like attributes (update_material_property("yield_stress.K", ...))
    def __getattr__(self, name):
        names = self.__dict__.get("names", [])
        if name in names:
            return self.__dict__["values"][names.index(name)]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        names = self.__dict__.get("names", [])
        if name in names:
            self.__dict__["values"][names.index(name)] = float(value)
        else:
            object.__setattr__(self, name, value)


class Behavior:
    """Common part: law id, parameter vector, field names and sizes."""

    law: int = -1
    gradient_name = "strain"
    flux_name = "stress"

    def params(self):
        raise NotImplementedError

    def flat_properties(self):
        raise NotImplementedError


class SmallStrainBehavior(Behavior):
    """gradient ``strain`` (6), flux ``stress`` (6): ``jaxmat.py:166-169, :177-180``."""

    gradient_name = "strain"
    flux_name = "stress"
    ngrad = 6
    nflux = 6


class FiniteStrainBehavior(Behavior):
    """gradient ``F`` (9), flux ``PK1`` (9): ``jaxmat.py:170-171, :181-182``."""

    gradient_name = "F"
    flux_name = "PK1"
    ngrad = 9
    nflux = 9


def _check_hardening(yield_stress):
    if isinstance(yield_stress, (LinearHardening, VoceHardening, CustomHardening)):
        return yield_stress
    if callable(yield_stress):   # what the reference passes (tests/test_FeFp_jax.py:14-19)
        return CustomHardening.from_callable(yield_stress)
    raise TypeError(
        "yield_stress must be a callable yield_stress(p) (arithmetic operators and numpy ufuncs), or a "
        "materials.LinearHardening / VoceHardening / CustomHardening instance"
    )


def _hardening_params(e, y):
    if isinstance(y, LinearHardening):
        return [e.E, e.nu, y.sig0, y.H]
    if isinstance(y, CustomHardening):
        return [e.E, e.nu, y.sig0] + y.coefficients()
    return [e.E, e.nu, y.sig0, y.sigu, y.b]


def _hardening_properties(y):
    if isinstance(y, CustomHardening):
        return {"yield_stress.sig0": y.sig0, **{f"yield_stress.{n}": v for n, v in zip(y.names, y.values)}}
    return {f"yield_stress.{k}": v for k, v in vars(y).items()}


class ElasticBehavior(SmallStrainBehavior):
    """Linear isotropic elasticity as a small-strain behaviour."""

    law = _lib.LAW_ELASTIC_ISO

    def __init__(self, elasticity: LinearElasticIsotropic):
        self.elasticity = elasticity

    def params(self):
        return [self.elasticity.E, self.elasticity.nu]

    def flat_properties(self):
        return {"elasticity.E": self.elasticity.E, "elasticity.nu": self.elasticity.nu}


class vonMisesIsotropicHardening(SmallStrainBehavior):
    """Small-strain J2 plasticity with isotropic hardening (``plane_elastoplasticity.py:67-71``)."""

    def __init__(self, elasticity: LinearElasticIsotropic, yield_stress):
        self.elasticity = elasticity
        self.yield_stress = _check_hardening(yield_stress)
        self.law = (
            _lib.LAW_J2_LINEAR if isinstance(self.yield_stress, LinearHardening) else _lib.LAW_J2_VOCE
        )
        self.custom_hardening = self.yield_stress if isinstance(self.yield_stress, CustomHardening) else None

    def params(self):
        return _hardening_params(self.elasticity, self.yield_stress)

    def flat_properties(self):
        out = {"elasticity.E": self.elasticity.E, "elasticity.nu": self.elasticity.nu}
        out.update(_hardening_properties(self.yield_stress))
        return out


class FeFpJ2Plasticity(FiniteStrainBehavior):
    """Finite-strain FeFp J2 plasticity (``tests/test_FeFp_jax.py:17-19``); Voce hardening."""

    def __init__(self, elasticity: LinearElasticIsotropic, yield_stress):
        self.elasticity = elasticity
        self.yield_stress = _check_hardening(yield_stress)
        self.law = _lib.LAW_FEFP_J2_LINEAR if isinstance(self.yield_stress, LinearHardening) else _lib.LAW_FEFP_J2_VOCE
        self.custom_hardening = self.yield_stress if isinstance(self.yield_stress, CustomHardening) else None

    def params(self):
        return _hardening_params(self.elasticity, self.yield_stress)

    def flat_properties(self):
        out = {"elasticity.E": self.elasticity.E, "elasticity.nu": self.elasticity.nu}
        out.update(_hardening_properties(self.yield_stress))
        return out
