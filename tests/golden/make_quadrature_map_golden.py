"""Generates ``tests/golden/quadrature_map_ref.npz``: the per-Gauss-point fields the REFERENCE's own ``QuadratureMap``
leaves after ``update()`` / ``advance()`` (build container only: imports ``/root/reference``).

dolfinx, ufl and basix are not installed here; ``oracle/dolfinx_doubles.py`` replaces what the reference's
``quadrature_map.py`` / ``quadrature_function.py`` / ``utils.py`` call INTO them by numpy-backed doubles, so that the
reference's own code runs unmodified: ``QuadratureMap.__init__ / register_gradient / update / advance / update_fluxes /
update_internal_state_variables / get_gradient_vals / initialize_state``, ``QuadratureExpression.eval``, ``_get_vals``,
``_update_vals``, ``_build_cell_to_dofs_map``.  The material behind it is the oracle-backed J2 + Voce law of
``tests/oracle_material.py`` (the law itself is pinned elsewhere; what this fixture pins is the CALLER: which rows a map over
all cells / over a subset of cells hands to ``integrate``, where flux, tangent and internal state variables land in the
quadrature Functions, what ``advance`` does).

    python tests/golden/make_quadrature_map_golden.py

Two cases (37 hexahedra x 8 Gauss points: all cells, and 20 of them), the same sequence each:
update, update (a second Newton iterate), advance, update, advance, update (unloading), advance; recorded after every
operation: stress, jacobian_flatten, p, epsp ``x.array``.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import constitutive_np as onp  # noqa: E402
from oracle import dolfinx_doubles as dd  # noqa: E402
from oracle.ref_import import REFERENCE_ROOT  # noqa: E402
from oracle_material import OracleJ2Material  # noqa: E402

E, NU, SIG0, SIGU, B = 70e3, 0.3, 350.0, 500.0, 1e3   # demos/jax/elastoplasticity/plane_elastoplasticity.py:60-71
NCELL, NQP = 37, 8
OPS = ["update", "update", "advance", "update", "advance", "update", "advance"]
STRAIN_OF_OP = [0, 1, None, 2, None, 3, None]


def strains():
    rng = np.random.default_rng(2024)
    n = NCELL * NQP
    mu = E / 2 / (1 + NU)
    ey = SIG0 / (2 * mu) * np.sqrt(2.0 / 3.0)
    d = rng.standard_normal((n, 6))
    d /= np.linalg.norm(d, axis=1)[:, None]
    hat = d * (rng.uniform(0.0, 4.0, n) * ey)[:, None]
    wiggle = 0.05 * ey * rng.standard_normal((n, 6))
    return np.stack([0.6 * hat, 0.6 * hat + wiggle, hat, 0.4 * hat])


def fields(q):
    return {"stress": q.fluxes["stress"].x.array.copy(), "jacobian": q.jacobian_flatten.x.array.copy(),
            "p": q.internal_state_variables["p"].x.array.copy(), "epsp": q.internal_state_variables["epsp"].x.array.copy()}


def run(qm, cells, eps_all):
    now = {"k": 0}
    mesh = dd.Mesh(NCELL, "hexahedron", 3)
    q = qm.QuadratureMap(mesh, 2, OracleJ2Material(E, NU, onp.VoceHardening(SIG0, SIGU, B)), cells=cells)
    q.register_gradient("strain", dd.PointwiseExpression(lambda c: eps_all[now["k"]].reshape(NCELL, NQP * 6)[c], 6))
    out = []
    for op, k in zip(OPS, STRAIN_OF_OP):
        if op == "update":
            now["k"] = k
            q.update()
        else:
            q.advance()
        out.append(fields(q))
    return out


def main():
    eps_all = strains()
    subset = np.sort(np.random.default_rng(5).choice(NCELL, size=20, replace=False)).astype(np.int32)
    with dd.installed(REFERENCE_ROOT) as qm:
        full = run(qm, None, eps_all)
        part = run(qm, subset, eps_all)
    save = {"E": E, "nu": NU, "sig0": SIG0, "sigu": SIGU, "b": B, "ncell": NCELL, "nqp": NQP, "strains": eps_all, "subset": subset,
            "ops": np.array(OPS), "strain_of_op": np.array([-1 if k is None else k for k in STRAIN_OF_OP])}
    for tag, rec in (("full", full), ("subset", part)):
        for i, f in enumerate(rec):
            for name, a in f.items():
                save[f"{tag}_{i}_{name}"] = a
    assert np.abs(full[0]["stress"]).max() > 100.0 and np.abs(full[2]["p"]).max() > 0.0
    other = np.setdiff1d(np.arange(NCELL), subset)
    assert not part[-1]["stress"].reshape(NCELL, -1)[other].any()       # a subset map leaves the other cells alone
    np.savez_compressed(os.path.join(HERE, "quadrature_map_ref.npz"), **save)
    print("wrote quadrature_map_ref.npz:", {k: v.shape for k, v in save.items() if hasattr(v, "shape") and v.ndim}.__len__(), "arrays")


if __name__ == "__main__":
    main()
