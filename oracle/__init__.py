"""CPU oracle for the constitutive-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dolfinx_materials_amd/`` (the
product) imports this package: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may use it, as the *checker*.

Parity pinning status (see DESIGN.md "Oracle"):

* ``elastic_iso``          -- PINNED against the reference itself
  (``dolfinx_materials/python_materials/elasticity.py:12-24`` run through
  ``dolfinx_materials/generic.py:176-189`` in the build container; vectors
  committed under ``tests/golden/``).
* ``j2`` (linear hardening) -- PINNED against the closed-form radial return
  that is in the reference tree as MFront source
  (``tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77``) and the
  analytical known answer of ``tests/mfront/test_elastoplasticity.py:31-36``.
* ``j2`` (Voce hardening)   -- "parity unpinned" w.r.t. ``jaxmat`` (third-party,
  pinned only as ``jaxmat>=0.0.1`` in ``setup.cfg:20``, absent here); restates
  the published radial-return algorithm (Simo & Hughes 1998, Box 3.2) with the
  Voce law of ``tests/test_FeFp_jax.py:14-15``; tangent pinned by AD/FD.
* ``fefp_j2``               -- "parity unpinned" (``tests/test_FeFp_jax.py`` has
  no assertions); the algorithm is the build's own documented choice.
"""
