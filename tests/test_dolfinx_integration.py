"""The engine behind the REAL ``QuadratureMap``.  Needs ``dolfinx`` and the
reference package ``dolfinx_materials`` (hence jax): neither exists in the build container nor on the GPU box, so
this module is skipped there; it is the test a maintainer runs after ``pip install`` -- the script of
``INTEGRATION.md`` section 1 with assertions.

Checked: (i) ``QuadratureMap.update()`` drives ``HIPMaterial.integrate`` and the quadrature Functions receive
the oracle's stress / tangent / state; (ii) ``NonlinearMaterialProblem.solve()`` reproduces the closed-form
uniaxial answer in 3-D (the analogue of ``tests/mfront/test_elastoplasticity.py:14-36``); (iii)
``Hex8Mesh.from_dolfinx`` / ``SimplexMesh.from_dolfinx`` evaluate the same strain at the same Gauss points as the
compiled UFL expression."""
import numpy as np
import pytest

dolfinx = pytest.importorskip("dolfinx")
pytest.importorskip("dolfinx_materials")
pytestmark = pytest.mark.gpu

E, NU, SIG0, H = 70e3, 0.3, 250.0, 5e3


def _setup(n=3, deg_quad=2, accelerated=False, tangent_layout="full"):
    import ufl
    from dolfinx import fem, mesh
    if accelerated:   # the third import swap of INTEGRATION.md section 1
        from dolfinx_materials_amd.quadrature_map import QuadratureMap
    else:
        from dolfinx_materials.quadrature_map import QuadratureMap
    from dolfinx_materials.utils import symmetric_tensor_to_vector
    from mpi4py import MPI

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    domain = mesh.create_unit_cube(MPI.COMM_WORLD, n, n, n, mesh.CellType.hexahedron)
    V = fem.functionspace(domain, ("P", 1, (3,)))
    u = fem.Function(V, name="Displacement")
    material = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)),
                           tangent_layout=tangent_layout)
    qmap = QuadratureMap(domain, deg_quad, material)
    strain = lambda w: symmetric_tensor_to_vector(ufl.sym(ufl.grad(w)))   # noqa: E731  (Mandel: utils.py:146-165)
    qmap.register_gradient(material.gradient_names[0], strain(u))
    return domain, V, u, material, qmap, strain


@pytest.mark.parametrize("accelerated", [False, True])
def test_update_fills_the_quadrature_functions_with_the_oracle_result(accelerated):
    from oracle import constitutive_np as onp

    domain, V, u, material, qmap, _ = _setup(accelerated=accelerated)
    rng = np.random.default_rng(0)
    A = 4e-3 * rng.standard_normal((3, 3))
    x = V.tabulate_dof_coordinates()
    u.x.array[:] = (x @ A.T).reshape(-1)                     # homogeneous displacement gradient A
    qmap.update()
    npts = len(qmap.dofs)
    eps = np.tile(onp.tensor_to_mandel(0.5 * (A + A.T)[None])[0], (npts, 1))
    ref = onp.j2_update(eps, np.zeros((npts, 6)), np.zeros(npts), E, NU, onp.LinearHardening(SIG0, H))
    assert ref["plastic"].all()
    sig = qmap.fluxes["stress"].x.array.reshape(-1, 6)
    assert np.abs(sig - ref["sig"]).max() < 1e-9 * np.abs(ref["sig"]).max()
    ct = qmap.jacobian_flatten.x.array.reshape(-1, 36)
    assert np.abs(ct - ref["Ct"].reshape(-1, 36)).max() < 1e-9 * np.abs(ref["Ct"]).max()
    qmap.advance()
    assert np.abs(qmap.internal_state_variables["p"].x.array - ref["p"]).max() < 1e-12


def test_accelerated_quadrature_map_is_the_reference_class_with_three_methods_replaced():
    """`dolfinx_materials_amd.quadrature_map.QuadratureMap`: same constructor, same attributes and forms; after the same
    displacement history its Functions hold bit for bit what the reference class leaves in its own, the flux and
    `jacobian_flatten` memory is the material's (page-locked) output, the gradient Function's memory its input."""
    from dolfinx_materials.quadrature_map import QuadratureMap as Reference

    from dolfinx_materials_amd.quadrature_map import AcceleratedUpdate, QuadratureMap

    assert issubclass(QuadratureMap, Reference) and issubclass(QuadratureMap, AcceleratedUpdate)
    ref = _setup(accelerated=False)
    acc = _setup(accelerated=True)
    rng = np.random.default_rng(3)
    x = ref[1].tabulate_dof_coordinates()
    for step in range(3):
        A = (2e-3 + 2e-3 * step) * rng.standard_normal((3, 3))
        for domain, V, u, material, qmap, _ in (ref, acc):
            u.x.array[:] = (x @ A.T).reshape(-1)
            qmap.update()
            qmap.update()
        for name in ("stress",):
            assert np.array_equal(ref[4].fluxes[name].x.array, acc[4].fluxes[name].x.array)
        assert np.array_equal(ref[4].jacobian_flatten.x.array, acc[4].jacobian_flatten.x.array)
        for domain, V, u, material, qmap, _ in (ref, acc):
            qmap.advance()
        for name in ("p", "epsp"):
            assert np.array_equal(ref[4].internal_state_variables[name].x.array, acc[4].internal_state_variables[name].x.array)
    m, q = acc[3], acc[4]
    assert set(m._bound) == {"flux", "tangent", "gradient", "isv:p", "isv:epsp"}
    assert m._bound["flux"].ctypes.data == q.fluxes["stress"].x.array.ctypes.data
    assert m._bound["tangent"].ctypes.data == q.jacobian_flatten.x.array.ctypes.data
    assert m._bound["gradient"].ctypes.data == q.gradients["strain"].function.x.array.ctypes.data
    q.close()


@pytest.mark.parametrize("accelerated", [False, True])
def test_snes_solve_reaches_the_closed_form_uniaxial_answer(accelerated):
    import ufl
    from dolfinx import fem
    from dolfinx_materials.solvers import NonlinearMaterialProblem

    domain, V, u, material, qmap, strain = _setup(n=3, accelerated=accelerated)
    du, v = ufl.TrialFunction(V), ufl.TestFunction(V)
    sig = qmap.fluxes["stress"]
    Res = ufl.dot(sig, strain(v)) * qmap.dx
    Jac = qmap.derivative(Res, u, du)
    fdim = domain.topology.dim - 1

    def face(axis, value):
        facets = dolfinx.mesh.locate_entities_boundary(domain, fdim, lambda x: np.isclose(x[axis], value))
        Vs, _ = V.sub(axis).collapse()
        return fem.locate_dofs_topological((V.sub(axis), Vs), fdim, facets), Vs

    ux = None
    bcs = []
    for axis, value in ((0, 0.0), (1, 0.0), (2, 0.0), (0, 1.0)):
        dofs, Vs = face(axis, value)
        g = fem.Function(Vs)
        if value == 1.0:
            ux = g
        bcs.append(fem.dirichletbc(g, dofs, V.sub(axis)))
    opts = {"snes_type": "newtonls", "snes_linesearch_type": "none", "snes_atol": 1e-10, "snes_rtol": 1e-10,
            "ksp_type": "preonly", "pc_type": "lu"}   # tests/uniaxial_tension.py:74-82
    problem = NonlinearMaterialProblem(qmap, Res, u, bcs=bcs, J=Jac, petsc_options_prefix="amd", petsc_options=opts)
    qmap.update()
    for k in range(1, 9):
        exx = 2e-2 * k / 8
        ux.x.array[:] = exx
        problem.solve()
    expect = (SIG0 + H * exx) / (1 + H / E)
    sxx = qmap.fluxes["stress"].x.array.reshape(-1, 6)[:, 0]
    assert np.allclose(sxx, expect, rtol=1e-7)


@pytest.mark.parametrize("layout", ["sym", "coef", "pack4"])
def test_tangent_form_over_a_packed_jacobian_function_assembles_the_same_matrix(layout):
    """SURVEY 8(f) row 4, the UFL side: the accelerated map of a packed-tangent material has a ``jacobian_flatten`` of 21 / 9 / 4
    components and ``jacobians[block]`` written in terms of it (and of the stress Function for ``"pack4"``);
    ``qmap.derivative(Res, u, du)`` -- the reference's method, unchanged -- assembled over it gives the matrix of the full
    36-component map at the same plastic state."""
    import ufl
    from dolfinx import fem
    from dolfinx.fem.petsc import assemble_matrix

    full = _setup(accelerated=True)
    packed = _setup(accelerated=True, tangent_layout=layout)
    width = {"sym": 21, "coef": 9, "pack4": 4}[layout]
    npts = len(packed[4].dofs)
    assert packed[4].jacobian_flatten.x.array.size == npts * width and full[4].jacobian_flatten.x.array.size == npts * 36
    x = full[1].tabulate_dof_coordinates()
    mats = []
    for domain, V, u, material, qmap, strain in (full, packed):
        du, v = ufl.TrialFunction(V), ufl.TestFunction(V)
        Res = ufl.dot(qmap.fluxes["stress"], strain(v)) * qmap.dx
        Jac = fem.form(qmap.derivative(Res, u, du))
        u.x.array[:] = (x @ (5e-3 * np.random.default_rng(7).standard_normal((3, 3))).T).reshape(-1)
        u.x.array[:] += 3e-4 * np.sin(7.0 * x).reshape(-1)      # a non-homogeneous state: every point its own direction
        qmap.update()
        assert material.last_stats["n_plastic"] > 0
        A = assemble_matrix(Jac)
        A.assemble()
        mats.append(A)
    assert np.array_equal(full[4].fluxes["stress"].x.array, packed[4].fluxes["stress"].x.array)
    diff = mats[0].copy()
    diff.axpy(-1.0, mats[1])
    assert diff.norm() <= 1e-12 * mats[0].norm()
    for _, _, _, _, qmap, _ in (full, packed):
        qmap.close()


@pytest.mark.parametrize("layout", ["sym", "pack4"])
def test_snes_solve_with_a_packed_tangent_reaches_the_closed_form_answer_in_as_many_iterations(layout):
    """The Newton loop of ``tests/uniaxial_tension.py`` in 3-D over a packed-tangent map: same closed-form stress, and the
    iteration counts of the full-layout run (the tangent is the same operator)."""
    its = {}
    for lay in ("full", layout):
        its[lay] = _uniaxial(_setup(n=3, accelerated=True, tangent_layout=lay))
    assert its[layout] == its["full"]


def _uniaxial(setup):
    import ufl
    from dolfinx import fem
    from dolfinx_materials.solvers import NonlinearMaterialProblem

    domain, V, u, material, qmap, strain = setup
    du, v = ufl.TrialFunction(V), ufl.TestFunction(V)
    Res = ufl.dot(qmap.fluxes["stress"], strain(v)) * qmap.dx
    Jac = qmap.derivative(Res, u, du)
    fdim = domain.topology.dim - 1
    ux, bcs = None, []
    for axis, value in ((0, 0.0), (1, 0.0), (2, 0.0), (0, 1.0)):
        facets = dolfinx.mesh.locate_entities_boundary(domain, fdim, lambda x: np.isclose(x[axis], value))
        Vs, _ = V.sub(axis).collapse()
        dofs = fem.locate_dofs_topological((V.sub(axis), Vs), fdim, facets)
        g = fem.Function(Vs)
        if value == 1.0:
            ux = g
        bcs.append(fem.dirichletbc(g, dofs, V.sub(axis)))
    opts = {"snes_type": "newtonls", "snes_linesearch_type": "none", "snes_atol": 1e-10, "snes_rtol": 1e-10,
            "ksp_type": "preonly", "pc_type": "lu"}   # tests/uniaxial_tension.py:74-82
    problem = NonlinearMaterialProblem(qmap, Res, u, bcs=bcs, J=Jac, petsc_options_prefix="amd_packed", petsc_options=opts)
    qmap.update()
    iterations = []
    for k in range(1, 9):
        exx = 2e-2 * k / 8
        ux.x.array[:] = exx
        problem.solve()
        iterations.append(problem.solver.getIterationNumber())   # the SNES of solvers.py:182-196
    expect = (SIG0 + H * exx) / (1 + H / E)
    assert np.allclose(qmap.fluxes["stress"].x.array.reshape(-1, 6)[:, 0], expect, rtol=1e-7)
    qmap.close()
    return iterations


def test_device_gradient_adapter_matches_the_ufl_expression():
    from dolfinx_materials_amd.gradient import Hex8Mesh

    domain, V, u, material, qmap, _ = _setup(n=3)
    rng = np.random.default_rng(1)
    u.x.array[:] = 1e-3 * rng.standard_normal(u.x.array.size)
    qmap.update()                                            # fills the strain Function through fem.Expression
    eps_ufl = qmap.gradients["strain"].function.x.array.reshape(-1, 6)
    dmesh = Hex8Mesh.from_dolfinx(V, 2)
    assert dmesh.npoints == eps_ufl.shape[0]
    sig_dev, _, _ = material.integrate_displacement(dmesh, u.x.array)
    sig_ufl = qmap.fluxes["stress"].x.array.reshape(-1, 6)
    assert np.abs(np.asarray(sig_dev) - sig_ufl).max() < 1e-9 * np.abs(sig_ufl).max()


@pytest.mark.parametrize("cell,order", [("tetrahedron", 2), ("tetrahedron", 1), ("triangle", 2)])
def test_simplex_adapter_matches_the_ufl_expression(cell, order):
    """`SimplexMesh.from_dolfinx` on the spaces of the reference's demos (P2 on gmsh tetrahedra with quadrature degree 2,
    finite_strain_elastoplasticity.py:115-117; P2 triangles in plane strain, plane_elastoplasticity.py:96-100): the
    strain the device evaluates from ``u.x.array`` is the one ``fem.Expression`` tabulates, point for point."""
    import ufl
    from dolfinx import fem, mesh
    from dolfinx_materials.quadrature_map import QuadratureMap
    from mpi4py import MPI

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.gradient import SimplexMesh
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    if cell == "triangle":
        domain = mesh.create_unit_square(MPI.COMM_WORLD, 4, 4, mesh.CellType.triangle)
    else:
        domain = mesh.create_unit_cube(MPI.COMM_WORLD, 3, 3, 3, mesh.CellType.tetrahedron)
    tdim = domain.topology.dim
    V = fem.functionspace(domain, ("P", order, (tdim,)))
    u = fem.Function(V)
    material = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)))
    qmap = QuadratureMap(domain, 2, material)
    e = ufl.sym(ufl.grad(u))
    r2 = np.sqrt(2.0)
    if tdim == 2:   # plane strain embedding (plane_elastoplasticity.py:114-124)
        strain = ufl.as_vector([e[0, 0], e[1, 1], 0.0, r2 * e[0, 1], 0.0, 0.0])
    else:
        strain = ufl.as_vector([e[0, 0], e[1, 1], e[2, 2], r2 * e[0, 1], r2 * e[0, 2], r2 * e[1, 2]])
    qmap.register_gradient("strain", strain)
    rng = np.random.default_rng(2)
    u.x.array[:] = 2e-3 * rng.standard_normal(u.x.array.size)
    qmap.update()
    sig_ufl = qmap.fluxes["stress"].x.array.reshape(-1, 6)
    dmesh = SimplexMesh.from_dolfinx(V, 2)
    assert dmesh.npoints == sig_ufl.shape[0] and dmesh.displacement_size == u.x.array.size
    sig_dev, _, _ = material.integrate_displacement(dmesh, u.x.array)
    assert np.abs(np.asarray(sig_dev) - sig_ufl).max() < 1e-9 * np.abs(sig_ufl).max()


def test_two_maps_over_disjoint_cells_equal_the_reference_class():
    """The multi-material pattern of ``demos/multimaterials/multimaterials.py:253-257`` -- one ``QuadratureMap`` per material over its
    cells -- with the accelerated class: stress, tangent block and (default mode) the internal state variables of point i are
    written into row ``dofs[i]`` of each map's own Functions by ``integrate_rows`` (round 6: the state fields too, inside the same
    call); every Function equals what the reference class leaves, rows of the other map's cells untouched (zero)."""
    import ufl
    from dolfinx import fem, mesh
    from dolfinx_materials.quadrature_map import QuadratureMap as Reference
    from dolfinx_materials.utils import symmetric_tensor_to_vector
    from mpi4py import MPI

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from dolfinx_materials_amd.quadrature_map import QuadratureMap as Accelerated

    domain = mesh.create_unit_cube(MPI.COMM_WORLD, 4, 4, 4, mesh.CellType.hexahedron)
    V = fem.functionspace(domain, ("P", 1, (3,)))
    u = fem.Function(V)
    ncell = domain.topology.index_map(3).size_local
    cells_a = np.arange(0, ncell, 2, dtype=np.int32)
    cells_b = np.arange(1, ncell, 2, dtype=np.int32)
    strain = symmetric_tensor_to_vector(ufl.sym(ufl.grad(u)))
    beh = lambda s0: jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(s0, H))   # noqa: E731
    sets = {}
    for name, cls in (("reference", Reference), ("accelerated", Accelerated)):
        maps = [cls(domain, 2, JAXMaterial(beh(SIG0)), cells=cells_a), cls(domain, 2, JAXMaterial(beh(0.6 * SIG0)), cells=cells_b)]
        for q in maps:
            q.register_gradient("strain", strain)
        sets[name] = maps
    x = V.tabulate_dof_coordinates()
    rng = np.random.default_rng(4)
    for step in range(3):
        u.x.array[:] = (x @ ((3e-3 + 2e-3 * step) * rng.standard_normal((3, 3))).T).reshape(-1) + 2e-4 * np.sin(5.0 * x).reshape(-1)
        for maps in sets.values():
            for q in maps:
                q.update()
        for qr, qa in zip(sets["reference"], sets["accelerated"]):
            assert qa.material.delivers_state_outputs == frozenset(qa.material.internal_state_variables)
            assert np.array_equal(qr.fluxes["stress"].x.array, qa.fluxes["stress"].x.array)
            assert np.array_equal(qr.jacobian_flatten.x.array, qa.jacobian_flatten.x.array)
            for key in ("p", "epsp"):
                assert np.array_equal(qr.internal_state_variables[key].x.array, qa.internal_state_variables[key].x.array), key
        for maps in sets.values():
            for q in maps:
                q.advance()
    qa = sets["accelerated"][0]
    other = np.setdiff1d(np.arange(len(qa.fluxes["stress"].x.array) // 6), qa.dofs)
    assert not qa.fluxes["stress"].x.array.reshape(-1, 6)[other].any() and qa.fluxes["stress"].x.array.any()
    for q in sets["accelerated"]:
        q.close()
