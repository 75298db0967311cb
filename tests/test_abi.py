"""CPU tests of the drop-in boundary: libdxmat.so loads, exports every symbol include/dxmat.h
declares, reports the law table, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dolfinx_materials_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="dxmat.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dxm_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_declare_the_same_symbols():
    syms = header_symbols()
    assert len(syms) >= 20
    assert syms == sorted(_lib.SYMBOLS)
    # one header, one contract: the "experimental" side door of ABI 3-5 (placement search, launch timing, assembly kernels) is gone
    headers = [f for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith((".h", ".hpp"))]   # (stray editor files do not count)
    assert headers == ["dxmat.h"] and not hasattr(_lib, "EXPERIMENTAL_SYMBOLS")


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.dxm_abi_version() == 6


def test_library_exports_the_header_and_nothing_else():
    """`nm -D libdxmat.so` = the list of include/dxmat.h: no assembly kernels (FEM
    assembly stays on the host; the stand-in loop's operators are examples/libdxmfem.so), no measurement helpers."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "dolfinx_materials_amd", "libdxmat.so")], capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r"\sT\s+dxm_", ln)})
    assert exported == header_symbols()
    for gone in ("dxm_tune_placement", "dxm_time_device", "dxm_mesh_internal_force_device", "dxm_mesh_tangent_apply_device",
                 "dxm_mesh_tangent_diagonal_device", "dxm_mesh_set_weights"):
        assert gone not in exported
    # nothing else with C linkage slips out either: every defined text symbol is a dxm_* entry point or C++-mangled / runtime glue
    other = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r"\sT\s+(?!dxm_|_Z|_init|_fini|__)", ln)})
    assert other == [], other


def test_law_table():
    expect = {
        _lib.LAW_ELASTIC_ISO: (6, 6, 2, {}, 384),
        _lib.LAW_J2_LINEAR: (6, 6, 4, {"p": 1, "epsp": 6}, 496),
        _lib.LAW_J2_VOCE: (6, 6, 5, {"p": 1, "epsp": 6}, 496),
        _lib.LAW_FEFP_J2_VOCE: (9, 9, 5, {"p": 1, "be_bar": 6}, 976),
        _lib.LAW_FEFP_J2_LINEAR: (9, 9, 4, {"p": 1, "be_bar": 6}, 976),
    }
    for law, (ng, nf, npar, isv, alg) in expect.items():
        i = _lib.law_info(law)
        assert (i.n_grad, i.n_flux, i.n_params, i.algorithmic_bytes_per_point) == (ng, nf, npar, alg)
        got = {i.isv_name[f].decode(): i.isv_dim[f] for f in range(i.n_isv_fields)}
        assert got == isv and i.n_isv_total == sum(isv.values())
    with pytest.raises(_lib.DxmError):
        _lib.law_info(99)


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Stats) == 40
    assert C.sizeof(_lib.LawInfo) == 16 + 16 + 32 + 8


def test_protocol_surface_without_gpu():
    """Names/sizes the QuadratureMap consumes exist before any device is touched
    (quadrature_map.py:84-117)."""
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from dolfinx_materials_amd.python_materials import LinearElasticIsotropic

    el = jm.LinearElasticIsotropic(E=70e3, nu=0.3)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.VoceHardening(350.0, 500.0, 1e3)))
    assert m.gradients == {"strain": 6} and m.fluxes == {"stress": 6}
    assert m.internal_state_variables == {"p": 1, "epsp": 6}
    assert m.tangent_blocks == {("stress", "strain"): (6, 6)}
    assert m.variables == {"strain": 6, "stress": 6, "p": 1, "epsp": 6}
    assert m.rotation_matrix is None and m.name == "vonMisesIsotropicHardening"
    assert m.material_properties["yield_stress.sigu"] == 500.0
    f = JAXMaterial(jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0)))
    assert f.gradients == {"F": 9} and f.fluxes == {"PK1": 9}
    assert f.internal_state_variables == {"p": 1, "be_bar": 6}
    e = LinearElasticIsotropic(70e3, 0.3)
    assert e.gradients == {"Strain": 6} and e.fluxes == {"Stress": 6} and e.internal_state_variables == {}
    traced = jm.vonMisesIsotropicHardening(el, lambda p: 250.0 + 1e3 * p)  # a Python callable is traced (tests/test_FeFp_jax.py:14-19)
    assert traced.custom_hardening.expr_dR == "(1000.0)" and traced.params()[2] == 250.0
    with pytest.raises(TypeError):
        jm.vonMisesIsotropicHardening(el, 250.0)
    assert JAXMaterial(jm.ElasticBehavior(el), True).jit is True  # JAXMaterial(behavior, jit=True): jaxmat.py:144
    with pytest.raises(TypeError):
        JAXMaterial(jm.ElasticBehavior(el), 1)  # a GPU index must be passed as device=
    with pytest.raises(NotImplementedError, match="external state variable"):
        m.initialize_external_state_variable("Temperature", np.zeros(4))
    with pytest.raises(_lib.DxmError):
        m.integrate(np.zeros((4, 6)))  # set_data_manager not called


def test_no_cpu_fallback():
    """Without a usable HIP device the product path must raise, never compute on the host."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    m = JAXMaterial(jm.ElasticBehavior(jm.LinearElasticIsotropic(E=1.0, nu=0.2)))
    with pytest.raises(_lib.DxmError, match="no usable HIP device"):
        m.set_data_manager(8)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "dolfinx_materials_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle_c" not in txt and "liboracle" not in txt, f


def test_header_is_plain_c99_and_the_c_host_example_links():
    """The boundary is a C ABI: include/dxmat.h must compile as strict C99, and a host written in plain C
    (examples/c_host/j2_batch.c, no Python, no torch) must link against libdxmat.so."""
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "dxmat.h"\nint main(void) { return dxm_abi_version() == DXM_ABI_VERSION ? 0 : 1; }\n')
        subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-c", src,
                        "-o", os.path.join(d, "t.o")], check=True)
    subprocess.run(["make", "-C", os.path.join(root, "examples", "c_host"), "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", os.path.join(root, "examples", "c_host")], check=True, capture_output=True)
    assert os.path.exists(os.path.join(root, "examples", "c_host", "j2_batch"))


def test_host_copy_moves_every_byte_without_a_gpu():
    """`dxm_host_copy`: the multi-threaded snapshot copy the Python layer uses at `advance` (pure host code)."""
    import numpy as np

    from dolfinx_materials_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(0)
    for nbytes, threads in ((0, 4), (1000, 4), (5 << 20, 1), ((9 << 20) + 13, 8), ((6 << 20) + 1, 64)):
        src = rng.integers(0, 255, nbytes, dtype=np.uint8)
        dst = np.zeros(nbytes + 16, dtype=np.uint8)
        assert lib.dxm_host_copy(dst.ctypes.data + 8, src.ctypes.data, nbytes, threads) == 0
        assert np.array_equal(dst[8:8 + nbytes], src) and not dst[:8].any() and not dst[8 + nbytes:].any()
    assert lib.dxm_host_copy(None, None, 8, 2) < 0


def test_row_scatter_and_gather_match_numpy_indexing_without_a_gpu():
    """`dxm_host_scatter_rows` / `dxm_host_gather_rows`: `array[index] = values` and `array[index]` of `utils.py:136-143` /
    `quadrature_map.py:271` on several threads (what a QuadratureMap over a subset of the cells does per update)."""
    import numpy as np

    from dolfinx_materials_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(1)
    for n, total, width, threads in ((0, 10, 6, 4), (1, 3, 36, 4), (257, 1000, 1, 4), (20_000, 50_000, 36, 8), (70_001, 70_001, 6, 64), (9000, 20_000, 81, 0)):
        rows = np.ascontiguousarray(rng.permutation(total)[:n], dtype=np.int64)
        src = rng.standard_normal((n, width))
        dst = rng.standard_normal((total, width))
        want = dst.copy()
        want[rows] = src
        assert lib.dxm_host_scatter_rows(dst.ctypes.data, src.ctypes.data, rows.ctypes.data, n, width, threads) == 0
        assert np.array_equal(dst, want)
        out = np.full((n, width), np.nan)
        assert lib.dxm_host_gather_rows(out.ctypes.data, dst.ctypes.data, rows.ctypes.data, n, width, threads) == 0
        assert np.array_equal(out, dst[rows])
    assert lib.dxm_host_scatter_rows(None, None, None, 5, 6, 2) < 0
