"""The GPU-free host code of libdxmat.so (``dolfinx_materials_amd/csrc/host_side.hpp``: worker pool, chunk planner + staging ring,
page-locked range table, the bit-exact tangent rebuilds, threaded row moves, upload-route state machine) under ThreadSanitizer
and AddressSanitizer + UBSan, on the CPU box: ``tests/host_side_harness.cpp`` is compiled twice with clang++ and run on
seeded inputs; its rebuilt tangent blocks are compared with ``oracle/host_rebuild_np.py`` (0 ulp against the fused-multiply-add
emulation on a sample, 2 ulp-of-the-block against plain numpy on everything).  Also: the Python array reaper of
``lazy_rows.py`` under ``faulthandler`` with 10^4 drops."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import host_rebuild_np as hr  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(ROOT, "tests", "host_side_harness.cpp")
HDR = os.path.join(ROOT, "dolfinx_materials_amd", "csrc", "host_side.hpp")
OUT = os.path.join(ROOT, "tests", "_san")
BUILDS = {
    "tsan": ["-fsanitize=thread"],
    "asan_ubsan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"],
}


def _compiler():
    for c in (CLANG, shutil.which("clang++"), shutil.which("g++")):
        if c and os.path.exists(c):
            return c
    return None


def _build(kind):
    cc = _compiler()
    if cc is None:
        pytest.skip("no C++ compiler with sanitizer runtimes")
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, f"harness_{kind}")
    stale = not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(SRC), os.path.getmtime(HDR))
    if stale:
        # the product's host flags that matter for the rebuilt values: contraction allowed where the source does not forbid it
        cmd = [cc, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=fast", "-pthread", *BUILDS[kind], SRC, "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _inputs(n, seed=7):
    rng = np.random.default_rng(seed)
    M = 2 * n
    mu = 70e3 / 2.6
    coef = np.concatenate([rng.normal(4e4, 1e3, (n, 1)), rng.normal(5e4, 1e3, (n, 1)), -rng.uniform(0, 4e4, (n, 1)), rng.normal(0, 0.7, (n, 6))], axis=1)
    sg = rng.normal(0, 300.0, (n, 6))
    cw = np.concatenate([coef[:, :3], rng.uniform(1e-3, 1e-2, (n, 1))], axis=1)
    rec = rng.normal(0, 1.0, (n, 54)) * np.repeat([1.0, mu, mu, 1.0, mu, mu], 9)[None, :]
    pk = rng.normal(0, 300.0, (n, 9))
    lm = np.array([70e3 * 0.3 / 1.3 / 0.4, mu])
    rows = rng.permutation(M)[:n].astype(np.int64)
    return dict(n=n, M=M, coef=coef, sg=sg, cw=cw, rec=rec, pk=pk, lm=lm, rows=rows)


def _write(path, d):
    with open(path, "wb") as f:
        np.array([d["n"], d["M"]], dtype=np.int64).tofile(f)
        for k in ("coef", "sg", "cw", "rec", "pk", "lm"):
            np.ascontiguousarray(d[k], dtype=np.float64).tofile(f)
        d["rows"].tofile(f)


@pytest.mark.parametrize("kind", list(BUILDS))
def test_host_side_under_sanitizers(kind, tmp_path):
    exe = _build(kind)
    n = 70_001   # ragged: not a multiple of 256; large enough for 2+ chunks of the packed form
    d = _inputs(n)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _write(fin, d)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1 abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1 halt_on_error=1")
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=900, env=env)
    report = (r.stdout + r.stderr)[-6000:]
    assert r.returncode == 0 and "all ok" in r.stdout, report
    for marker in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer"):
        assert marker not in r.stderr, report
    out = np.fromfile(fout, dtype=np.float64)
    ct_coef, ct_pack4, ct_fefp, ct_const = np.split(out, np.cumsum([n * 36, n * 36, n * 81]))
    ct_coef, ct_pack4, ct_fefp, ct_const = ct_coef.reshape(n, 36), ct_pack4.reshape(n, 36), ct_fefp.reshape(n, 81), ct_const.reshape(n, 36)
    # everything against plain numpy: a fused and an unfused evaluation differ by at most a rounding of the largest term
    for got, exp in ((ct_coef, hr.coef_np(d["coef"])), (ct_pack4, hr.pack4_np(d["sg"], d["cw"])), (ct_fefp, hr.fefp_np(d["rec"]))):
        scale = np.abs(exp).max(axis=1, keepdims=True)
        assert np.all(np.abs(got - exp) <= 4 * np.finfo(float).eps * scale)
    assert np.array_equal(ct_const, hr.const_np(d["lm"][0], d["lm"][1], n))
    # a sample against the exact emulation of the product's operation order: 0 ulp
    idx = np.r_[0, 1, 255, 256, n - 1, np.random.default_rng(3).integers(0, n, 40)]
    assert np.array_equal(ct_coef[idx], hr.coef_exact(d["coef"][idx]))
    assert np.array_equal(ct_pack4[idx], hr.pack4_exact(d["sg"][idx], d["cw"][idx]))
    assert np.array_equal(ct_fefp[idx], hr.fefp_exact(d["rec"][idx]))


def test_product_library_exports_the_same_host_code():
    """libdxmat.so is built from the same header: the ABI entry points that are pure host code (threaded copy, row scatter /
    gather) agree with numpy, without a GPU."""
    from dolfinx_materials_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(1)
    n, w = 40_000, 36
    rows = rng.permutation(2 * n)[:n].astype(np.int64)
    src = rng.standard_normal((n, w))
    dst = np.full((2 * n, w), -1.0)
    assert lib.dxm_host_scatter_rows(dst.ctypes.data, src.ctypes.data, rows.ctypes.data, n, w, 16) == 0
    ref = np.full((2 * n, w), -1.0)
    ref[rows] = src
    assert np.array_equal(dst, ref)
    back = np.empty_like(src)
    assert lib.dxm_host_gather_rows(back.ctypes.data, dst.ctypes.data, rows.ctypes.data, n, w, 16) == 0
    assert np.array_equal(back, src)
    a, b = rng.standard_normal(3_000_001), np.zeros(3_000_001)
    assert lib.dxm_host_copy(b.ctypes.data, a.ctypes.data, a.nbytes, 8) == 0
    assert np.array_equal(a, b)


def test_reaper_survives_ten_thousand_drops():
    """``lazy_rows._Reaper``: 10^4 large arrays handed to the helper thread under ``faulthandler`` in a child interpreter; the
    queue drains, nothing crashes, and small arrays are left to the caller."""
    code = r"""
import faulthandler, sys, time, threading
faulthandler.enable()
faulthandler.dump_traceback_later(120, exit=True)
import numpy as np
sys.path.insert(0, %r)
from dolfinx_materials_amd.lazy_rows import _Reaper
r = _Reaper()
small = np.empty(10)
r.drop(small); r.drop(None)
assert r._q is None, "small arrays must not start the thread"
for k in range(10_000):
    a = np.empty(1 << 20)          # 8 MiB, untouched pages
    if k %% 97 == 0:
        a[::4096] = 1.0              # some are really resident
    r.drop(a)
    del a
t0 = time.time()
while not r._q.empty() and time.time() - t0 < 60:
    time.sleep(0.01)
assert r._q.empty(), "reaper did not drain"
assert any(t.name == "dxm-array-reaper" and t.daemon for t in threading.enumerate())
print("reaper ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "reaper ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
