"""ctypes wrapper of the plain-C oracle (``oracle/oracle_c.c``).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle_dxmat.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        vp, i64, dbl, ci = C.c_void_p, C.c_int64, C.c_double, C.c_int
        lib.orc_elastic_iso.restype = None
        lib.orc_elastic_iso.argtypes = [i64, vp, dbl, dbl, vp, vp, ci]
        lib.orc_j2.restype = i64
        lib.orc_j2.argtypes = [i64, vp, vp, vp, dbl, dbl, ci, dbl, dbl, dbl, dbl, vp, vp, vp, vp, C.POINTER(i64), ci]
        lib.orc_fefp.restype = i64
        lib.orc_fefp.argtypes = [i64, vp, vp, vp, dbl, dbl, ci, dbl, dbl, dbl, dbl, vp, vp, vp, vp, vp, C.POINTER(i64), ci]
        lib.orc_max_threads.restype = ci
        _lib = lib
    return _lib


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def elastic_iso(eps, E, nu, nthreads=1):
    eps = _c(eps)
    n = eps.shape[0]
    sig = np.empty((n, 6))
    ct = np.empty((n, 6, 6))
    load().orc_elastic_iso(n, eps.ctypes.data, E, nu, sig.ctypes.data, ct.ctypes.data, nthreads)
    return sig, ct


def j2(eps, epsp_n, p_n, E, nu, kind, sig0, h1, h2=0.0, rtol=1e-14, nthreads=1, out=None):
    """kind 0: linear (h1 = H); kind 1: Voce (h1 = sigu, h2 = b)."""
    eps, epsp_n, p_n = _c(eps), _c(epsp_n), _c(p_n).reshape(-1)
    n = eps.shape[0]
    if out is None:
        out = dict(sig=np.empty((n, 6)), epsp=np.empty((n, 6)), p=np.empty(n), Ct=np.empty((n, 6, 6)))
    npl = C.c_int64(0)
    notconv = load().orc_j2(
        n, eps.ctypes.data, epsp_n.ctypes.data, p_n.ctypes.data, E, nu, kind, sig0, h1, h2, rtol,
        out["sig"].ctypes.data, out["epsp"].ctypes.data, out["p"].ctypes.data, out["Ct"].ctypes.data,
        C.byref(npl), nthreads,
    )
    out["n_plastic"] = npl.value
    out["n_not_converged"] = int(notconv)
    return out


def fefp(F9, cpinv_n, p_n, E, nu, sig0, sigu, b=0.0, rtol=1e-14, nthreads=1, out=None, tangent=True, kind=1):
    """kind 1: Voce (sigu, b); kind 0: linear hardening (sigu holds H)."""
    F9, cpinv_n, p_n = _c(F9), _c(cpinv_n), _c(p_n).reshape(-1)
    n = F9.shape[0]
    if out is None:
        out = dict(P=np.empty((n, 9)), be_bar=np.empty((n, 6)), cpinv=np.empty((n, 6)), p=np.empty(n), Ct=np.empty((n, 9, 9)))
    npl = C.c_int64(0)
    notconv = load().orc_fefp(
        n, F9.ctypes.data, cpinv_n.ctypes.data, p_n.ctypes.data, E, nu, kind, sig0, sigu, b, rtol,
        out["P"].ctypes.data, out["be_bar"].ctypes.data, out["cpinv"].ctypes.data, out["p"].ctypes.data,
        out["Ct"].ctypes.data if tangent else None, C.byref(npl), nthreads,
    )
    out["n_plastic"] = npl.value
    out["n_not_converged"] = int(notconv)
    return out


def max_threads():
    return load().orc_max_threads()
