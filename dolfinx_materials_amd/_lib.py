"""ctypes binding of ``libdxmat.so`` (C ABI declared in ``include/dxmat.h``).

There is no fallback: if the shared library is missing or no HIP device is usable the
constructors raise.  ``build()`` compiles the library in-tree with ``hipcc --offload-arch=gfx950``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DXM_LIB_PATH") or os.path.join(_HERE, "libdxmat.so")  # override: A/B of two builds
CSRC_DIR = os.path.join(_HERE, "csrc")

DXM_MAX_STATE_FIELDS = 4
LAW_ELASTIC_ISO, LAW_J2_LINEAR, LAW_J2_VOCE, LAW_FEFP_J2_VOCE, LAW_FEFP_J2_LINEAR = 0, 1, 2, 3, 4
S0, S1 = 0, 1


class DxmError(RuntimeError):
    """Hard error reported by libdxmat (negative return code)."""


class LawInfo(C.Structure):
    _fields_ = [
        ("n_grad", C.c_int32),
        ("n_flux", C.c_int32),
        ("n_params", C.c_int32),
        ("n_isv_fields", C.c_int32),
        ("isv_dim", C.c_int32 * DXM_MAX_STATE_FIELDS),
        ("isv_name", C.c_char_p * DXM_MAX_STATE_FIELDS),
        ("n_isv_total", C.c_int32),
        ("algorithmic_bytes_per_point", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_points", C.c_int64),
        ("n_plastic", C.c_int64),
        ("n_not_converged", C.c_int64),
        ("n_nan", C.c_int64),
        ("max_local_iters", C.c_int32),
        ("upload", C.c_int32),
    ]

    #: dxm_stats.upload: how the gradient array reached the GPU in a host-buffer call (include/dxmat.h DXM_UPLOAD_*)
    UPLOAD = {0: None, 1: "dma (caller's array page-locked)", 2: "dma (page-locked for the call)", 3: "staged through the ring", 4: "runtime pageable path"}

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "upload"}

    @property
    def upload_mode(self):
        return self.UPLOAD.get(int(self.upload))


_dp = C.POINTER(C.c_double)
_h = C.c_void_p

#: every symbol include/dxmat.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dxm_abi_version": (C.c_int, []),
    "dxm_has_custom_hardening": (C.c_int, []),
    "dxm_last_error": (C.c_char_p, []),
    "dxm_device_count": (C.c_int, []),
    "dxm_law_info_get": (C.c_int, [C.c_int, C.POINTER(LawInfo)]),
    "dxm_create": (_h, [C.c_int, _dp, C.c_int, C.c_int64, C.c_int]),
    "dxm_destroy": (C.c_int, [_h]),
    "dxm_npoints": (C.c_int64, [_h]),
    "dxm_law": (C.c_int, [_h]),
    "dxm_set_params": (C.c_int, [_h, _dp, C.c_int]),
    "dxm_set_newton": (C.c_int, [_h, C.c_int, C.c_double]),
    "dxm_set_tangent_layout": (C.c_int, [_h, C.c_int]),
    "dxm_tangent_size": (C.c_int, [_h]),
    "dxm_set_state": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "dxm_get_state": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "dxm_advance": (C.c_int, [_h]),
    "dxm_revert": (C.c_int, [_h]),
    "dxm_io_held": (C.c_int, [_h, C.c_int]),
    "dxm_get_io": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "dxm_integrate_rows": (C.c_int, [_h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "dxm_integrate": (
        C.c_int,
        [_h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)],
    ),
    "dxm_integrate_device": (
        C.c_int,
        [_h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "dxm_get_stats": (C.c_int, [_h, C.POINTER(Stats)]),
    "dxm_isv_device": (C.c_int, [_h, C.c_int, C.c_void_p, C.c_void_p]),
    "dxm_state_ptr": (C.c_void_p, [_h, C.c_int, C.c_int, C.c_int]),
    "dxm_integrate_displacement_device": (C.c_int, [_h, _h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dxm_expand_tangent_device": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "dxm_expand_tangent_pack4_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "dxm_kernel_name": (C.c_char_p, [_h]),
    "dxm_launch_generation": (C.c_uint64, [_h]),
    "dxm_notify_replay": (C.c_int, [_h]),
    "dxm_set_option": (C.c_int, [_h, C.c_char_p, C.c_double]),
    "dxm_isv_host": (C.c_int, [_h, C.c_int, C.c_void_p]),
    "dxm_bind_isv_output": (C.c_int, [_h, C.c_int, C.c_void_p]),
    "dxm_host_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]),
    "dxm_host_scatter_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "dxm_host_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "dxm_host_index_range": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dxm_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "dxm_host_unregister": (C.c_int, [C.c_void_p]),
    "dxm_host_alloc": (C.c_void_p, [C.c_uint64]),
    "dxm_host_free": (C.c_int, [C.c_void_p]),
    "dxm_mesh_create_hex8": (_h, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int]),
    "dxm_mesh_create_tet4": (_h, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "dxm_mesh_create_simplex": (
        _h,
        [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int],
    ),
    "dxm_mesh_destroy": (C.c_int, [_h]),
    "dxm_mesh_npoints": (C.c_int64, [_h]),
    "dxm_mesh_displacement_size": (C.c_int64, [_h]),
    "dxm_mesh_gradient_device": (C.c_int, [_h, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dxm_integrate_displacement": (
        C.c_int,
        [_h, _h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)],
    ),
    "dxm_integrate_displacement_rows": (
        C.c_int,
        [_h, _h, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)],
    ),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile ``libdxmat.so`` for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC_DIR], check=True)
    return LIB_PATH


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` / ``libhsa-runtime64.so``.  If
    libdxmat (linked against ``/opt/rocm``) initialises HIP first and torch is imported later, the
    process ends up with two HIP/HSA runtimes and torch reports "No HIP GPUs are available".
    Importing torch first puts its runtime in the global symbol scope, libdxmat binds to it, and
    device pointers, streams and events are shared (this is what ``bench.py`` relies on when it
    passes ``torch.cuda.current_stream().cuda_stream``).  Skipped when torch is not installed or
    ``DXM_NO_TORCH_PRELOAD=1``."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("DXM_NO_TORCH_PRELOAD") == "1":
        return
    if importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def load() -> C.CDLL:
    """Load the library and bind every ABI symbol; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DxmError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `make -C dolfinx_materials_amd/csrc`). "
            "dolfinx_materials_amd has no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _bind(lib: C.CDLL, strict: bool = True) -> C.CDLL:
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        except AttributeError:
            if strict:
                raise
            continue  # tools/ab_inproc.py: older builds of the library beside the current one
        fn.restype = res
        fn.argtypes = args
    return lib


_custom_libs = {}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def jit_cache_dir() -> str:
    """Where the JIT-compiled copies of the library live: ``$DXM_JIT_CACHE``, else
    ``$XDG_CACHE_HOME/dolfinx_materials_amd/jit`` (``~/.cache/...``), else a per-user directory under the
    system's temporary directory when the home directory is not writable.  Never inside the package."""
    import tempfile

    cand = []
    if os.environ.get("DXM_JIT_CACHE"):
        cand.append(os.environ["DXM_JIT_CACHE"])
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    cand.append(os.path.join(base, "dolfinx_materials_amd", "jit"))
    cand.append(os.path.join(tempfile.gettempdir(), f"dolfinx_materials_amd_jit_{os.getuid()}"))
    for d in cand:
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK | os.X_OK):
                return d
        except OSError:
            continue
    raise DxmError(f"no writable directory for the JIT cache among {cand}")


JIT_CACHE_MAX = int(os.environ.get("DXM_JIT_CACHE_MAX", "32"))


def _evict_jit_cache(root: str, keep: str) -> None:
    """Least-recently-used eviction: at most ``JIT_CACHE_MAX`` compiled laws stay in the cache (``keep`` always does).
    A library another process has loaded stays mapped after its file is unlinked."""
    import shutil

    try:
        entries = []
        for name in os.listdir(root):
            lib = os.path.join(root, name, "libdxmat_custom.so")
            if os.path.isfile(lib):
                entries.append((os.stat(lib).st_mtime, os.path.join(root, name)))
        entries.sort(reverse=True)
        for _, d in entries[JIT_CACHE_MAX:]:
            if os.path.abspath(d) != os.path.abspath(keep):
                shutil.rmtree(d, ignore_errors=True)
    except OSError:
        pass   # housekeeping only


def load_custom(expr_R: str, expr_dR: str) -> C.CDLL:
    """Build (once, cached under :func:`jit_cache_dir`) and load a copy of libdxmat whose
    "voce" kernels integrate a user-supplied isotropic hardening law: ``expr_R`` / ``expr_dR`` are C
    expressions for R(p) and dR/dp in the variables ``p``, ``sig0`` and ``c[0..5]``.  This is the
    counterpart of handing a Python ``yield_stress(p)`` callable to jaxmat and letting ``jax.jit``
    compile it on the first pass (``tests/test_FeFp_jax.py:14-19``, ``jaxmat.py:214-216``): here
    hipcc compiles the fused gfx950 kernels with the law inlined (a few seconds).  The cache key is the
    law together with the kernel sources; the least recently used entries beyond ``DXM_JIT_CACHE_MAX``
    (32) are evicted."""
    import hashlib

    key = (expr_R, expr_dR)
    if key in _custom_libs:
        return _custom_libs[key]
    srcs = [os.path.join(CSRC_DIR, f) for f in sorted(os.listdir(CSRC_DIR)) if f.endswith((".hip", ".hpp"))]
    srcs += [os.path.join(os.path.dirname(_HERE), "include", h) for h in ("dxmat.h",)]
    h = hashlib.sha1()
    h.update(expr_R.encode() + b"\0" + expr_dR.encode())
    for f in srcs:
        h.update(open(f, "rb").read())
    root = jit_cache_dir()
    out_dir = os.path.join(root, h.hexdigest()[:16])
    out = os.path.join(out_dir, "libdxmat_custom.so")
    if not os.path.exists(out):
        os.makedirs(out_dir, exist_ok=True)
        cmd = [
            HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-shared",
            "-DDXM_CUSTOM_HARDENING", f"-DDXM_CUSTOM_R={expr_R}", f"-DDXM_CUSTOM_DR={expr_dR}",
            "-o", None, os.path.join(CSRC_DIR, "dxmat.hip"),
        ]
        # every rank of a multi-process run may hit the cold cache at once: each compiles into its own
        # temporary file and publishes it with an atomic rename (identical content, last one wins)
        import tempfile

        fd, tmp = tempfile.mkstemp(prefix="libdxmat_custom.", suffix=f".{os.getpid()}.tmp", dir=out_dir)
        os.close(fd)
        cmd[cmd.index(None)] = tmp
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            try:
                os.remove(tmp)
            except OSError:
                pass
            raise DxmError(f"compiling the custom hardening law failed:\n{r.stderr[-2000:]}")
        os.replace(tmp, out)
        _evict_jit_cache(root, out_dir)
    else:
        try:
            os.utime(out)   # recently used
        except OSError:
            pass
    _share_hip_runtime_with_torch()
    lib = _bind(C.CDLL(out))
    assert lib.dxm_has_custom_hardening() == 1
    _custom_libs[key] = lib
    return lib


def last_error(lib=None) -> str:
    return ((lib or load()).dxm_last_error() or b"").decode()


def check(rc: int, lib=None) -> int:
    """Raise on hard errors (<0); pass soft codes (>=0) through."""
    if rc < 0:
        raise DxmError(f"libdxmat error {rc}: {last_error(lib)}")
    return rc


def device_count() -> int:
    n = load().dxm_device_count()
    return max(n, 0)


def _free_pinned(lib, ptr):
    lib.dxm_host_free(ptr)


class PinnedArray:
    """A C-contiguous fp64 numpy array in page-locked host memory (``dxm_host_alloc``).

    The memory belongs to the ARRAY, not to this object: numpy views keep the underlying ctypes
    buffer alive, and the page-locked block is released by a finalizer on that buffer, i.e. when
    the last view of it is gone.  ``release()`` only drops this object's own reference, so arrays
    already handed to a caller (``integrate`` results, state dicts) never dangle."""

    def __init__(self, shape):
        import weakref

        import numpy as np

        self.shape = tuple(int(s) for s in shape)
        n = 1
        for s in self.shape:
            n *= s
        lib = load()
        ptr = lib.dxm_host_alloc(8 * max(n, 1))
        if not ptr:
            raise DxmError(f"dxm_host_alloc failed: {last_error()}")
        buf = (C.c_double * max(n, 1)).from_address(ptr)
        # atexit=False: at interpreter exit the OS reclaims the block; no HIP calls during teardown
        weakref.finalize(buf, _free_pinned, lib, ptr).atexit = False
        self.ptr = ptr
        self.array = np.frombuffer(buf, dtype=np.float64, count=n).reshape(self.shape)

    def release(self):
        self.array = None

    free = release


def law_info(law: int, lib=None) -> LawInfo:
    info = LawInfo()
    lib = lib or load()
    check(lib.dxm_law_info_get(law, C.byref(info)), lib)
    return info
