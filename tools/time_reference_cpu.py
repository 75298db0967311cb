#!/usr/bin/env python3
"""Times the reference's own CPU path for BASELINE.json configs[0] in THIS container (the Python
reference cannot travel to the GPU box): LinearElasticIsotropic through generic.Material.integrate
(python_materials/elasticity.py:21-24, generic.py:176-189 -> Python loop generic.py:77-79), 1e5 points.
Also times the same J2 law used for the protocol golden (a per-point Python law through the same
machinery) and the build's numpy / C oracles on the same inputs, for scale.

    python tools/time_reference_cpu.py > profiles/archive/r01_reference_cpu_container.json
"""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import constitutive_np as onp  # noqa: E402
from oracle import oracle_c  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402

warnings.simplefilter("ignore")
generic, pm = import_reference()
n = 100_000
rng = np.random.default_rng(0)
eps = 1e-3 * rng.standard_normal((n, 6))
mat = pm.LinearElasticIsotropic(70e3, 0.3)
mat.set_data_manager(n)
t = []
for _ in range(3):
    t0 = time.perf_counter()
    sig, isv, Ct = mat.integrate(eps)
    t.append(time.perf_counter() - t0)
ref_s = min(t)
t0 = time.perf_counter()
for _ in range(20):
    onp.elastic_iso(eps, 70e3, 0.3)
np_s = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    oracle_c.elastic_iso(eps, 70e3, 0.3, nthreads=1)
c_s = (time.perf_counter() - t0) / 20
print(json.dumps({
    "where": "build container (8-core Xeon 2.1 GHz, no GPU); the Python reference never travels to the GPU box",
    "config": "BASELINE.json configs[0]: LinearElasticIsotropic(70e3, 0.3), 1e5 Gauss points, eps = 1e-3 N(0,1), rng(0)",
    "reference_generic_material_python_loop": {"seconds_per_integrate": round(ref_s, 3), "Mpoints_per_s": round(n / ref_s / 1e6, 4), "cores": 1},
    "oracle_numpy_vectorised": {"seconds": round(np_s, 5), "Mpoints_per_s": round(n / np_s / 1e6, 2), "cores": 1},
    "oracle_c_port": {"seconds": round(c_s, 5), "Mpoints_per_s": round(n / c_s / 1e6, 2), "cores": 1},
    "max_abs_diff_reference_vs_oracle": float(np.abs(np.asarray(sig) - onp.elastic_iso(eps, 70e3, 0.3)[0]).max()),
}, indent=1))
