import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from helpers import E, NU, SIG0_LIN, H_LIN, j2_history
n = 10_000_000
h = j2_history(n)
for mode in ("bound", "coef"):
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)), tangent_layout="coef" if mode == "coef" else "full")
    m.set_data_manager(n)
    if mode == "bound":
        f, j = np.zeros(n * 6), np.zeros(n * 36)
        m.bind_outputs(flux=f, tangent=j)
    m.integrate(h[1]); m.data_manager.update(); m.integrate(h[2]); m.integrate(h[2])
    m.set_option("tune_verbose", 1)
    print("mode", mode, flush=True)
    t0 = time.perf_counter(); m.integrate(h[2]); print("call ms", (time.perf_counter() - t0) * 1e3, flush=True)
    m.close()
