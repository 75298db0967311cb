#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer form (what QuadratureMap.update() hands over: numpy
arrays in, numpy arrays out).  Reported in DESIGN.md; never the headline `value`."""
import argparse
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[100_000, 1_000_000, 10_000_000])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    for n in a.points:
        h = j2_history(n)
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.integrate(h[1])
        m.data_manager.update()
        m.integrate(h[2])
        t0 = time.perf_counter()
        for _ in range(a.reps):
            m.integrate(h[2])
        dt = (time.perf_counter() - t0) / a.reps
        print(json.dumps({"law": "j2_linear", "points": n, "host_path_ms": round(dt * 1e3, 3), "Mpoints_per_s": round(n / dt / 1e6, 2),
                          "pcie_bytes_per_point": 48 + 48 + 56 + 288, "GBs_over_pcie": round(n * 440 / dt / 1e9, 2)}), flush=True)
        m.close()


if __name__ == "__main__":
    main()
