#!/usr/bin/env python3
"""Soak of the host-buffer form (round 6: three-stream scheme, row deliveries, sym rebuild): for ragged batch sizes, every tangent
layout and both ISV modes, `reps` calls with the shipped options must deliver, bit for bit, what ONE call with whole chunks
alternating on two streams (`split_streams = 0`) delivered -- a missing dependency between the upload / kernel stream, the two
download streams and the worker threads would show up as a differing byte sooner or later.  One JSON line per configuration."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(reps=25):
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_V, SIGU_V, B_V, j2_history

    beh = lambda: jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_V, SIGU_V, B_V))   # noqa: E731
    bad = 0
    for n in (300_001, 1_000_003, 4_999_999):
        h = j2_history(n, seed=n % 97, sig0=SIG0_V)
        for layout in ("full", "pack4", "sym"):
            for rows_mode in (False, True):
                width = {"full": 36, "pack4": 4, "sym": 21}[layout]
                total = n + 1001 if rows_mode else n
                rows = np.ascontiguousarray(np.random.default_rng(n).permutation(total)[:n], dtype=np.int64) if rows_mode else None
                res = []
                t0 = time.perf_counter()
                for split in (0, 1):
                    m = JAXMaterial(beh(), tangent_layout=layout)
                    m.set_data_manager(n)
                    m.set_option("split_streams", split)
                    flux, jac, grad = np.full(total * 6, np.nan), np.full(total * width, np.nan), np.zeros(n * 6)
                    fields = {"p": np.full(total, np.nan), "epsp": np.full(total * 6, np.nan)}
                    if rows_mode:
                        m.bind_state_outputs(fields, deliver=True, rows=True)
                    else:
                        m.bind_outputs(flux=flux, tangent=jac)
                        m.bind_state_outputs(fields, deliver=True)
                    m.bind_inputs(gradient=grad)
                    g = grad.reshape(n, 6)
                    g[...] = h[1]
                    call = (lambda: m.integrate_rows(g, rows, flux, jac)) if rows_mode else (lambda: m.integrate(g))
                    call()
                    m.data_manager.update()
                    g[...] = h[2]
                    call()
                    first = [a.copy() for a in (flux, jac, fields["p"], fields["epsp"])]
                    same = True
                    for _ in range(reps if split else 2):
                        flux[...] = np.nan
                        jac[...] = np.nan
                        fields["p"][...] = np.nan
                        fields["epsp"][...] = np.nan
                        call()
                        same = same and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(first, (flux, jac, fields["p"], fields["epsp"])))
                    res.append((first, same))
                    m.close()
                across = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res[0][0], res[1][0]))
                ok = bool(res[0][1] and res[1][1] and across)
                bad += not ok
                print(json.dumps({"points": n, "layout": layout, "rows_form": rows_mode, "calls_with_split_streams": reps, "every_call_identical": bool(res[1][1]),
                                  "identical_to_alternating_chunks": bool(across), "ok": ok, "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    print(json.dumps({"configurations_failed": bad}))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 25) else 0)
