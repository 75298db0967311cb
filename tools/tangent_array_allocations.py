"""Kernel time of the J2 update (1e7 points, device-resident) against WHERE THE CALLER'S ARRAYS SIT: after dxm_tune_placement has placed
the state, six further allocations of the tangent array, three of the flux array, three of the strain array, then the state search
once more against the fastest tangent array.

    python tools/tangent_array_allocations.py
"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from helpers import E, NU, SIG0_LIN, H_LIN, j2_history, to_device
n = 10_000_000
dev = torch.device("cuda:0")
h = j2_history(n)
eps = [to_device(x) for x in h[:3]]
flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
m.set_data_manager(n)
st = torch.cuda.current_stream().cuda_stream
def t(g, f, c, reps=20):
    for _ in range(3): m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
m.integrate_device(eps[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st); torch.cuda.synchronize(); m.data_manager.update()
before = t(eps[1], flux, ct)
info = m.tune_placement(eps[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), max_candidates=4)
after = t(eps[1], flux, ct)
out = {"untuned_ms": round(before, 4), "tuned_ms": round(after, 4), "ct_alternatives_ms": [], "flux_alternatives_ms": [], "eps_alternatives_ms": []}
keep = []
for k in range(6):
    c2 = torch.empty((n, 36), dtype=torch.float64, device=dev); keep.append(c2)
    out["ct_alternatives_ms"].append(round(t(eps[1], flux, c2), 4))
for k in range(3):
    f2 = torch.empty((n, 6), dtype=torch.float64, device=dev); keep.append(f2)
    out["flux_alternatives_ms"].append(round(t(eps[1], f2, ct), 4))
for k in range(3):
    e2 = eps[1].clone(); keep.append(e2)
    out["eps_alternatives_ms"].append(round(t(e2, flux, ct), 4))
best_ct = keep[int(np.argmin(out["ct_alternatives_ms"]))]
out["best_ct_then_retuned_state_ms"] = None
info2 = m.tune_placement(eps[1].data_ptr(), flux.data_ptr(), best_ct.data_ptr(), max_candidates=4)
out["best_ct_then_retuned_state_ms"] = round(t(eps[1], flux, best_ct), 4)
print(json.dumps(out))
