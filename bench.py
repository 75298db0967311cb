#!/usr/bin/env python3
"""Headline benchmark: M quadrature-point updates/s (stress + consistent tangent, fp64).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md section 8(d) cfg 2): small-strain J2 plasticity with
linear isotropic hardening (E=70e3, nu=0.3, sig0=250, H=5e3), 1e7 Gauss points PER GPU (weak
scaling: the points are independent, every rank owns its own block and its state stays resident
on its GPU), 4-increment load/unload strain history seeded by default_rng(1234 + rank).
One "step" = one ``integrate`` over the whole batch (strain and old state read, stress, new
state and 6x6 tangent written), cycling through increments 2, 3, 4 of the history, each from the
converged state of the previous increment -- the cadence of QuadratureMap.update() inside a SNES
solve (reference solvers.py:72, quadrature_map.py:297-334).  Inputs and outputs are device
resident in the AoS layout of the dolfinx quadrature Functions.

Setup (untimed): the three load-step contexts are built by integrating the earlier increments; every array stays where its
FIRST allocation put it (the library has no placement search: profiles/r05_placement_decision.md).  Then ~0.5 s of back-to-back launches (the box leaves
its idle state), the W warm-up steps, and the K timed steps with a side thread sampling the GPU's sysfs telemetry (sclk, mclk,
fclk, power, busy) every 5 ms.  `value`, `roofline.frac` and `roofline.kernel_ms` are of THAT configuration.  Afterwards, as
context: the same kernel interleaved with two arithmetic-free streaming probes of its traffic mix on the same box
(`roofline.frac_of_stream_probe`).

The JSON line also carries
  roofline      achieved algorithmic HBM GB/s of the constitutive kernel (496 B/point x points
                per launch / mean launch duration from HIP events on the launch stream);
  box           what the box looked like from its own side before, during and after the timed steps (tools/box_telemetry.py);
  cpu_baseline  the plain-C oracle ("port") timed on this box's host cores on a bounded sample;
  gather_inclusive  (N > 1) the same steps followed by an RCCL all-gather of stress and tangent;
  other_laws    (N = 1) kernel rates of the elastic, J2-Voce and FeFp laws at the same batch size, each with its own
                `frac_of_stream_probe` (the arithmetic-free kernel with that law's streams on the same arrays);
  host_path     (N = 1, context, never `value`) the PCIe-inclusive drop-in form: `integrate` on host arrays, one
                `QuadratureMap.update()` of the accelerated map against the reference's cadence (`accelerated_update`,
                `as_reference_update`), and the same map over a packed tangent Function (`accelerated_update_packed`: pack4 / sym,
                also side by side with the full-layout map).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

E, NU, SIG0, H = 70e3, 0.3, 250.0, 5e3
ALG_BYTES = 496  # read eps 6 + eps_p 6 + p 1, write sig 6 + eps_p 6 + p 1 + Ct 36
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def history(n, seed, sig0=None):
    """eps_k = (k/3) eps_hat for k = 1..3, then unloading to 0.5 eps_hat (SURVEY 8(d) cfg 2)."""
    rng = np.random.default_rng(seed)
    mu = E / 2 / (1 + NU)
    epsy = (SIG0 if sig0 is None else sig0) / (2 * mu) * np.sqrt(2.0 / 3.0)
    d = rng.standard_normal((n, 6))
    d /= np.linalg.norm(d, axis=1)[:, None]
    eps_hat = d * (rng.uniform(0.0, 4.0, n) * epsy)[:, None]
    return [eps_hat / 3.0, eps_hat * (2.0 / 3.0), eps_hat, 0.5 * eps_hat]


def cpu_baseline_scan(sample, seed, budget_s):
    """Plain-C oracle ("port") on the host cores, same workload on a bounded sample: thread counts scanned twice, then two
    longer runs at the best count.  Runs in a child process of its own (`bench.py --cpu-baseline-child`) so that the OpenMP
    environment -- thread binding -- is what the parent asked for and not what torch's runtime was initialised with."""
    from oracle import oracle_c

    ncpu = os.cpu_count() or 1
    h = history(sample, seed)
    state = []
    epsp, p = np.zeros((sample, 6)), np.zeros(sample)
    for k in range(3):  # states after increments 1..3
        r = oracle_c.j2(h[k], epsp, p, E, NU, 0, SIG0, H, nthreads=min(ncpu, 16))
        epsp, p = r["epsp"].copy(), r["p"].copy()
        state.append((epsp, p))
    out = dict(sig=np.empty((sample, 6)), epsp=np.empty((sample, 6)), p=np.empty(sample), Ct=np.empty((sample, 6, 6)))

    def run(nt, budget):
        calls, t0 = 0, time.perf_counter()
        while True:
            k = 1 + calls % 3  # increments 2, 3, 4
            oracle_c.j2(h[k], state[k - 1][0], state[k - 1][1], E, NU, 0, SIG0, H, nthreads=nt, out=out)
            calls += 1
            el = time.perf_counter() - t0
            if (el > budget and calls >= 3) or calls >= 90:
                return sample * calls / el / 1e6

    cand = sorted({t for t in (1, 4, 8, 16, 32, 64, 128, ncpu) if t <= ncpu})
    scans = [{t: run(t, budget_s / (4.0 * len(cand))) for t in cand} for _ in range(2)]
    lo = {t: min(s_[t] for s_ in scans) for t in cand}
    hi = {t: max(s_[t] for s_ in scans) for t in cand}
    best = max(hi, key=hi.get)
    finals = [run(best, budget_s / 4.0) for _ in range(2)]
    return {"best_threads": best, "best": round(max(finals + [hi[best]]), 3), "min_at_best_threads": round(min(finals + [lo[best]]), 3),
            "single_thread": round(hi[1], 3), "thread_scan_min_max": {str(t): [round(lo[t], 2), round(hi[t], 2)] for t in cand},
            "thread_counts": cand, "visible_cores": ncpu,
            "omp_env": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")}}


def cpu_baseline(sample, seed, budget_s=14.0):
    """Two child processes, one with its OpenMP threads bound to cores (OMP_PROC_BIND=close, OMP_PLACES=cores), one
    unbound (on a shared host the first cores may be somebody else's): `value` is the best rate either reached, with the
    thread count it used."""
    import subprocess

    runs = {}
    for label, env_add in (("bound_to_cores", {"OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}), ("unbound", {})):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMP_PROC_BIND", "OMP_PLACES")}
        env.update(env_add)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-sample", str(sample),
                                "--cpu-seed", str(seed), "--cpu-budget", str(budget_s / 2.0)], env=env, capture_output=True, text=True, timeout=300)
            runs[label] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as exc:
            runs[label] = {"error": repr(exc)}
    ok = {k: v for k, v in runs.items() if "best" in v}
    if not ok:   # never lose the line over the context figure
        runs["in_process"] = cpu_baseline_scan(sample, seed, budget_s / 2.0)
        ok = {"in_process": runs["in_process"]}
    which = max(ok, key=lambda k: ok[k]["best"])
    w = ok[which]
    out = {
        "value": w["best"],
        "unit": "Mpoints/s",
        "cores": w["best_threads"],
        "kind": "port",
        "sample": f"{sample} points of the same J2 history (increments 2-4), oracle/oracle_c.c with OpenMP, best of thread counts "
        f"{w['thread_counts']} (each scanned twice, then two longer runs) on a host with {w['visible_cores']} visible cores; "
        f"`value` = the best rate of the {which} run",
        "threads_pinned": which == "bound_to_cores",
        "value_min_max": [w["min_at_best_threads"], w["best"]],
        "single_thread_value": w["single_thread"],
        "runs": runs,
    }
    # the reference's own CPU path cannot travel to the GPU box (pure Python under /root/reference): its figure
    # was measured in the build container with tools/time_reference_cpu.py and is carried along for context
    rfile = os.path.join(ROOT, "profiles", "r01_reference_cpu_container.json")
    if os.path.exists(rfile):
        try:
            r = json.load(open(rfile))
            out["reference_python_path"] = {"value": r["reference_generic_material_python_loop"]["Mpoints_per_s"], "unit": "Mpoints/s",
                                            "cores": 1, "config": r["config"], "where": r["where"]}
        except Exception:
            pass
    return out


def as_reference_update(qmap):
    """What the reference's ``QuadratureMap.update()`` does AROUND ``material.integrate`` per global Newton iteration,
    restated for a ``field_map.FieldMapBase`` (numpy stand-ins for the quadrature Functions): the cost a user of the
    unmodified reference class pays with any material behind it.  quadrature_map.py:304-313 (every gradient scattered
    into its Function, gathered back through ``dofs``, concatenated), :321-324 (integrate, then three full-array NaN
    passes -- the ISV one is where a lazily fetched array gets downloaded), :331-334 + utils.py:136-143 (flux, every
    internal state variable and the flattened tangent scattered through an index rebuilt on every call)."""
    m = qmap.material
    if not qmap._initialized:
        state = {name: f.values[qmap.dofs] for name, f in {**qmap.fluxes, **qmap.internal_state_variables}.items()}
        for name, g in qmap.gradients.items():
            g.eval(qmap.cells)
            state[name] = g.function.values[qmap.dofs, :]
        m.set_initial_state_dict(state)
        qmap._initialized = True
    blocks = []
    for name in m.gradients:
        g = qmap.gradients[name]
        g.eval(qmap.cells)
        blocks.append(g.function.values[qmap.dofs, :])
    flux, isv, ct = m.integrate(np.concatenate(blocks, axis=1))
    assert not np.any(np.isnan(flux))
    assert not np.any(np.isnan(isv))
    assert not np.any(np.isnan(ct))

    def scatter(field, array):
        flat = np.asarray(array).ravel()
        width = len(flat) // len(qmap.cells)                       # nqp * dim values per cell
        index = np.add.outer(qmap.cells * width, np.arange(width)).ravel()
        field.x.array[index] = flat

    for fields, sizes, block in ((qmap.fluxes, m.fluxes, flux), (qmap.internal_state_variables, m.internal_state_variables, isv)):
        col = 0
        for name, dim in sizes.items():
            w = max(1, dim)
            scatter(fields[name], block[:, col:col + w])
            col += w
    scatter(qmap.jacobian_flatten, ct)


def as_reference_advance(qmap):
    """quadrature_map.py:350-360 in the same terms: roll the state, final flux / ISVs scattered into the Functions."""
    m = qmap.material
    m.data_manager.update()
    final = m.get_final_state_dict()
    for name, f in {**qmap.fluxes, **qmap.internal_state_variables}.items():
        flat = np.asarray(final[name]).ravel()
        width = len(flat) // len(qmap.cells)
        f.x.array[np.add.outer(qmap.cells * width, np.arange(width)).ravel()] = flat


def host_path(jm, JAXMaterial, dev_index, n, seed, reps=9):
    """PCIe-inclusive rate of the drop-in form: numpy arrays in, numpy arrays out (`integrate(gradients)` as
    QuadratureMap.update calls it, quadrature_map.py:321).  Context only, never `value`: the transfer, not the
    kernel, is the whole cost when the consumer lives on the host."""
    h = history(n, seed)

    def timed(bind, pageable_dma=False, fresh=False, devices=None, calls=25):
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)), device=dev_index, devices=devices)
        m.set_data_manager(n)
        if pageable_dma:
            m.set_option("pageable_dma", 1)
        if bind:   # the arrays of the caller's quadrature Functions, page-locked in place (field_map.py)
            flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
            m.bind_outputs(flux=flux_fn, tangent=jac_fn)
        m.integrate(h[0])
        m.data_manager.update()
        ts = []
        same = np.array(h[1])
        # five untimed calls first: the handle measures in its calls 2-5 whether page-locking the pageable strain array for the
        # call or staging it is faster on this host (option register_input = 1) and keeps the winner; then `calls` timed ones
        for k in range(5 + calls):
            if fresh:
                g = np.array(h[1])   # QuadratureMap.update builds a new gradient array per call (quadrature_map.py:304-313)
            else:
                g = same
                g[...] = h[1]        # ... or the same Function memory, rewritten by Expression.eval before every update
            t0 = time.perf_counter()
            m.integrate(g)
            if k >= 5:
                ts.append(time.perf_counter() - t0)
            del g
        uploads.append(m.last_upload)
        m.close()
        spread.append([round(float(np.min(ts)) * 1e3, 2), round(float(np.median(ts)) * 1e3, 2), round(float(np.max(ts)) * 1e3, 2)])
        return float(np.median(ts))

    uploads, spread = [], []

    def update_cadence(accelerated, reps, nqp=8, isv_mode=None, layout="full"):
        """One ``QuadratureMap.update()`` at n points (n / 8 hexahedra with 8 Gauss points), numpy stand-ins for the
        quadrature Functions (field_map.py): the reference's cadence around ``integrate`` (as_reference_update above)
        against ``quadrature_map.AcceleratedUpdate``.  The gradient "expression" hands out the precomputed strain rows."""
        from dolfinx_materials_amd.field_map import FieldMapBase, QuadratureFieldMap

        ncell = n // nqp
        npts = ncell * nqp
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)), device=dev_index,
                        tangent_layout=layout)
        q = (QuadratureFieldMap if accelerated else FieldMapBase)(ncell, nqp, m)
        if isv_mode is not None:
            q.isv_every_update = isv_mode
        q.register_gradient("strain", None)

        class Ready:
            """Stand-in for the compiled gradient expression at zero cost on both sides: the reference's cadence gets a
            ready (ncell, nqp * 6) array back, as from ``Expression.eval(mesh, cells)``; the accelerated one finds the
            values already where ``Expression.eval(mesh, cells, values=...)`` would have written them."""
            rows = None

            def eval(self, mesh, cells, values=None):
                if values is None:
                    return self.rows
                if values.ctypes.data != self.rows.ctypes.data:
                    values[...] = self.rows
                return values

        ready = q.gradients["strain"].expression = Ready()

        def set_strain(k):
            if accelerated:   # in the gradient Function's own memory
                ready.rows = q.gradients["strain"].function.x.array.reshape(ncell, nqp * 6)
                ready.rows[...] = h[k][:npts].reshape(ncell, nqp * 6)
            else:
                ready.rows = h[k][:npts].reshape(ncell, nqp * 6)

        set_strain(0)
        in_integrate = []
        inner = m.integrate

        def timed_integrate(g, dt=0):
            t0 = time.perf_counter()
            out = inner(g, dt)
            in_integrate.append(time.perf_counter() - t0)
            return out

        m.integrate = timed_integrate
        step = q.update if accelerated else (lambda: as_reference_update(q))
        step()
        (q.advance if accelerated else (lambda: as_reference_advance(q)))()
        set_strain(1)
        step()
        ts = []
        del in_integrate[:]
        for _ in range(reps):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        (q.advance if accelerated else (lambda: as_reference_advance(q)))()
        t_adv = time.perf_counter() - t0
        fields = {"stress": q.fluxes["stress"].x.array, "jacobian": q.jacobian_flatten.x.array, "p": q.internal_state_variables["p"].x.array,
                  "epsp": q.internal_state_variables["epsp"].x.array}
        dt_ = float(np.median(ts))
        rec = {"value": round(npts / dt_ / 1e6, 2), "unit": "Mpoints/s", "ms_per_update": round(dt_ * 1e3, 2), "points": npts,
               "ms_inside_integrate": round(float(np.median(in_integrate)) * 1e3, 2), "ms_per_advance": round(t_adv * 1e3, 2), "updates_timed": reps}
        return rec, fields, (q, m)

    def device_gradient_update(reps, ncube=108):
        """The accelerated map with the gradient evaluated on the GPU (`register_device_gradient`): per update only the nodal
        displacement vector goes up (3 doubles per node: 31 MB for 108^3 hexahedra = 1.008e7 Gauss points), stress and the
        32 B/point tangent form come back into the bound fields."""
        from dolfinx_materials_amd.field_map import QuadratureFieldMap
        from dolfinx_materials_amd.gradient import Hex8Mesh

        g = np.arange(ncube + 1) / ncube
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        mm = ncube + 1
        I, J, K = (a.ravel() for a in np.meshgrid(np.arange(ncube), np.arange(ncube), np.arange(ncube), indexing="ij"))
        corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
        conn = np.stack([((I + a) * mm + (J + b)) * mm + (K + c) for a, b, c in corners], axis=1).astype(np.int32)
        ncell = conn.shape[0]
        rng = np.random.default_rng(7)
        u = {"now": 2e-3 * coords.copy() @ rng.standard_normal((3, 3)) + 2e-5 * rng.standard_normal(coords.shape)}
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)), device=dev_index)
        q = QuadratureFieldMap(ncell, 8, m)
        mesh = Hex8Mesh(coords, conn, device=dev_index)
        q.register_device_gradient(mesh, lambda: u["now"].reshape(-1))
        q.update()
        q.advance()
        u["now"] = 1.5 * u["now"]
        q.update()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            q.update()
            ts.append(time.perf_counter() - t0)
        dt_ = float(np.median(ts))
        rec = {"value": round(ncell * 8 / dt_ / 1e6, 2), "unit": "Mpoints/s", "ms_per_update": round(dt_ * 1e3, 2), "points": ncell * 8,
               "plastic_fraction": round(m.last_stats["n_plastic"] / (ncell * 8), 3),
               "pcie_bytes_per_point": {"h2d_displacement": round(24.0 * coords.shape[0] / (ncell * 8), 2), "d2h_stress": 48, "d2h_tangent_coefficients": 32},
               "note": "quadrature_map.AcceleratedUpdate.register_device_gradient on a structured hex8 mesh: the strain is evaluated inside the update kernel"}
        q.close()
        m.close()
        mesh.close()
        return rec

    def subset_update(reps, npts=5_000_000):
        """A map over a SUBSET of the cells (a multi-material problem: every other cell of 2 * npts / 8 hexahedra): nothing can be
        bound; the engine delivers every point into its row of the Functions (`integrate_rows`), against `integrate` followed by the
        library's threaded row scatter and by numpy's fancy assignment (`utils.py:136-143`)."""
        from dolfinx_materials_amd.field_map import QuadratureFieldMap

        out = {}
        for label, mode in (("rows_by_the_engine", "engine"), ("rows_on_library_threads", "threads"), ("rows_by_numpy", "numpy"), ("rows_by_the_engine_pack4", "engine")):
            ncell = npts // 8
            cells = np.arange(0, 2 * ncell, 2)
            m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)), device=dev_index,
                            tangent_layout="pack4" if label.endswith("pack4") else "full")
            if mode == "numpy":
                m.scatter_rows = m.gather_rows = None
            q = QuadratureFieldMap(2 * ncell, 8, m, cells=cells)
            if mode != "engine":
                q._accel_plan().row_outputs = False
            strain = h[0][:ncell * 8]
            q.register_gradient("strain", lambda c, strain=strain: strain.reshape(len(c), -1))
            q.update()
            q.advance()
            buf = q._accel_plan().grad_buffers["strain"]   # the next strain where Expression.eval(..., values=) writes it
            buf[...] = h[1][:ncell * 8]

            class Ready:
                def eval(self, mesh, cells, values=None):
                    return values

            q.gradients["strain"].expression = Ready()
            q.update()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                q.update()
                ts.append(time.perf_counter() - t0)
            out[label] = round(float(np.median(ts)) * 1e3, 2)
            if label.endswith("pack4"):   # (another tangent Function: compared through the stress and the state fields)
                out["pack4_same_stress_and_state"] = bool(out["stress_check"] == float(q.fluxes["stress"].x.array[::997].sum())
                                                          and out["p_check"] == float(q.internal_state_variables["p"].x.array[::997].sum()))
            else:
                out.setdefault("check", []).append(float(q.jacobian_flatten.x.array[::997].sum()))
                out["stress_check"] = float(q.fluxes["stress"].x.array[::997].sum())
                out["p_check"] = float(q.internal_state_variables["p"].x.array[::997].sum())
            q.close()
            m.close()
        same = out["check"][0] == out["check"][1] == out["check"][2]
        return {"points_in_map": npts, "points_in_fields": 2 * npts, "ms_per_update": out["rows_by_the_engine"],
                "ms_per_update_integrate_then_threaded_scatter": out["rows_on_library_threads"],
                "ms_per_update_integrate_then_numpy_assignment": out["rows_by_numpy"], "same_fields": bool(same),
                "ms_per_update_pack4_layout": out["rows_by_the_engine_pack4"], "pack4_same_stress_and_state": out.get("pack4_same_stress_and_state"),
                "value": round(npts / out["rows_by_the_engine"] / 1e3, 2), "unit": "Mpoints/s",
                "note": "HIPMaterial.integrate_rows (dxm_integrate_rows): the threads that rebuild the tangent blocks store stress, block and (default ISV mode, "
                        "bind_state_outputs(rows=True)) the internal state variables in the point's row of the Functions over all cells"}

    def packed_legs(fast, f_fields, keep_f, lazy_full=None):
        """the same accelerated update with a material that hands its tangent out packed -- jacobian_flatten is
        21 / 4 doubles per point and `jacobians[block]` (what derivative() contracts, quadrature_map.py:132-158) is written in terms
        of it (`quadrature_map.tangent_entries`).  Nothing is rebuilt on the host.  Flux / ISV Functions must be those of the full
        map bit for bit; the 6x6 block evaluated from the packed Function on a sample of points must be the full map's.  Last, the
        pack4 map and the full map run side by side, update by update (this changes the full map's fields: every comparison with
        them comes first)."""
        out = {}
        sample = np.linspace(0, fast["points"] - 1, 4096).astype(np.int64)
        want = keep_f[0].tangent_block_values(rows=sample)

        def leg(layout, isv_mode=None):
            rec, fields, keep = update_cadence(True, reps, isv_mode=isv_mode, layout=layout)
            same = all(np.array_equal(fields[k], f_fields[k]) for k in ("stress", "p", "epsp"))
            got = keep[0].tangent_block_values(rows=sample)
            err = float(np.abs(got - want).max() / np.abs(want).max())
            return rec, bool(same), err, keep

        for layout in ("sym", "pack4"):
            rec, same, err, keep = leg(layout)
            rec.update({"flux_and_isv_fields_bit_identical_to_full_layout": same, "tangent_block_max_rel_diff_on_4096_points": err,
                        "ms_over_full_layout": round(rec["ms_per_update"] / fast["ms_per_update"], 3),
                        "pcie_bytes_per_point_d2h": 48 + 32 + 56,   # (sym: the 21 entries are rebuilt on the host from the same four coefficients)
                        "host_bytes_written_per_point": {"pack4": 136, "sym": 272}[layout]})
            out[layout] = rec
            if layout == "pack4":
                # the opt-in ISV mode: 80 B/point come down per update and nothing else happens on the host
                lrec, lsame, _lerr, lkeep = leg(layout, isv_mode="lazy")
                lkeep[0].close()
                lkeep[1].close()
                del lkeep
                rec["with_isv_every_update_lazy"] = {"value": lrec["value"], "ms_per_update": lrec["ms_per_update"], "ms_per_advance": lrec["ms_per_advance"],
                                                     "same_fields_after_advance": lsame,
                                                     "ms_over_full_layout_lazy": round(lrec["ms_per_update"] / lazy_full["ms_per_update"], 3) if lazy_full else None}
                # the two maps side by side, update by update (single legs move by +-4 ms with the host's load: the ratio of
                # neighbouring calls is what this lease says about the layouts), in both ISV modes
                inter = {}
                for mode in (True, "lazy"):
                    keep_f[0].isv_every_update = keep[0].isv_every_update = mode
                    tf, tp = [], []
                    for r_ in range(reps + 2):
                        for q_, ts_ in ((keep_f[0], tf), (keep[0], tp)):
                            t0 = time.perf_counter()
                            q_.update()
                            if r_ >= 2:
                                ts_.append(time.perf_counter() - t0)
                    mf, mp = float(np.median(tf)) * 1e3, float(np.median(tp)) * 1e3
                    inter["isv_every_update" if mode is True else "isv_lazy"] = {"full_ms": round(mf, 2), "pack4_ms": round(mp, 2), "pack4_over_full": round(mp / mf, 3)}
                keep_f[0].isv_every_update = True
                rec["interleaved_with_full_layout"] = inter
            keep[0].close()
            keep[1].close()
            del keep
        out["note"] = ("HIPMaterial(tangent_layout=...) behind the same AcceleratedUpdate: the tangent Function holds (c1, c2, c3, w) [pack4: the flow direction "
                       "is dev(stress) w, read from the stress Function in the UFL expression] or the 21 upper-triangle entries [sym: rebuilt by the host threads from the same 32 B/point, 168 instead of 288 B/point of stores]; pack4: no host thread rebuilds "
                       "anything, the form compiler evaluates the block at assembly.  `interleaved_with_full_layout`: the pack4 map and the full map "
                       "updated alternately in one loop -- the comparison that does not depend on what the host did between two legs")
        return out

    def cadence_pair():
        fast, f_fields, keep_f = update_cadence(True, reps)                      # the default: ISV Functions written in every update, like the reference
        lazy, l_fields, keep_l = update_cadence(True, reps, isv_mode="lazy")     # opt-in: ISVs cross PCIe when somebody looks, and at advance()
        lazy_same = all(np.array_equal(f_fields[k], l_fields[k]) for k in f_fields)
        keep_l[0].close()
        keep_l[1].close()
        del l_fields, keep_l
        fast["with_isv_every_update_lazy"] = {"value": lazy["value"], "ms_per_update": lazy["ms_per_update"], "ms_per_advance": lazy["ms_per_advance"],
                                              "same_fields_after_advance": bool(lazy_same)}
        try:
            slow, s_fields, keep_s = update_cadence(False, 2)
            same = all(np.array_equal(f_fields[k], s_fields[k]) for k in f_fields)
            keep_s[1].close()
            del s_fields, keep_s
        except MemoryError as exc:
            slow, same = {"error": repr(exc)}, None
        try:   # (after the comparison with the reference's cadence: the side-by-side loop in here moves the full map on)
            packed = packed_legs(fast, f_fields, keep_f, lazy_full=lazy)
        except Exception as exc:  # context only
            packed = {"error": repr(exc)}
        keep_f[0].close()
        keep_f[1].close()
        if same is None:
            return {"accelerated_update": fast, "accelerated_update_packed": packed, "as_reference_update": slow}
        fast["note"] = ("quadrature_map.AcceleratedUpdate (field_map.QuadratureFieldMap): flux / jacobian_flatten memory bound as the material's output arrays, "
                        "gradient evaluated into its page-locked Function memory and uploaded by DMA, NaN count from the kernel's status record, "
                        "internal state variables written into their Functions in every update like the reference (isv_every_update = True, the default: "
                        "+56 B/point over PCIe); `with_isv_every_update_lazy`: the opt-in mode that downloads them when somebody looks and at advance()")
        slow["note"] = ("the reference's cadence around the same HIPMaterial.integrate: gradient scattered into its Function and gathered back, concatenate, three np.isnan "
                        "passes (the ISV one downloads 56 B/point), flux / ISVs / tangent scattered through a per-call np.add.outer index "
                        "(quadrature_map.py:304-334, utils.py:136-143)")
        return {"accelerated_update": fast, "accelerated_update_packed": packed, "as_reference_update": slow, "fields_bit_identical": bool(same),
                "accelerated_over_reference": round(slow["ms_per_update"] / fast["ms_per_update"], 2)}

    def page_lock_probe():
        """What page-locking a fresh strain array costs on this host (the library does it per call, option register_input):
        ~1 ms per 480 MB on transparent huge pages, 7-17 ms on 4 KiB pages (then the library stages through its ring)."""
        from dolfinx_materials_amd import _lib

        lib = _lib.load()
        ts = []
        for _ in range(3):
            a = np.array(h[1])
            t0 = time.perf_counter()
            rc = lib.dxm_host_register(a.ctypes.data, a.nbytes)
            ts.append(time.perf_counter() - t0)
            if rc == 0:
                lib.dxm_host_unregister(a.ctypes.data)
            del a
        return round(float(np.median(ts)) * 1e3 * 480e6 / (n * 48), 2)

    # the first host-buffer leg of the process runs 0-8 ms slower than the same leg later on (whichever variant comes first:
    # worker threads, page-locked areas and the runtime's transfer paths are set up in it): one throw-away leg, then the figures
    timed(True, calls=3)
    del uploads[:], spread[:]
    dt_own, dt, dt_fast = timed(False), timed(True), timed(True, pageable_dma=True)
    dt_fresh, dt_fresh_fast = timed(True, fresh=True), timed(True, pageable_dma=True, fresh=True)
    out = {"value": round(n / dt / 1e6, 2), "unit": "Mpoints/s", "ms_per_call": round(dt * 1e3, 3), "points": n,
           "into_the_materials_own_arrays": {"value": round(n / dt_own / 1e6, 2), "ms_per_call": round(dt_own * 1e3, 3)},
           "new_strain_array_every_call": {"value": round(n / dt_fresh / 1e6, 2), "ms_per_call": round(dt_fresh * 1e3, 3),
                                           "with_option_pageable_dma": round(n / dt_fresh_fast / 1e6, 2),
                                           "note": "what QuadratureMap.update hands over (quadrature_map.py:304-313): a newly allocated array per call, "
                                                   "page-locked by the library for the call (~1 ms on huge pages) or staged, whichever the handle measured faster, and released off the calling thread afterwards"},
           "with_option_pageable_dma": {"value": round(n / dt_fast / 1e6, 2), "ms_per_call": round(dt_fast * 1e3, 3),
                                        "note": "the pageable strain array handed to the runtime's own transfer path instead of the library's page-locked "
                                                "staging ring: faster, but exposed to the runtime's cache of on-the-fly page-locked ranges (DESIGN.md section 1)"},
           "page_lock_ms_per_480MB": page_lock_probe(),
           "strain_upload": {"own_arrays": uploads[0], "bound_arrays": uploads[1], "pageable_dma": uploads[2], "new_array_every_call": uploads[3]},
           "ms_per_call_min_median_max": {"own_arrays": spread[0], "bound_arrays": spread[1], "pageable_dma": spread[2], "new_array_every_call": spread[3],
                                          "new_array_every_call_pageable_dma": spread[4]},
           "calls_timed_per_leg": 25, "host_loadavg": open("/proc/loadavg").read().split()[:3] if os.path.exists("/proc/loadavg") else None,
           "legs_note": "every leg: 5 untimed calls (the handle's upload probing), then the median of 25; the same-array legs rewrite the array before each call like "
                        "Expression.eval does.  The call is bound by 16 host threads writing 288 B/point into host memory shared with the node's other tenants: "
                        "on this pool single legs move by +-4 ms from run to run and their ORDER is not significant (profiles/r04_hostpath_upload_modes.md: "
                        "register_input forced to 0 / 2 / adaptive x same / new / re-copied / rewritten array, 30 calls each, three boxes)",
           "pcie_bytes_per_point": {"h2d_strain": 48, "d2h_stress": 48, "d2h_tangent_coefficients": 32, "isv": "on demand (56)"},
           "GBs_over_pcie": round(n * 128 / dt / 1e9, 1),
           "note": "host buffers in and out through dxm_integrate: chunk-pipelined on two streams; of the tangent only (c1, c2, c3, w) cross PCIe -- the flow "
                   "direction is dev(stress) w by construction of the kernel -- and the (N,6,6) block is rebuilt by 16 host threads with the kernel's own "
                   "expression, bit-identical to the full download (what bounds the call now is those threads writing 288 B/point into host memory); "
                   "`value`: results delivered into caller-owned arrays (bind_outputs: the x.array of the quadrature Functions), the pageable strain array "
                   "page-locked for the duration of the call and uploaded by DMA, or staged through the page-locked ring by the worker threads: the handle measures both in its first calls and keeps the faster (option register_input; strain_upload says which)"}
    try:
        out.update(cadence_pair())
    except Exception as exc:  # context only
        out["accelerated_update"] = {"error": repr(exc)}
    try:
        out["accelerated_update_device_gradient"] = device_gradient_update(reps)
    except Exception as exc:  # context only
        out["accelerated_update_device_gradient"] = {"error": repr(exc)}
    try:
        out["accelerated_update_subset_of_cells"] = subset_update(reps)
    except Exception as exc:  # context only
        out["accelerated_update_subset_of_cells"] = {"error": repr(exc)}
    # one process, all GPUs of the node: G handles, G chunk pipelines, G PCIe links into the same host arrays.  In a child
    # process: the form has only ever run on one GPU (devices=[0, 0] in the tests), and whatever a first run on several does
    # must not cost the bench line
    try:
        import torch

        G = torch.cuda.device_count()
        if G > 1:
            import subprocess

            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--devices-child", str(G), "--points", str(n), "--cpu-seed", str(seed)],
                               capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["devices"] = json.loads(lines[-1]) if lines else {"error": f"child exited with {r.returncode}: {r.stderr[-300:]}"}
    except Exception as exc:  # context only
        out["devices"] = {"error": repr(exc)}
    return out


def devices_child(G, n, seed, reps=7):
    """`bench.py --devices-child G`: HIPMaterial(devices=[0..G-1]) in the host-buffer form with a new strain array per call, results
    into one bound host array (the G-link form of `host_path.new_strain_array_every_call`)."""
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    h = history(n, seed)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)), devices=list(range(G)))
    m.set_data_manager(n)
    flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.integrate(h[0])
    m.data_manager.update()
    ts = []
    for k in range(5 + reps):
        g = np.array(h[1])
        t0 = time.perf_counter()
        m.integrate(g)
        if k >= 5:
            ts.append(time.perf_counter() - t0)
        del g
    dt = float(np.median(ts))
    ok = bool(np.isfinite(flux_fn[::1009]).all() and m.last_stats["n_nan"] == 0 and m.last_stats["n_points"] == n)
    m.close()
    return {"value": round(n / dt / 1e6, 2), "unit": "Mpoints/s", "ms_per_call": round(dt * 1e3, 3), "gpus": G, "points": n, "results_finite": ok,
            "note": "HIPMaterial(behavior, devices=[0..G-1]): contiguous point blocks, one handle and one chunk pipeline per GPU, every "
                    "GPU's DMA delivering into its rows of the one bound host array; no collective (new strain array every call)"}


def other_laws(torch, jm, JAXMaterial, dev, n, reps=30, blocks_per_cu=0):
    """Kernel rates of the other laws of the path at the same batch size (device-resident, HIP events, every array where
    its first allocation put it), for context next to the headline: elastic, J2 Voce (cfg 3 parameters), FeFp J2 (cfg 4
    parameters, F = I + t (eps diag(1,-1/2,-1/2) + 0.2 eps G) as in SURVEY.md 8(d))."""
    out = {}
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev).manual_seed(4321)
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    mu = E / 2 / (1 + NU)

    def strain(sig0):
        d = torch.randn((n, 6), generator=gen, device=dev, dtype=torch.float64)
        d /= d.norm(dim=1, keepdim=True)
        return d * (torch.rand((n, 1), generator=gen, device=dev, dtype=torch.float64) * 4.0 * sig0 / (2 * mu) * np.sqrt(2.0 / 3.0))

    def fgrad(t):
        F = torch.zeros((n, 9), dtype=torch.float64, device=dev)
        G = torch.randn((n, 9), generator=torch.Generator(device=dev).manual_seed(99), device=dev, dtype=torch.float64)
        F += t * 0.2 * 2e-2 * G
        F[:, 0] += 1 + 2e-2 * t
        F[:, 1] += 1 - 1e-2 * t
        F[:, 2] += 1 - 1e-2 * t
        return F

    cases = [
        ("elastic", jm.ElasticBehavior(el), lambda: (strain(SIG0) * 0.5, strain(SIG0))),
        ("j2_voce", jm.vonMisesIsotropicHardening(el, jm.VoceHardening(350.0, 500.0, 1e3)), lambda: (lambda e: (e * 0.66, e))(strain(350.0))),
        ("fefp_j2_voce", jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0)), lambda: (fgrad(0.5), fgrad(1.0))),
    ]
    for name, beh, make_inputs in cases:
        m = JAXMaterial(beh, device=dev.index or 0)
        m.set_data_manager(n)
        if blocks_per_cu:
            m.set_option("blocks_per_cu", blocks_per_cu)
        ng, nf = m._info.n_grad, m._info.n_flux
        g0, g1 = make_inputs()
        ab = m.algorithmic_bytes_per_point
        flux = torch.empty((n, nf), dtype=torch.float64, device=dev)
        ct = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
        m.integrate_device(g0.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        # a newly allocated array needs 10-20 launches to reach its steady time (the FeFp kernel: 1.70 -> 1.63 ms over the first
        # fifteen, profiles/archive/r03_fefp_v2_kernel_stats.csv): twenty untimed launches, then the median of `reps`
        for _ in range(20):
            m.integrate_device(g1.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            m.integrate_device(g1.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            b.record()
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in ev]
        ms = float(np.median(ts))
        rc, stats = m.stats()
        out[name] = {
            "Mpoints_per_s": round(n / ms / 1e3, 1), "kernel_ms": round(ms, 4), "algorithmic_bytes_per_point": ab,
            "GBs": round(ab * n / ms / 1e6, 1), "frac": round(ab * n / ms / 1e6 / HBM_PEAK_GBS, 4),
            "kernel_ms_min_max": [round(min(ts), 4), round(max(ts), 4)], "launches": reps,
            "plastic_fraction": round(stats["n_plastic"] / n, 4), "not_converged": stats["n_not_converged"], "n_nan": stats["n_nan"],
        }
        try:
            out[name].update(law_stream_probe(torch, m, name, n, g1, flux, ct, st, ab))
        except Exception as exc:   # context only
            out[name]["stream_probe"] = {"error": repr(exc)}
        if name.startswith("fefp"):
            # SURVEY 8(d) counts an F_n read (976 B/point) that this kernel does not need: its state is the material
            # tensor Cp^-1, so 952 B/point actually cross the HBM interface
            out[name].update(bytes_moved_per_point=952, GBs_moved=round(952 * n / ms / 1e6, 1),
                             frac_moved=round(952 * n / ms / 1e6 / HBM_PEAK_GBS, 4))
        m.close()
        del g0, g1, flux, ct
        torch.cuda.empty_cache()
    return out


def law_stream_probe(torch, m, name, n, grad, flux, ct, st, alg_bytes, reps=15):
    """`frac_of_stream_probe` of one of the other laws: its kernel interleaved, launch by launch (each waited for), with the
    arithmetic-free kernel of tools/libstreammix.so that has THIS law's stream structure and runs on the SAME arrays -- the bench's
    gradient / flux / tangent arrays and the handle's own resident state (`dxm_state_ptr`), so that placement is common to both.
    elastic: strain in, stress + tangent out (384 B/point, no state); j2_voce: the 17-stream J2 shape (496 B/point);
    fefp: F + 7 state slots in, PK1 + 13 state slots + the 81-entry tangent out (952 B/point cross HBM).  Medians over `reps` rounds
    after 3 untimed ones.  > 1 would mean the kernel is faster than its arithmetic-free twin; measurement only, never on the product path."""
    import ctypes as C

    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    n64 = n // 64 * 64
    nblk = 2048
    h = m._handle
    if name == "elastic":
        lib.stream_mix_elastic_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        probe = lambda: lib.stream_mix_elastic_shape_launch(grad.data_ptr(), flux.data_ptr(), ct.data_ptr(), n64, nblk, st or None)   # noqa: E731
        moved = 384
    else:
        p0, p1 = m._lib.dxm_state_ptr(h, 0, 0, 0), m._lib.dxm_state_ptr(h, 1, 0, 0)
        ld_b = (m._lib.dxm_state_ptr(h, 0, 1, 0) or 0) - (p0 or 0)
        if not (p0 and p1 and p0 != p1 and ld_b > 0 and ld_b % 8 == 0 and ld_b // 8 >= n):
            return {"stream_probe": {"error": "the handle's state addresses are not usable for the probe"}}
        ld = ld_b // 8
        if name.startswith("fefp"):
            lib.stream_mix_fefp_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            # no dynamic LDS, no spin, whole-tile tangent rounds (ppr = 64), no prefetch, SoA state: the plain shape
            probe = lambda: lib.stream_mix_fefp_shape_launch(grad.data_ptr(), p0, p1, ld, flux.data_ptr(), ct.data_ptr(), n64, nblk, 0, 0, 0, 64, 0, 0, st or None)   # noqa: E731
            moved = 952
        else:
            lib.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
            probe = lambda: lib.stream_mix_j2_shape_launch(grad.data_ptr(), p0, p1, ld, flux.data_ptr(), ct.data_ptr(), n64, nblk, st or None)   # noqa: E731
            moved = 496
    kernel = lambda: m.integrate_device(grad.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)   # noqa: E731
    times = {"kernel": [], "probe": []}
    for r in range(reps + 3):
        for key, fn in (("kernel", kernel), ("probe", probe)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            if r >= 3:
                times[key].append(a.elapsed_time(b))
    kernel()   # flux / tangent / s1 hold the kernel's results again
    torch.cuda.synchronize()
    k_ms, p_ms = float(np.median(times["kernel"])), float(np.median(times["probe"]))
    return {"frac_of_stream_probe": round(p_ms / k_ms, 4),
            "stream_probe": {"kernel_ms_interleaved": round(k_ms, 4), "probe_ms": round(p_ms, 4), "probe_GBs": round(moved * n64 / p_ms / 1e6, 1),
                             "probe_frac_of_peak": round(moved * n64 / p_ms / 1e6 / HBM_PEAK_GBS, 4), "bytes_moved_per_point": moved, "rounds": reps,
                             "on_the_kernels_own_arrays": True}}


def stream_probes(torch, dev, n, stream, kernel_launch, eps, flux, ct, reps=30, state=None):
    """The headline kernel interleaved, launch by launch, with two arithmetic-free kernels of tools/libstreammix.so that move
    its 496 B/point: `linear` = two perfectly linear 16 B-per-lane streams (104 B read, 392 B written with non-temporal
    stores), `j2_shape` = the kernel's own stream structure (strain AoS + 7 SoA slots in; stress AoS + 7 SoA slots + tangent
    out), reading the bench's strain array and the handle's own resident state (`state` = device addresses of s0, s1 and
    their leading dimension: the SAME memory the kernel reads and writes, so that placement is common to both) and writing the
    bench's flux / tangent arrays and s1.  Medians over `reps` rounds after 5 untimed
    ones.  Measurement infrastructure (never on the product path); what the box gives a streaming kernel of this mix NOW."""
    import ctypes as C

    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_nt_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    n64 = n // 64 * 64
    rb, wb, nblk = 104, 392, 2048
    rbuf = torch.randn(n64 * rb // 8, dtype=torch.float64, device=dev)
    wbuf = torch.empty(n64 * wb // 8, dtype=torch.float64, device=dev)
    if state is not None:
        s0_ptr, s1_ptr, ld = state
        sa = sb = None
    else:
        ld = n64 + 32
        sa = torch.randn(7 * ld, dtype=torch.float64, device=dev)
        sb = torch.empty(7 * ld, dtype=torch.float64, device=dev)
        s0_ptr, s1_ptr = sa.data_ptr(), sb.data_ptr()
    lib.stream_mix_launch.argtypes = lib.stream_mix_nt_launch.argtypes
    big = torch.zeros(n64 * (rb + wb) // 8, dtype=torch.float64, device=dev)   # one direction only, the same bytes per launch
    legs = {
        "kernel": kernel_launch,
        "read_only": lambda: lib.stream_mix_launch(big.data_ptr(), wbuf.data_ptr(), n64, rb + wb, 0, nblk, stream or None),
        "write_only": lambda: lib.stream_mix_nt_launch(rbuf.data_ptr(), big.data_ptr(), n64, 0, rb + wb, nblk, stream or None),
        "linear": lambda: lib.stream_mix_nt_launch(rbuf.data_ptr(), wbuf.data_ptr(), n64, rb, wb, nblk, stream or None),
        "j2_shape": lambda: lib.stream_mix_j2_shape_launch(eps.data_ptr(), s0_ptr, s1_ptr, ld, flux.data_ptr(), ct.data_ptr(), n64, nblk, stream or None),
    }
    times = {k: [] for k in legs}
    for r in range(reps + 5):
        for k, fn in legs.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            if r >= 5:
                times[k].append(a.elapsed_time(b))
    kernel_launch()   # flux / tangent hold the kernel's results again
    torch.cuda.synchronize()
    med = {k: float(np.median(v)) for k, v in times.items()}
    out = {"kernel_ms": round(med["kernel"], 4), "linear_probe_ms": round(med["linear"], 4), "j2_shape_probe_ms": round(med["j2_shape"], 4),
           "linear_probe_GBs": round(496 * n64 / med["linear"] / 1e6, 1), "j2_shape_probe_GBs": round(496 * n64 / med["j2_shape"] / 1e6, 1),
           "kernel_over_linear_probe": round(med["linear"] / med["kernel"], 4), "kernel_over_j2_shape_probe": round(med["j2_shape"] / med["kernel"], 4),
           "read_only_probe_ms": round(med["read_only"], 4), "write_only_probe_ms": round(med["write_only"], 4),
           "read_only_probe_GBs": round(496 * n64 / med["read_only"] / 1e6, 1), "write_only_probe_GBs": round(496 * n64 / med["write_only"] / 1e6, 1),
           "rounds": reps, "j2_shape_probe_on_the_kernels_own_arrays": state is not None,
           "note": "tools/stream_mix.hip: no arithmetic, the kernel's bytes; the 17-stream probe reads and writes the kernel's own arrays (strain, resident state, flux, tangent): same placement; interleaved with the kernel in one process (each launch waited for), so all "
                   "see the same box at the same time; > 1 means the kernel is faster than the probe.  read_only / write_only: the same 4.96 GB per launch in "
                   "one direction (profiles/r04_box_survey.md: reads run at 0.79-0.80 ms on every lease, writes at 0.80 on some GPUs and 0.86 on others)"}
    del rbuf, wbuf, sa, sb, big
    torch.cuda.empty_cache()
    return out


def _code_only(text):
    """C / C++ source without comments and with runs of white space collapsed (string literals are kept as they are): what the
    compiler sees.  A comment edit does not make a measurement belong to other code."""
    import re

    pattern = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    text = pattern.sub(lambda m: " " if m.group(0).startswith("/") else m.group(0), text)
    return re.sub(r"\s+", " ", text).strip()


def source_hash(root=None, read=None):
    """Identity of the code the numbers belong to: sha1 over the kernel sources and the ABI header (what `make` compiles into
    libdxmat.so) with comments and white space taken out (`_code_only`); stamped into profiles/pmc_traffic.json by
    tools/summarize_profile.py.  `read(path) -> str` lets a caller hash another revision (`git show <rev>:<path>`)."""
    import hashlib

    root = root or ROOT
    d = os.path.join(root, "dolfinx_materials_amd", "csrc")
    names = sorted(os.path.join("dolfinx_materials_amd", "csrc", f) for f in os.listdir(d) if f.endswith((".hip", ".hpp"))) + [os.path.join("include", "dxmat.h")]
    read = read or (lambda rel: open(os.path.join(root, rel), encoding="utf-8").read())
    h = hashlib.sha1()
    for rel in names:
        h.update(rel.encode() + b"\0" + _code_only(read(rel)).encode() + b"\0")
    return h.hexdigest()[:16]


def pmc_child(args):
    """`bench.py --pmc-child`: the few launches of the headline kernel a `rocprofv3 --pmc` pass is wrapped around
    (same points per launch as the bench, the state of load increment 3, no tuning, nothing else on the GPU)."""
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = args.points
    dev = torch.device("cuda", 0)
    hist = history(n, 1234, SIG0 if args.law == "j2_linear" else 350.0)
    eps = [to_dev(h, dev) for h in hist[:3]]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    hard = jm.LinearHardening(SIG0, H) if args.law == "j2_linear" else jm.VoceHardening(350.0, 500.0, 1e3)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), hard))
    m.set_data_manager(n)
    st = torch.cuda.current_stream().cuda_stream
    for k in range(2):
        m.integrate_device(eps[k].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
    for _ in range(6):
        m.integrate_device(eps[2].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
    torch.cuda.synchronize()
    m.close()


def live_traffic(args, kernel_prefix="small_strain_kernel<1"):
    """HBM bytes per launch of the headline kernel, measured in THIS run: two `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE: separate passes, counters only besides --kernel-trace) around `bench.py --pmc-child`, started
    before this process touches the GPU.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: on gfx950 it reports
    half the bytes of wide coalesced streaming reads); both are KiB.  Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler: no nested pass"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if args.law == "j2_voce":
        kernel_prefix = "small_strain_kernel<2"
    means = {}
    tmp = tempfile.mkdtemp(prefix="dxm_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--points", str(args.points), "--law", args.law]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            env["TMPDIR"] = "/tmp"
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_prefix in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return None, f"{counter} pass failed (rc {r.returncode}, {len(vals)} samples): {r.stderr[-200:]}"
            means[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as exc:
        return None, repr(exc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    read_b = 2.0 * means["FETCH_SIZE"][0] * 1024.0
    write_b = means["WRITE_SIZE"][0] * 1024.0
    return read_b + write_b, {"source": "two rocprofv3 --pmc passes (FETCH_SIZE x2, WRITE_SIZE; KiB) around `bench.py --pmc-child` in this run",
                              "hbm_read_bytes": read_b, "hbm_write_bytes": write_b, "launches_sampled": means["FETCH_SIZE"][1]}


def to_dev(a, dev):
    """numpy -> device through a page-locked staging tensor: pageable memory is never handed to the GPU runtime (its
    cache of on-the-fly page-locked ranges goes stale when host addresses are recycled)."""
    import torch

    pin = torch.empty(a.shape, dtype=torch.float64, pin_memory=True)
    pin.numpy()[...] = a
    out = pin.to(dev)
    torch.cuda.synchronize()
    return out


def to_cpu(t):
    """device tensor -> CPU tensor through a page-locked staging tensor (gloo debug mode only)."""
    import torch

    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    torch.cuda.synchronize()
    return pin.clone()


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start the N ranks ourselves.

    Runs BEFORE anything touches the GPU (torch is not even imported here): the children are fresh
    ``python bench.py`` processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly what
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would hand them; rank 0 prints
    the JSON line on the inherited stdout.  A rank that fails takes the others down with it (they
    would otherwise wait in the rendezvous) and the exit code is that rank's."""
    import subprocess

    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DXM_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
        if args.share_gpu:
            env["DXM_BENCH_SHARE_GPU"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:   # exact PIDs of our own children
                    q.terminate()
    return rc


def hbm_plan(law, n, world, gather, copy_probe, coefficient_gather=True):
    """Device bytes one rank of `run_workload(law, n)` holds at its peak, by purpose (fp64; J2: 7 doubles of resident state per
    point and side).  Printed per rank at N > 1 and checked against the free HBM before anything is allocated."""
    state = 3 * 2 * 7 * 8 * n                                   # three load-step handles, s0 + s1 each
    plan = {"strain_increments_x4": 4 * 48 * n, "resident_state_3_handles": state, "local_stress_and_tangent": (48 + 288) * n}
    if copy_probe:                                              # rank 0 of the headline: 1 GiB source + four 1 GiB destinations, two probe arrays
        plan["copy_and_stream_probes"] = 5 * (1 << 30) + 2 * 392 * n
    if gather:
        plan["gathered_stress_and_tangent"] = (48 + 288) * n * world
        if coefficient_gather:
            plan["coefficient_gather_leg"] = state + 32 * n + 32 * n * world
    plan["total"] = sum(plan.values())
    return plan


def check_hbm_budget(torch, dev, rank, world, blocks):
    """`blocks`: name -> hbm_plan(...) of the workloads this rank will run one after the other (each releases its arrays on return).
    One line per rank on stderr; a warning above 0.9 x the free HBM of this rank's GPU, SystemExit only when the estimate exceeds
    ALL of it -- the first lease of a multi-GPU node should not be spent on an allocation failure 40 s into the run, nor on an
    estimate that is off."""
    need = max(p["total"] for p in blocks.values())
    try:
        free_b, total_b = torch.cuda.mem_get_info(dev)
    except Exception as exc:   # the check itself must never cost the run
        print("[bench.py hbm budget] " + json.dumps({"rank": rank, "world": world, "largest_block_GB": round(need / 1e9, 2), "error": repr(exc)}),
              file=sys.stderr, flush=True)
        return None
    line = {"rank": rank, "world": world, "device": str(dev), "hbm_free_GB": round(free_b / 1e9, 2), "hbm_total_GB": round(total_b / 1e9, 2),
            "largest_block_GB": round(need / 1e9, 2), "blocks_GB": {k: {kk: round(vv / 1e9, 3) for kk, vv in p.items()} for k, p in blocks.items()}}
    print("[bench.py hbm budget] " + json.dumps(line), file=sys.stderr, flush=True)
    # `need` is a hand-written estimate (state + gathered arrays; the library's padding and scratch are not in it): it only ABORTS the
    # run when it exceeds ALL the free HBM -- an estimate that is off must not cost a multi-GPU lease (one rank's exit ends the
    # others); above 0.9 x free it warns, and the JSON line records the estimate next to torch's own peak so that the plan can be checked
    if need > free_b:
        raise SystemExit(f"bench.py rank {rank}: the run would allocate {need / 1e9:.1f} GB on {dev} but only {free_b / 1e9:.1f} GB of HBM are free; "
                         f"lower --points / --cfg3-points or pass --no-gather / --no-cfg3")
    if need > 0.9 * free_b:
        print(f"[bench.py hbm budget] WARNING rank {rank}: estimated {need / 1e9:.1f} GB of {free_b / 1e9:.1f} GB free on {dev} (above 0.9 x free): continuing",
              file=sys.stderr, flush=True)
    return line


def main():
    # Every rank, whoever started it (this file's launch_ranks, torch.distributed.run, a test): the hosts of this pool support
    # dmabuf IPC only; without the variable RCCL's hipIpcGetMemHandle fails with "invalid argument".  Before torch is imported.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=10_000_000, help="Gauss points per GPU")
    ap.add_argument("--law", choices=["j2_linear", "j2_voce"], default="j2_linear",
                    help="j2_voce + --points 12500000 is cfg 3 (sig0=350, sigu=500, b=1e3); the default is cfg 2")
    ap.add_argument("--gather-steps", type=int, default=3)
    ap.add_argument("--no-p2p-gather", action="store_true",
                    help="skip the point-to-point all-gather schedule (sharding.allgather_rows_p2p) next to the collective")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: compute-only line, no gather-inclusive leg")
    ap.add_argument("--no-cfg3", action="store_true", help="N > 1: skip the cfg 3 block (J2 + Voce, 1e8 points over the ranks)")
    ap.add_argument("--cfg3-points", type=int, default=0, help="Gauss points per GPU of the cfg 3 block (default 1e8 / N)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug: all ranks use GPU 0 and the gloo backend (exercises the N > 1 path on a 1-GPU box; "
                         "same as DXM_BENCH_SHARE_GPU=1; never for reported numbers)")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="debug, --gpus 1 only: join an RCCL process group of one rank and run the gather legs through it "
                         "(the device-tensor collectives of the N > 1 path on a 1-GPU box; never for reported numbers)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-seconds", type=float, default=0.5, help="back-to-back launches before the warm-up steps (the box leaves its idle state)")
    ap.add_argument("--no-stream-probe", action="store_true", help="skip the context leg `roofline.stream_probe` (the kernel interleaved with two arithmetic-free streaming kernels)")
    ap.add_argument("--no-telemetry", action="store_true", help="skip the box block (sysfs / rocm-smi reads around and during the timed steps)")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="option blocks_per_cu for every handle of the run (0: the library's default per law)")
    ap.add_argument("--no-other-laws", action="store_true", help="skip the per-law context numbers")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive host-buffer figure")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes for roofline.traffic (falls back to the stamped profiles/pmc_traffic.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=1234, help=argparse.SUPPRESS)
    ap.add_argument("--devices-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=7.0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.devices_child:
        print(json.dumps(devices_child(args.devices_child, args.points, args.cpu_seed)), flush=True)
        return
    if args.cpu_baseline_child:   # no GPU, no torch: only numpy and the C oracle
        print(json.dumps(cpu_baseline_scan(args.cpu_sample, args.cpu_seed, args.cpu_budget)), flush=True)
        return
    if os.environ.get("DXM_BENCH_SHARE_GPU") == "1":
        args.share_gpu = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    # the box, from its own side, before this process touches the GPU (rank 0; pure sysfs + rocm-smi child processes)
    telemetry, box_before = None, None
    if not args.no_telemetry and int(os.environ.get("RANK", "0")) == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import box_telemetry as telemetry

            box_before = telemetry.condensed(telemetry.snapshot(tools=False))
        except Exception as exc:   # context only
            telemetry, box_before = None, {"error": repr(exc)}
    # roofline.traffic: measured now, before this process initialises the GPU (rank 0 of a 1-GPU run only)
    traffic, traffic_detail = None, None
    if args.gpus == 1 and not args.no_live_traffic and "WORLD_SIZE" not in os.environ:
        traffic, traffic_detail = live_traffic(args)

    import torch
    import torch.distributed as dist

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from dolfinx_materials_amd.sharding import ShardPlan, allgather_rows, allgather_rows_p2p, allgather_tangent

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback"
    # DXM_BENCH_SHARE_GPU=1 (debug only): all ranks use GPU 0 and gloo, to exercise the N > 1
    # code path on a 1-GPU box; never used for reported numbers.
    share = args.share_gpu
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # --single-rank-group (debug): an RCCL group of ONE rank on a 1-GPU box, so that the device-tensor collectives of the
    # N > 1 path (all_gather_into_tensor, the p2p schedule, the coefficient gather + rebuild) run through RCCL itself
    grouped = world > 1 or args.single_rank_group
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if telemetry is not None:   # the card of THIS rank's HIP device (several usable render nodes on a multi-GPU node)
        try:
            pr = torch.cuda.get_device_properties(dev)
            telemetry.DEFAULT_PCI = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            if box_before and box_before.get("pci") and box_before["pci"].lower() != telemetry.DEFAULT_PCI:
                box_before = dict(telemetry.condensed(telemetry.snapshot(tools=False)), taken_after_gpu_init=True)
        except Exception:
            pass
    c = argparse.Namespace(torch=torch, dist=dist, jm=jm, JAXMaterial=JAXMaterial, ShardPlan=ShardPlan, allgather_rows=allgather_rows,
                           allgather_rows_p2p=allgather_rows_p2p, allgather_tangent=allgather_tangent, rank=rank, world=world, dev=dev,
                           dev_index=dev_index, share=share, grouped=grouped, args=args, telemetry=telemetry)
    n = args.points
    K, W = args.steps, args.warmup
    seed = 1234 + rank
    run_cfg3 = world > 1 and not args.no_cfg3 and args.law == "j2_linear"
    # (the debug share mode keeps all ranks on one GPU and gathers through gloo on the host: never the full size there)
    n3 = args.cfg3_points if args.cfg3_points else (min(n, 50_000, 100_000_000 // world) if share else 100_000_000 // world)
    if world > 1:
        blocks = {"headline": hbm_plan(args.law, n, 1 if share else world, grouped and not args.no_gather and not share, rank == 0 and not share)}
        if run_cfg3:
            blocks["cfg3"] = hbm_plan("j2_voce", n3, 1 if share else world, not args.no_gather and not share, False)
        if share:   # all ranks of the debug mode allocate on GPU 0
            for b in blocks.values():
                b["total"] *= world
        budget = check_hbm_budget(torch, dev, rank, world, blocks)
    else:
        budget = None
    head = run_workload(c, args.law, n, K, W, max(1, args.gather_steps), gather=grouped and not args.no_gather, copy_probe=True)
    # N > 1: cfg 3 of SURVEY.md 8(d) beside the weak-scaling headline -- J2 + Voce, 1e8 points sharded over the ranks,
    # compute-only and the three reassembly schedules
    cfg3 = None
    if run_cfg3:
        try:
            cfg3 = run_workload(c, "j2_voce", n3, max(10, min(K, 50)), min(W, 5), max(10, args.gather_steps), gather=not args.no_gather)
        except Exception as exc:   # context block: never lose the headline line
            cfg3 = {"error": repr(exc)}
    group_info = head.pop("group_info")
    if budget is not None and group_info is not None:
        # the plan against what happened (rank 0): torch's peak counts the bench's own arrays; the library's state, staging and
        # scratch are hipMalloc'ed outside torch and show up only in the drop of free HBM
        try:
            free_after, _ = torch.cuda.mem_get_info(dev)
            group_info["hbm_budget"] = {"estimate_largest_block_GB": budget["largest_block_GB"], "free_before_GB": budget["hbm_free_GB"],
                                        "torch_peak_allocated_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
                                        "free_after_GB": round(free_after / 1e9, 2)}
        except Exception:
            pass
    if grouped:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        elapsed, kern_ms, copy_gbs = head["elapsed"], head["kernel_ms"], head["copy_gbs"]
        probe = head.get("stream_probe")
        sig0 = SIG0 if args.law == "j2_linear" else 350.0
        value = n * world * K / elapsed / 1e6
        achieved = ALG_BYTES * n / (kern_ms * 1e-3) / 1e9
        if traffic is None:   # no live pass: the figure of the last committed profile, if it is of THIS code
            why = traffic_detail
            traffic_detail = {"source": None, "live_pass": why}
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            try:
                t = json.load(open(tfile))
                if t.get("points") == n and t.get("law") == args.law and t.get("source_hash") == source_hash():
                    traffic = t.get("hbm_bytes_per_launch")
                    traffic_detail["source"] = f"profiles/pmc_traffic.json ({t.get('source')}), stamped with the source hash of this build"
            except Exception:
                pass
        out = {
            "metric": "M quadrature-point updates/s (stress+tangent, fp64)",
            "value": round(value, 3),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload_name(args.law, n),
                "points_per_gpu": n,
                "law": args.law,
                "E": E, "nu": NU, "sig0": sig0, "H": H if args.law == "j2_linear" else None,
                "plastic_fraction_inc2_3_4": head["plastic_fraction"],
                "layout": "AoS (N,6)/(N,36) boundary arrays in HBM, SoA resident state",
                "sharding": "independent contiguous point blocks, no data-path collective",
                "placement": "every array where its first allocation put it",
                "settle_launches_before_warmup": head["settle_launches"],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_over_algorithmic": round(traffic / (ALG_BYTES * n), 4) if traffic else None,
                "traffic_measurement": traffic_detail,
                "source_hash": source_hash(),
                "kernel": head["kernel"],
                "kernel_ms": round(kern_ms, 4),
                "frac_of_stream_probe": (probe or {}).get("kernel_over_j2_shape_probe"),
                "stream_probe": probe,
                "algorithmic_bytes_per_point": ALG_BYTES,
                "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                "frac_of_measured_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
            },
        }
        if telemetry is not None or box_before is not None:
            box_after = None
            try:
                box_after = {k: v for k, v in telemetry.condensed(telemetry.snapshot(tools=False)).items()
                             if k in ("sclk", "mclk", "fclk", "power_w", "temp_c", "hbm_temp_c", "gpu_busy", "mem_busy", "vram_used", "vram_of_kfd_processes_on_my_gpu")} if telemetry else None
            except Exception as exc:
                box_after = {"error": repr(exc)}
            out["box"] = {"before": box_before, "during_timed_steps": head.get("box_during"), "after": box_after,
                          "note": "tools/box_telemetry.py: sysfs of the leased GPU (the card whose render node this process can open) + the firmware's "
                                  "gpu_metrics through rocm-smi; read by this process and its children, never under a profiler.  What eight leases of six "
                                  "different GPUs showed (profiles/r04_box_survey.jsonl): clocks, power cap, partition mode, temperatures, throttle "
                                  "residencies and VRAM co-tenancy are the same on boxes that run this kernel at 0.67 and at 0.79 of peak"}
        if group_info is not None:
            out["process_group"] = group_info
        if head["gather"] is not None:
            out["gather_inclusive"] = head["gather"]
        if cfg3 is not None:
            if "error" not in cfg3:
                k3, e3, n3_ = cfg3["steps"], cfg3["elapsed"], cfg3["points"]
                cfg3 = {
                    "workload": workload_name("j2_voce", n3_), "points_per_gpu": n3_, "points_total": n3_ * world,
                    "value": round(n3_ * world * k3 / e3 / 1e6, 3), "unit": "Mpoints/s", "steps": k3, "ms_per_step": round(e3 / k3 * 1e3, 4),
                    "kernel": cfg3["kernel"], "kernel_ms": round(cfg3["kernel_ms"], 4),
                    "frac": round(ALG_BYTES * n3_ / (cfg3["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "plastic_fraction_inc2_3_4": cfg3["plastic_fraction"],
                    "gather_inclusive": cfg3["gather"],
                    "note": "SURVEY.md 8(d) cfg 3: compute-only (`value`, no data-path collective) and, in `gather_inclusive`, stress + tangent reassembled "
                            "on every rank by the RCCL all-gather, by the point-to-point schedule and as coefficients rebuilt locally",
                }
            out["cfg3"] = cfg3
        if world == 1 and not args.no_other_laws:
            try:
                torch.cuda.empty_cache()
                out["other_laws"] = other_laws(torch, jm, JAXMaterial, dev, n, blocks_per_cu=args.blocks_per_cu)
            except Exception as exc:  # context only: never lose the headline line
                out["other_laws"] = {"error": repr(exc)}
        if world == 1 and not args.no_host_path and args.law == "j2_linear":
            try:
                torch.cuda.empty_cache()
                out["host_path"] = host_path(jm, JAXMaterial, dev_index, n, seed)
            except Exception as exc:  # context only
                out["host_path"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and args.law == "j2_linear":
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), seed)
        print(json.dumps(out), flush=True)


def workload_name(law, n):
    return (("cfg2: J2 von-Mises plasticity, linear isotropic hardening, small strain, " if law == "j2_linear" else
             "cfg3: J2 von-Mises plasticity, Voce hardening (sig0=350, sigu=500, b=1e3), small strain, ")
            + f"{n:.3g} Gauss points per GPU, stress + 6x6 consistent tangent, load/unload history increments 2-4")


def run_workload(c, law, n, K, W, G, gather, copy_probe=False):
    """One law at n points per rank: set-up (three load-step contexts), W warm-up steps, K timed steps
    bracketed by barrier + synchronize (max over ranks), then -- in a process group -- the same steps followed by the
    reassembly of stress and tangent on every rank, three ways.  Everything it allocated is released on return."""
    torch, dist, jm, JAXMaterial, args = c.torch, c.dist, c.jm, c.JAXMaterial, c.args
    rank, world, dev, share, grouped = c.rank, c.world, c.dev, c.share, c.grouped
    sig0 = SIG0 if law == "j2_linear" else 350.0   # demos/jax/elastoplasticity/plane_elastoplasticity.py:60-71
    hist = history(n, 1234 + rank, sig0)
    eps = [to_dev(h, dev) for h in hist]
    del hist
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def make(layout="full"):
        hard = jm.LinearHardening(SIG0, H) if law == "j2_linear" else jm.VoceHardening(350.0, 500.0, 1e3)
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), hard), device=c.dev_index, tangent_layout=layout)
        m.set_data_manager(n)
        if args.blocks_per_cu:
            m.set_option("blocks_per_cu", args.blocks_per_cu)
        return m

    # one load-step context per timed increment k = 2, 3, 4: s0 = converged state after k-1
    mats, plastic_frac = [], []
    for k in (2, 3, 4):
        m = make()
        for i in range(k - 1):
            m.integrate_device(eps[i].data_ptr(), flux.data_ptr(), ct.data_ptr(), stream)
            rc, st = m.stats()
            assert rc == 0 and st["n_nan"] == 0
            m.data_manager.update()
        mats.append(m)

    def events_ms(nsteps, reduce=np.mean):
        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
        for i, (a_, b_) in enumerate(ev_):
            a_.record()
            j_ = i % 3
            mats[j_].integrate_device(eps[j_ + 1].data_ptr(), flux.data_ptr(), ct.data_ptr(), stream)
            b_.record()
        torch.cuda.synchronize()
        return float(reduce([a_.elapsed_time(b_) for a_, b_ in ev_]))

    def step(i):
        j = i % 3
        mats[j].integrate_device(eps[j + 1].data_ptr(), flux.data_ptr(), ct.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # (the firmware counters are read through a rocm-smi child process, ~0.3 s during which the GPU idles: before the settling
    # launches and the warm-up, not between them and the timed steps)
    sampler = fw_before = None
    tel = c.telemetry if rank == 0 else None
    if tel is not None:
        try:   # context only: a telemetry failure must never cost the line
            fw_before = tel.metrics()
            sampler = tel.Sampler(period_s=0.005)
        except Exception as exc:
            fw_before, sampler = {"error": repr(exc)}, None
    # the box leaves its idle state: back-to-back launches for --settle-seconds, no host work in between (telemetry of the boxes
    # in profiles/r04_box_survey.jsonl: clocks do not move under this load and no launch but the very first after an idle gap
    # is slow -- the settling costs nothing and takes the question off the table)
    t_settle = time.perf_counter()
    n_settle = 0
    while time.perf_counter() - t_settle < args.settle_seconds:
        for i in range(20):
            step(n_settle + i)
        n_settle += 20
        torch.cuda.synchronize()
    for i in range(W):
        step(i)
    for j in range(3):
        step(j)
        rc, st = mats[j].stats()
        assert rc == 0 and st["n_nan"] == 0
        plastic_frac.append(st["n_plastic"] / n)

    # ---- the timed region: EXACTLY K steps between barrier + synchronize, every array where its first allocation put it -------
    try:
        box_before = tel.fast_read(tel.my_card()) if (tel is not None and tel.my_card()) else None   # a few file reads
    except Exception:
        box_before = None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    barrier()
    if sampler is not None:
        try:
            sampler.start()
        except Exception:
            sampler = None
    t0 = time.perf_counter()
    for i in range(K):
        ev[i][0].record()
        step(i)
        ev[i][1].record()
    barrier()
    t1 = time.perf_counter()
    if sampler is not None:
        try:
            sampler.stop()
        except Exception:
            pass
    elapsed = t1 - t0
    per_launch = [a.elapsed_time(b) for a, b in ev]
    kern_ms = float(np.mean(per_launch))
    box_during = None
    if tel is not None:
        try:
            fw_after = tel.metrics()
            box_during = {"sysfs_samples": sampler.summary() if sampler is not None else None, "firmware_counters": tel.metrics_delta(fw_before, fw_after),
                          "clocks_at_start": box_before,
                          "kernel_ms_min_median_max": [round(float(np.min(per_launch)), 4), round(float(np.median(per_launch)), 4), round(float(np.max(per_launch)), 4)],
                          "note": "sysfs_samples: min / median / max of a 5 ms sampler thread across the K timed steps (20 steps are ~20 ms: a handful of samples); "
                                  "firmware_counters: gpu_metrics accumulators read before the settling launches and after the timed steps (settling + warm-up + "
                                  "timed steps + ~0.3 s of tool time)"}
        except Exception as exc:   # context only
            box_during = {"error": repr(exc)}

    # ---- context, after the timed region -----------------------------------------------------------------------------------
    # (1) the same kernel against two arithmetic-free streaming kernels that move its bytes, interleaved launch by launch on this
    # box now: a linear 104 B-in / 392 B-out pair of streams with non-temporal stores, and the kernel's own 17-stream shape
    stream_probe = None
    if rank == 0 and copy_probe and not share and not args.no_stream_probe:
        try:
            hm = mats[1]   # (step(1): the handle of increment 3, strain eps[2])
            p0 = hm._lib.dxm_state_ptr(hm._handle, 0, 0, 0)
            p1 = hm._lib.dxm_state_ptr(hm._handle, 1, 0, 0)
            ld_b = (hm._lib.dxm_state_ptr(hm._handle, 0, 1, 0) or 0) - (p0 or 0)     # field 1 (epsp) starts one slot after field 0 (p)
            own = (p0, p1, ld_b // 8) if (p0 and p1 and p0 != p1 and ld_b > 0 and ld_b % 8 == 0 and ld_b // 8 >= n) else None
            stream_probe = stream_probes(torch, dev, n, stream, lambda: step(1), eps[2], flux, ct, state=own)
        except Exception as exc:   # context only
            stream_probe = {"error": repr(exc)}

    # device-copy bandwidth of THIS box (read + write bytes of a 1 GiB fp64 copy): the practical
    # ceiling a streaming kernel sees here.  The rate depends on which physical regions the two
    # buffers come from (5.2 vs 4.65 TB/s), so the best of four destinations
    # is reported.
    copy_gbs = None
    if rank == 0 and copy_probe:
        a_ = torch.empty(1 << 27, dtype=torch.float64, device=dev).normal_()
        dsts = [torch.empty_like(a_) for _ in range(4)]
        rates = []
        for b_ in dsts:
            for _ in range(3):
                b_.copy_(a_)
            cev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for x, y in cev:
                x.record()
                b_.copy_(a_)
                y.record()
            torch.cuda.synchronize()
            rates.append(2 * a_.numel() * 8 / (float(np.median([x.elapsed_time(y) for x, y in cev])) * 1e-3) / 1e9)
        copy_gbs = max(rates)
        del a_, b_, dsts
        torch.cuda.empty_cache()

    gather_out = None
    group_info = None
    cdev = torch.device("cpu") if share else dev  # gloo debug mode keeps collectives on the host
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # what the process group itself saw: every rank contributes 1
        seen = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        group_info = {"backend": dist.get_backend(), "ranks_in_group": dist.get_world_size(),
                      "ranks_counted_by_all_reduce": int(seen.item()),
                      "launcher": os.environ.get("DXM_BENCH_LAUNCHER", "torch.distributed.run"),
                      "share_gpu_debug_mode": bool(share),
                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}   # set by main() when the launcher did not
    if grouped and gather:
        # gather-inclusive variant: reassemble stress and tangent on every rank over xGMI.  Context
        # for cfg 3 only: a failure here (e.g. not enough HBM for the gathered buffers when the GPUs
        # are shared) must never lose the headline line, so it is reported instead of raised.
        # device tensors: the kernels write straight into this rank's rows of the gathered arrays and the gathers are
        # in place (no local copy); the gloo debug mode gathers host copies
        inplace = not share
        plan = g_flux = g_ct = my_flux = my_ct = None

        def timed_gather(fn):
            def gstep(i):
                if inplace:
                    j = i % 3
                    mats[j].integrate_device(eps[j + 1].data_ptr(), my_flux.data_ptr(), my_ct.data_ptr(), stream)
                    fn(my_flux, plan, out=g_flux)
                    fn(my_ct, plan, out=g_ct)
                else:
                    step(i)
                    fn(to_cpu(flux), plan, out=g_flux)
                    fn(to_cpu(ct), plan, out=g_ct)

            gstep(0)
            barrier()
            g0 = time.perf_counter()
            for i in range(G):
                gstep(i)
            barrier()
            gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=cdev)
            dist.all_reduce(gt, op=dist.ReduceOp.MAX)
            return {"value": round(n * world * G / float(gt.item()) / 1e6, 3), "unit": "Mpoints/s",
                    "ms_per_step": round(float(gt.item()) / G * 1e3, 4), "steps": G}

        try:
            plan = c.ShardPlan(n * world, world)
            g_flux = torch.empty((n * world, 6), dtype=torch.float64, device=cdev)
            g_ct = torch.empty((n * world, 36), dtype=torch.float64, device=cdev)
            my_flux, my_ct = plan.local_view(g_flux, rank), plan.local_view(g_ct, rank)
            gather_out = timed_gather(c.allgather_rows)
            gather_out["collective"] = ("RCCL in-place" if not share else "gloo (debug)") + " all_gather_into_tensor of stress (N,6) and tangent (N,36), fp64"
            gather_out["bytes_received_per_rank"] = int((world - 1) * n * 42 * 8)
            gather_out["link_GBs_per_rank"] = round(gather_out["bytes_received_per_rank"] / (gather_out["ms_per_step"] * 1e-3) / 1e9, 1)
        except Exception as exc:
            gather_out = {"error": repr(exc)}
        if not args.no_p2p_gather and "error" not in gather_out:
            # the same reassembly as one batch of point-to-point transfers (all links at once on the xGMI mesh)
            try:
                gather_out["p2p_schedule"] = timed_gather(c.allgather_rows_p2p)
            except Exception as exc:
                gather_out["p2p_schedule"] = {"error": repr(exc)}
        if "error" not in gather_out:
            # ... and with the tangent travelling as (c1, c2, c3, w) (32 instead of 288 B/point on the links: the flow
            # direction is rebuilt from the gathered stress) and rebuilt on every rank by dxm_expand_tangent_pack4_device
            # (bit-identical): kernels with tangent_layout="pack4"
            cmats = []
            try:
                ct9 = torch.empty((n, 4), dtype=torch.float64, device=dev)
                for k in (2, 3, 4):
                    m = make("pack4")
                    for i in range(k - 1):
                        m.integrate_device(eps[i].data_ptr(), flux.data_ptr(), ct9.data_ptr(), stream)
                        m.data_manager.update()
                    cmats.append(m)
                coef_all = torch.empty((n * world, 4), dtype=torch.float64, device=cdev)
                my_c9 = plan.local_view(coef_all, rank)

                def cstep(i):
                    j = i % 3
                    if inplace:
                        cmats[j].integrate_device(eps[j + 1].data_ptr(), my_flux.data_ptr(), my_c9.data_ptr(), stream)
                        c.allgather_rows(my_flux, plan, out=g_flux)
                        c.allgather_tangent(my_c9, plan, out=g_ct, coef_all=coef_all, flux_all=g_flux)
                    else:
                        cmats[j].integrate_device(eps[j + 1].data_ptr(), flux.data_ptr(), ct9.data_ptr(), stream)
                        c.allgather_rows(to_cpu(flux), plan, out=g_flux)
                        c.allgather_tangent(to_cpu(ct9), plan, out=g_ct, coef_all=coef_all, flux_all=g_flux)

                cstep(0)
                barrier()
                g0 = time.perf_counter()
                for i in range(G):
                    cstep(i)
                barrier()
                gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=cdev)
                dist.all_reduce(gt, op=dist.ReduceOp.MAX)
                gather_out["coefficient_gather"] = {
                    "value": round(n * world * G / float(gt.item()) / 1e6, 3), "unit": "Mpoints/s",
                    "ms_per_step": round(float(gt.item()) / G * 1e3, 4), "steps": G,
                    "bytes_received_per_rank": int((world - 1) * n * 10 * 8),
                    "note": "stress (N,6) + (c1, c2, c3, w) (N,4) all-gathered: 80 instead of 336 B/point on the links; the (N,36) tangent rebuilt locally "
                            "on every rank from both (dxm_expand_tangent_pack4_device), bit-identical to gathering full blocks"}
                del ct9, coef_all, my_c9
            except Exception as exc:
                gather_out["coefficient_gather"] = {"error": repr(exc)}
            for m in cmats:
                m.close()
        del g_flux, g_ct, my_flux, my_ct
    kernel = mats[0].kernel_name
    for m in mats:
        m.close()
    del eps, flux, ct
    torch.cuda.empty_cache()
    return {"elapsed": elapsed, "kernel_ms": kern_ms, "stream_probe": stream_probe, "box_during": box_during, "settle_launches": n_settle,
            "plastic_fraction": [round(x, 4) for x in plastic_frac],
            "copy_gbs": copy_gbs, "gather": gather_out, "group_info": group_info, "kernel": kernel, "steps": K, "points": n}


if __name__ == "__main__":
    main()
