/*
 * dxmat_experimental.h -- entry points libdxmat.so exports BESIDE the contract of dxmat.h: measurement helpers and research
 * code.  Nothing in the drop-in path (dolfinx_materials_amd.HIPMaterial.integrate / AcceleratedUpdate.update) calls them; they
 * may change or disappear between ABI versions.  Kept because bench.py / tools/ report figures measured with them and because
 * examples/device_fem.py (a matrix-free stand-in for the assembly the reference leaves to dolfinx) is built on the last four.
 */
#ifndef DXMAT_EXPERIMENTAL_H
#define DXMAT_EXPERIMENTAL_H

#include "dxmat.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- where the resident state sits (profiles/NOTES.md, "placement") ------------------------------------------------------
 * The update kernel's time depends on which physical memory the handle's state and the caller's arrays got (two levels, up to
 * 13 % apart at 1e7 J2 points, not steerable from user space).  dxm_tune_placement measures: it runs the update (as
 * dxm_integrate_device would, on the handle's own stream, synchronously) with the caller's device arrays on up to
 * max_candidates fresh state allocations and keeps the fastest; s0 is preserved, s1 / flux_dev / ct_dev / the stats end up as
 * after one dxm_integrate_device.  The second half of the candidates jump ahead by skip blocks that together stay below option
 * "tune_max_skip_bytes" (default 2 GiB); option "tune_verbose" logs every candidate (and the host-buffer form's chunk timeline)
 * to stderr.  ms_before / ms_after: kernel time on the initial / chosen placement; n_tried: candidates measured (any may be
 * NULL).  No-op for laws without state. */
int dxm_tune_placement(dxm_material* m, const double* grad_dev, double* flux_dev, double* ct_dev,
                       int max_candidates, double* ms_before, double* ms_after, int* n_tried);
/* `launches` updates with these device arrays on the handle's own stream (synchronous, two warm-up launches first): best launch
 * time in ms.  Acts like dxm_integrate_device otherwise. */
int dxm_time_device(dxm_material* m, const double* grad_dev, double* flux_dev, double* ct_dev, int launches, double* best_ms);

/* ---- assembly-side consumers on the device (hex8 meshes with 8 Gauss points per cell, small strain) --------------------------
 * What dolfinx assembly does with the quadrature Functions QuadratureMap.update filled -- `dot(sig, strain(v)) * dx` and its
 * derivative with the tangent blocks (tests/uniaxial_tension.py:62-67, quadrature_map.py:132-158) -- restated matrix-free:
 *   internal force   f = sum_q w detJ B_q^T sigma_q               flux_dev (npoints,6) Mandel -> f_dev (n_nodes*3)
 *   tangent apply    y = sum_q w detJ B_q^T Ct_q B_q x            ct_dev in `layout` (DXM_TANGENT_FULL or _COEF)
 *   tangent diagonal d = diag(sum_q w detJ B_q^T Ct_q B_q)        coefficient layout only
 * Deterministic (element values, then a node gather; no atomics), asynchronous on hip_stream.  Out of the scope of the path
 * (FEM assembly stays on the host). */
int dxm_mesh_set_weights(dxm_mesh* mesh, const double* weights /* nqp */);
int dxm_mesh_internal_force_device(dxm_mesh* mesh, const double* flux_dev, double* f_dev, void* hip_stream);
int dxm_mesh_tangent_apply_device(dxm_mesh* mesh, const double* ct_dev, int layout, const double* x_dev, double* y_dev,
                                  void* hip_stream);
int dxm_mesh_tangent_diagonal_device(dxm_mesh* mesh, const double* coef_dev, double* d_dev, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DXMAT_EXPERIMENTAL_H */
