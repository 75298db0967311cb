/*
 * dxmat.h -- C ABI of libdxmat.so, the MI355X (gfx950) batched constitutive-update engine.
 *
 * This is the drop-in boundary for ONE path of bleyerj/dolfinx_materials: the per-Gauss-point
 * constitutive update reached through
 *     NonlinearMaterialProblem._constitutive_update   (dolfinx_materials/solvers.py:173-176)
 *  -> QuadratureMap.update()                          (dolfinx_materials/quadrature_map.py:297-334)
 *  -> material.integrate(gradients)                   (dolfinx_materials/jaxmat.py:208-234,
 *                                                      dolfinx_materials/generic.py:176-189)
 *  -> batched_constitutive_update = vmap(jacfwd(constitutive_update))
 *                                                     (dolfinx_materials/jaxmat.py:147-155)
 *
 * The reference has no FFI of its own (its boundary is the duck-typed Python `Material` protocol);
 * the Python classes in dolfinx_materials_amd/ implement that protocol on top of these entry
 * points through ctypes.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - all floating point data is IEEE fp64;
 *   - "AoS" = row-major (npoints, dim), component fastest: the memory layout of a dolfinx
 *     quadrature Function (dolfinx_materials/utils.py:98-104, :136-143);
 *   - symmetric tensors are Mandel 6-vectors, non-symmetric ones 9-vectors in the order
 *     [11,22,33,12,21,13,31,23,32] (dolfinx_materials/utils.py:146-190);
 *   - the tangent is Ct[i][j] = d flux_i / d gradient_j, row-major (npoints, nflux*ngrad)
 *     (dolfinx_materials/quadrature_map.py:83-105, :334);
 *   - int return codes: 0 = ok, > 0 = number of points whose local Newton did not converge,
 *     < 0 = hard error (message from dxm_last_error());
 *   - HIP graphs: the device-pointer entry points are capturable.  A captured launch bakes in the two
 *     state buffers (advance() swaps them on the host) and the parameters, so a graph is valid only while
 *     dxm_launch_generation() returns the value it had at capture time; the value comes back when the
 *     buffers swap back (two graphs, captured at consecutive increments, serve a whole load history);
 *   - one handle per (material, device); a handle is not thread-safe (the reference calls the path
 *     from the PETSc SNES callback on one thread: dolfinx_materials/solvers.py:72); successive
 *     device-pointer launches on one handle must be ordered on the same HIP stream (advance()
 *     and revert() only swap pointers on the host);
 *   - there is NO CPU fallback: every entry point that computes fails with a negative code when no
 *     HIP device is usable;
 *   - the library exports what this header declares and nothing else (tests/test_abi.py).  The placement search and launch
 *     timing of ABI 3-5 are gone (profiles/r05_placement_decision.md: a search rescued 4 of 12 slow leases); the matrix-free
 *     assembly kernels of the stand-in FE loop are examples/csrc/dxmfem.hip (examples/libdxmfem.so), not product.
 */
#ifndef DXMAT_H
#define DXMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXM_ABI_VERSION 6   /* 6: dxm_tune_placement, dxm_time_device, the dxm_mesh_* assembly operators and the options "tune_verbose" /
                             *    "tune_max_skip_bytes" removed (there is no dxmat_experimental.h any more); option "verbose" */

/* Constitutive laws (what `behavior.constitutive_update` is in jaxmat.py:163). */
enum {
  /* sigma = C:eps, Ct = C.  Replaces python_materials/elasticity.py:21-24.
   * params = [E, nu] */
  DXM_LAW_ELASTIC_ISO = 0,
  /* small-strain J2, linear isotropic hardening R(p) = sig0 + H p, implicit radial return +
   * consistent tangent.  Spec: tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77.
   * params = [E, nu, sig0, H] */
  DXM_LAW_J2_LINEAR = 1,
  /* small-strain J2, Voce hardening R(p) = sig0 + (sigu - sig0)(1 - exp(-b p))
   * (jaxmat vonMisesIsotropicHardening + VoceHardening as constructed in
   * demos/jax/elastoplasticity/plane_elastoplasticity.py:60-73).
   * params = [E, nu, sig0, sigu, b] */
  DXM_LAW_J2_VOCE = 2,
  /* finite-strain FeFp J2 plasticity with Voce hardening, gradient F (9), flux PK1 (9)
   * (jaxmat FeFpJ2Plasticity as constructed in tests/test_FeFp_jax.py:7-20).
   * params = [E, nu, sig0, sigu, b] */
  DXM_LAW_FEFP_J2_VOCE = 3,
  /* the same finite-strain law with linear hardening R(p) = sig0 + H p.
   * params = [E, nu, sig0, H] */
  DXM_LAW_FEFP_J2_LINEAR = 4,
  DXM_LAW_COUNT = 5
};

/* Which state: s0 = beginning of the increment, s1 = end (generic.py:204-216, jaxmat.py:30-43). */
enum { DXM_S0 = 0, DXM_S1 = 1 };

#define DXM_MAX_STATE_FIELDS 4

typedef struct dxm_law_info {
  int32_t n_grad;         /* gradient size (6 strain / 9 F)              jaxmat.py:166-175 */
  int32_t n_flux;         /* flux size (6 stress / 9 PK1)                jaxmat.py:177-186 */
  int32_t n_params;       /* number of material parameters                                 */
  int32_t n_isv_fields;   /* number of user-visible internal state variables               */
  int32_t isv_dim[DXM_MAX_STATE_FIELDS];      /* size of each (scalar = 1)  jaxmat.py:188-193 */
  const char* isv_name[DXM_MAX_STATE_FIELDS]; /* e.g. "p", "epsp", "be_bar"                   */
  int32_t n_isv_total;    /* sum of isv_dim = columns of the isv array of integrate()      */
  int32_t algorithmic_bytes_per_point; /* 384 / 496 / 496 / 976    */
} dxm_law_info;

/* Per-batch status of the last integrate (the reference only has host-side NaN asserts,
 * quadrature_map.py:322-324, and no local-Newton report on the JAX path). */
typedef struct dxm_stats {
  int64_t n_points;
  int64_t n_plastic;        /* points that took the plastic branch */
  int64_t n_not_converged;  /* local Newton hit maxit              */
  int64_t n_nan;            /* points with a non-finite flux, isv or tangent (quadrature_map.py:322-324) */
  int32_t max_local_iters;
  int32_t upload;   /* host-buffer form: how the gradient reached the GPU in this call -- DXM_UPLOAD_* below; 0 otherwise */
} dxm_stats;

typedef struct dxm_material dxm_material; /* opaque handle */

/* ---- library / device ------------------------------------------------------------------- */
int dxm_abi_version(void);
/* 1 if this library was built with -DDXM_CUSTOM_HARDENING (a user-supplied isotropic hardening law
 * R(p), dR/dp compiled into the DXM_LAW_J2_VOCE / DXM_LAW_FEFP_J2_VOCE kernels; those laws then
 * take params = [E, nu, sig0, c0..c5]), 0 for the stock library. */
int dxm_has_custom_hardening(void);
/* Message of the last failing call on this thread ("" if none). */
const char* dxm_last_error(void);
/* Number of HIP devices, or < 0 if the HIP runtime cannot be initialised. */
int dxm_device_count(void);
int dxm_law_info_get(int law, dxm_law_info* out);

/* ---- life cycle: Material.set_data_manager(ngauss)  (generic.py:172-174, jaxmat.py:195-197,
 *      quadrature_map.py:231-233) ---------------------------------------------------------- */
/* Allocates device-resident SoA state s0/s1 for npoints Gauss points on `device` and
 * initialises it (p = 0, epsp = 0, be_bar = identity; cf. behavior.init_state, jaxmat.py:35). */
dxm_material* dxm_create(int law, const double* params, int n_params, int64_t npoints, int device);
int dxm_destroy(dxm_material* m);
int64_t dxm_npoints(const dxm_material* m);
int dxm_law(const dxm_material* m);
/* Material.update_material_property (generic.py:119-120): replace the parameter vector. */
int dxm_set_params(dxm_material* m, const double* params, int n_params);
/* Tangent layout integrate writes, doubles per point:
 *   DXM_TANGENT_FULL   n_flux*n_grad (36 / 81), row-major: what `jacobian_flatten` holds (quadrature_map.py:83-105);
 *   DXM_TANGENT_SYM    21 upper-triangle entries (i <= j), small-strain laws (symmetric tangent);
 *   DXM_TANGENT_COEF   9 = (c1, c2, c3, n[0..5]) of Ct = c1 1x1 + c2 I + c3 n x n, J2 laws
 *                      (tests/mfront/IsotropicLinearHardeningPlasticity.mfront:66-69 with M expanded);
 *   DXM_TANGENT_PACK4  4 = (c1, c2, c3, w), J2 laws: the kernels form n = dev(stress) w, so the stress of the same update
 *                      and these four rebuild the block bit for bit (dxm_expand_tangent_pack4_device). */
/* dxm_stats.upload */
enum { DXM_UPLOAD_NONE = 0, DXM_UPLOAD_PAGE_LOCKED = 1 /* the caller's array was page-locked already: DMA */,
       DXM_UPLOAD_REGISTERED = 2 /* page-locked by the library for the duration of the call: DMA */,
       DXM_UPLOAD_STAGED = 3 /* copied chunk by chunk into the library's page-locked ring by the worker threads */,
       DXM_UPLOAD_RUNTIME = 4 /* option pageable_dma: handed to the runtime's own pageable path */ };
enum { DXM_TANGENT_FULL = 0, DXM_TANGENT_SYM = 1, DXM_TANGENT_COEF = 2, DXM_TANGENT_PACK4 = 3 };
int dxm_set_tangent_layout(dxm_material* m, int layout);
/* doubles per point of the tangent array integrate writes (36 / 21 / 9 / 81). */
int dxm_tangent_size(const dxm_material* m);
/* Local Newton controls (Voce / traced hardening, FeFp): at most maxit iterations, stop when
 *   |r| <= max(tol, rtol * sigma_eq_trial),   tol = rtol * max(|sig0|, 2e-8 mu)
 * (an absolute floor from the initial yield stress -- or from the shear modulus when sig0 is 0 -- and a relative term that
 * follows the trial von Mises stress of the point: csrc/small_strain.hpp, csrc/dxm_common.hpp J2Params.tol / .rtol).
 * FeFp solves two equations: the same bound on the yield residual (as a Kirchhoff stress) AND |det(be_bar) - 1| <= 1e-14.
 * Points that stop at maxit are counted in dxm_stats.n_not_converged (and are the positive return value of the integrate calls). */
int dxm_set_newton(dxm_material* m, int maxit, double rtol);

/* ---- state: set_initial_state_dict / get_initial_state_dict / get_final_state_dict
 *      (generic.py:194-201, jaxmat.py:199-206; consumers quadrature_map.py:279,294,356-360) -- */
/* host AoS (npoints, isv_dim[field]) -> device SoA of state `which` */
int dxm_set_state(dxm_material* m, int which, int field, const double* host_aos);
/* device SoA -> host AoS (npoints, isv_dim[field]) */
int dxm_get_state(dxm_material* m, int which, int field, double* host_aos);
/* DataManager.update(): s0 <- s1   (generic.py:212-213, jaxmat.py:39-40; quadrature_map.py:355) */
int dxm_advance(dxm_material* m);
/* DataManager.revert(): s1 <- s0   (generic.py:215-216, jaxmat.py:42-43) */
int dxm_revert(dxm_material* m);
/* Gradient and flux of a state (generic.py:194-201: get_initial_state_dict()["Strain"] / ["Stress"]) from the device copies
 * of the last HOST-BUFFER call: dxm_get_io(m, which, kind, host_aos) downloads the gradient (kind 0) or flux (kind 1),
 * (npoints, n_grad | n_flux).  With option "keep_initial_io" dxm_advance keeps the copies of the accepted state as those of s0
 * (a pointer swap).  A state accepted from a call without host arrays (device pointers; a fused displacement for the gradient)
 * holds no copy.  dxm_io_held(m, which): bit 0 = a gradient is held for that state, bit 1 = a flux (negative: bad arguments). */
int dxm_io_held(const dxm_material* m, int which);
int dxm_get_io(dxm_material* m, int which, int kind, double* host_aos);

/* ---- the hot path: Material.integrate(gradients, dt)  (jaxmat.py:208-234, generic.py:176-189;
 *      consumer quadrature_map.py:321) ------------------------------------------------------ */
/* Host-buffer form (what QuadratureMap hands over): grad_aos (npoints, n_grad) in host memory;
 * writes flux_aos (npoints, n_flux), isv_aos (npoints, n_isv_total) and ct_aos
 * (npoints, n_flux*n_grad) to host memory (any of the three may be NULL to skip its download; the ISVs
 * are only needed at advance() and can be fetched later with dxm_isv_host).
 * Reads state s0, writes state s1.  Synchronous.  `stats` may be NULL. */
int dxm_integrate(dxm_material* m, const double* grad_aos, double dt, double* flux_aos,
                  double* isv_aos, double* ct_aos, dxm_stats* stats);
/* The same for a QuadratureMap over a SUBSET of the cells (quadrature_map.py:66-73 with `cells`; one map per material in a
 * multi-material problem): grad_aos (npoints, n_grad) are the map's own points as above, but flux_rows / ct_rows are the BASES of
 * the quadrature Functions over ALL cells and point i belongs in their row rows[i] (the `dofs` index the map built once,
 * quadrature_map.py:231-233) -- what `_update_vals(field, values, cells)` does with a fancy assignment per array per update
 * (utils.py:136-143).  Of each point 80 B cross PCIe into the library's page-locked landing areas and the worker threads that
 * rebuild the (6, 6) blocks store them, and the stress, straight into their rows; the caller's arrays need not be page-locked.
 * (The elastic law's constant block is filled in by the same threads, the FeFp laws move their 54 building blocks + 9 stress
 * components per point.)  A handle with a packed tangent layout (DXM_TANGENT_SYM / _COEF / _PACK4): ct_rows has dxm_tangent_size
 * doubles per row, the kernel's own 21 / 9 / 4 numbers land and are moved to their rows as they are -- nothing is rebuilt.  The index
 * holds each row once and is NOT range-checked here (dxm_host_index_range is the check; the Python layer runs it per call);
 * internal state variables: bound with dxm_bind_isv_output they are delivered into their rows by the same call, else
 * dxm_isv_host / dxm_get_state when needed. */
int dxm_integrate_rows(dxm_material* m, const double* grad_aos, double dt, double* flux_rows, double* ct_rows,
                       const int64_t* rows, dxm_stats* stats);
/* Device-pointer form: all three arrays are device memory on the handle's device (e.g. torch
 * tensors' data_ptr()); the kernel is enqueued on `hip_stream` (a hipStream_t, NULL = default
 * stream) and the call returns without synchronising.  Returns 0 or < 0. */
int dxm_integrate_device(dxm_material* m, const double* grad_dev, double dt, double* flux_dev,
                         double* ct_dev, void* hip_stream);
/* Waits for the last integrate on the handle and returns its status (same code as dxm_integrate). */
int dxm_get_stats(dxm_material* m, dxm_stats* stats);
/* Packs the user-visible ISVs of state `which` into a device AoS (npoints, n_isv_total) array,
 * enqueued on hip_stream (the `_hcat_mixed` of jaxmat.py:46-58, :227-229, on device). */
int dxm_isv_device(dxm_material* m, int which, double* isv_aos_dev, void* hip_stream);
/* Device address of component `comp` of SoA state field `field` (npoints contiguous doubles). */
const double* dxm_state_ptr(const dxm_material* m, int which, int field, int comp);
/* Rebuild full tangents from their coefficient form on the device: coef_dev (npoints, 9) as written with
 * DXM_TANGENT_COEF -> ct_dev (npoints, 36), the block DXM_TANGENT_FULL writes, bit for bit; asynchronous on
 * hip_stream of `device`.  For consumers that move the 72 B/point form (e.g. across xGMI: an all-gather of
 * coefficients followed by this kernel instead of an all-gather of 288 B/point blocks) and need the full block. */
int dxm_expand_tangent_device(const double* coef_dev, int64_t npoints, double* ct_dev, int device, void* hip_stream);
/* The same from the 32 B/point form: flux_dev (npoints, 6) the stress and pack_dev (npoints, 4) the (c1, c2, c3, w) of the
 * same update (DXM_TANGENT_PACK4) -> ct_dev (npoints, 36), bit for bit what DXM_TANGENT_FULL writes.  An all-gather of
 * stress + this form moves 80 instead of 120 (coefficients) or 336 (full blocks) B/point across xGMI. */
int dxm_expand_tangent_pack4_device(const double* flux_dev, const double* pack_dev, int64_t npoints, double* ct_dev, int device,
                                    void* hip_stream);
/* Name of the HIP kernel integrate launches for this handle (for profile filtering). */
const char* dxm_kernel_name(const dxm_material* m);
/* Identity of the launch configuration: changes whenever a launch captured into a HIP graph before would
 * no longer do what a fresh call does -- dxm_advance (the two state buffers swap: the low bit flips and
 * flips back at the next advance), dxm_set_params / dxm_set_newton / dxm_set_tangent_layout /
 * dxm_set_option and anything else that moves the resident state (the upper bits increase).  Replay a captured graph only while the
 * value equals the one read at capture time.  dxm_revert does not change it. */
uint64_t dxm_launch_generation(const dxm_material* m);
/* Tell the handle that the caller has replayed a HIP graph containing a launch of this handle: the replay
 * is invisible to the library, but it rewrote state s1 (so the next dxm_advance must swap the buffers) and
 * the flux / tangent / stats.  Call after every replay, before dxm_advance / dxm_get_stats / dxm_get_state. */
int dxm_notify_replay(dxm_material* m);
/* Per-handle options (no environment variables are read by the library):
 *   "pipeline"       1 | 0   host-buffer form: chunked upload / kernel / download on several streams (default 1)
 *   "split_streams"  1 | 0   how the chunks use the streams when the gradient array is page-locked (DMA uploads).  1 (default):
 *                            uploads and kernels of all chunks on one stream, the downloads of chunk c on one of two others
 *                            behind an event (7 sqrt(npoints / 1e6) chunks, at most 24): the device-to-host direction, 80-136 B/point against 48 up, does
 *                            not wait behind uploads queued on its own stream.  0: whole chunks alternate between two streams
 *                            (rounds 1-5; still what staged uploads and the displacement forms use)
 *   "max_chunks"     1..64   upper bound on the chunks of that pipeline (default 64)
 *   "packed_transfer" 2|1|0  host-buffer form, full tangent layout, >= packed_min_points.  1: move the 9 coefficients
 *                            of Ct = c1 1x1 + c2 I + c3 n x n (72 instead of 288 B/point; nothing for the elastic
 *                            law) / for the FeFp laws the 54 building blocks of the 9x9 tangent (432 instead of
 *                            648 B/point), and rebuild the block on the host with the kernel's own expression.
 *                            2 (default): the J2 laws move (c1, c2, c3, w) only, 32 B/point -- the kernels build
 *                            the tangent with n = dev(stress) w, so the host rebuilds n from the stress it receives
 *                            anyway (needs a flux destination in page-locked / registered memory, else as 1); FeFp
 *                            as 1.  0: the full block crosses PCIe.  All three deliver the same bits.  A J2 handle with the
 *                            DXM_TANGENT_SYM layout: 2 moves (c1, c2, c3, w) as well and the workers rebuild its 21 entries
 *                            (32 instead of 168 B/point over PCIe); 1 | 0 download the kernel's 21 entries
 *   "register_input" 2 | 1 | 0  host-buffer form, gradient array in ordinary (pageable) memory.  2: page-locked for the
 *                            duration of the call (hipHostRegister ... hipHostUnregister before the call returns) and
 *                            uploaded by DMA: ~1 ms per 480 MB on transparent huge pages (numpy's default).  Arrays on
 *                            4 KiB pages register slowly (7-17 ms): after three such registrations in a row the handle
 *                            stages the next 20 calls.  0: always staged through the page-locked ring by the worker
 *                            threads.  1 (default): the handle measures -- calls 2-5 alternate between the two, the
 *                            faster (by the whole call; page-locking wins within 5 %) is kept and the other one is
 *                            tried again once in 32 calls; which one wins depends on the host (2-3 ms either way on
 *                            most, 5-9 ms for staging on some).  dxm_stats.upload reports what each call did
 *   "keep_initial_io" 0 | 1  dxm_advance keeps the device copies of gradient and flux of the accepted state as those of
 *                            s0 (dxm_get_io above); default 0
 *   "query_foreign_pointers" 1 | 0  (process-wide) host pointers that this library did not page-lock itself
 *                            (dxm_host_alloc / dxm_host_register) are looked up with hipPointerGetAttributes (1,
 *                            default) or treated as pageable and staged (0)
 *   "packed_min_points" >= 0 batch size from which packed_transfer applies (default 32768: below, waking the
 *                            worker threads costs what the bytes save)
 *   "pageable_dma"   0 | 1   host-buffer form: 1 hands gradient / result arrays in ordinary (pageable) memory to the
 *                            runtime's own transfer path instead of staging them through page-locked memory.  Faster
 *                            (the strain upload then costs the host nothing), but exposed to the runtime's cache of
 *                            on-the-fly page-locked ranges: with host arrays that are freed and reallocated between
 *                            calls, about one process in 25 of the test suite died with "Memory access fault by GPU"
 *                            (DESIGN.md section 1).  Default 0
 *   "host_threads"   1..256  worker threads of that rebuild and of the staging copies (default 16)
 *   "fused_gradient" 1 | 0   displacement forms: evaluate the gradient inside the update kernel where the mesh
 *                            allows (default 1)
 *   "blocks_per_cu"  1..256  grid size of the update kernel in workgroups per CU (default 32 small strain,
 *                            the resident 2 for FeFp)
 *   "verbose"        0 | 1   host-buffer form: log the chunk timeline, page-locking and upload-mode decisions to stderr (default 0) */
int dxm_set_option(dxm_material* m, const char* name, double value);
/* get_initial_state_dict / get_final_state_dict without a device array of the caller: packs the
 * user-visible ISVs of state `which` and downloads them into host memory (npoints, n_isv_total).  This is
 * how the Python layer serves the `isv` array of integrate() on first access (jaxmat.py:227-229). */
int dxm_isv_host(dxm_material* m, int which, double* isv_aos);
/* QuadratureMap.update writes every internal state variable into its quadrature Function after each integrate
 * (quadrature_map.py:332, :343-348: `_update_vals(isv, isv_vals[:, buff:buff+dim], cells)`).  Bind the memory of such a Function --
 * C-contiguous (npoints, dim) rows of field `field`, page-locked by dxm_host_alloc / dxm_host_register -- and the host-buffer
 * forms (dxm_integrate, dxm_integrate_displacement, ..._rows) deliver that field of the final state into it chunk by chunk inside
 * their transfer pipeline: the caller finds the Functions written when the call returns, without a second pass over the state.
 * NULL unbinds (do so before un-page-locking the array).  The device-pointer forms do not deliver.
 * In the rows forms (dxm_integrate_rows, dxm_integrate_displacement_rows) the bound pointer is, like their flux / tangent arguments,
 * the BASE of the Function over all cells: the field of point i goes to its row rows[i] (stored by the worker threads from the
 * library's own page-locked landing area).  A handle serves either kind of call with a given binding, not both. */
int dxm_bind_isv_output(dxm_material* m, int field, double* host_aos);

/* ---- pinned host memory for the host-buffer form --------------------------------------------
 * integrate() returns arrays owned by the material (the reference returns views of its state
 * manager: generic.py:185-189); allocating them page-locked lets the D2H copies run at full PCIe
 * rate without the runtime's staging copy.  Returns NULL on failure.
 * Output arrays in ordinary (pageable) memory are accepted everywhere, but the library never lets the GPU write into
 * them directly: they are filled through its own page-locked staging and a CPU copy, after the transfers of the call
 * (slower; and not subject to the runtime's cache of on-the-fly page-locked ranges -- DESIGN.md section 1). */
void* dxm_host_alloc(uint64_t bytes);
int dxm_host_free(void* p);
/* Page-lock an existing host range in place (e.g. the `x.array` of the dolfinx quadrature Functions that
 * QuadratureMap.update scatters into, quadrature_map.py:331-334, utils.py:136-143), so that integrate()
 * can deliver straight into it at full PCIe rate; undo with dxm_host_unregister before the memory is freed. */
/* bytes from src to dst (host memory, non-overlapping) on `threads` threads (<= 0: 8): a 480 MB numpy copy takes 50 ms on one
 * core; the Python layer snapshots bound gradient / flux arrays with this when an increment is accepted. */
int dxm_host_copy(void* dst, const void* src, uint64_t bytes, int threads);
/* Rows of `width` doubles through an index on `threads` threads (<= 0: 8), host memory, no overlap between dst and src:
 * scatter: dst[rows[i]] = src[i]; gather: dst[i] = src[rows[i]] for i < n.  What a QuadratureMap over a SUBSET of the cells does
 * with every array per update (utils.py:136-143 `array[index] = values`, quadrature_map.py:271 `_get_vals(f)[self.dofs]`);
 * the index holds each row once (the caller's responsibility, as in the reference) and is not range-checked. */
int dxm_host_scatter_rows(double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads);
int dxm_host_gather_rows(double* dst, const double* src, const int64_t* rows, int64_t n, int width, int threads);
/* Smallest and largest entry of such an index on `threads` threads (<= 0: 8): the range check that dxm_integrate_rows and the
 * two calls above leave to the caller (~1 ms per 1e7 entries).  n == 0: *lo = INT64_MAX, *hi = INT64_MIN. */
int dxm_host_index_range(const int64_t* rows, int64_t n, int threads, int64_t* lo, int64_t* hi);
int dxm_host_register(void* p, uint64_t bytes);
int dxm_host_unregister(void* p);

/* ---- gradient evaluation on device (the step before the path; hex8, tet4, Lagrange simplices of any order) --
 * Replaces, for the device-resident flow, QuadratureExpression.eval -> fem.Expression.eval
 * (quadrature_function.py:45-51; consumer quadrature_map.py:247-253): only the displacement
 * vector crosses PCIe, the (npoints, n_grad) gradient array is produced in HBM in the layout
 * the constitutive kernels read (point = cell * nqp + q). */
typedef struct dxm_mesh dxm_mesh; /* opaque: device copies of coordinates and connectivity */
/* coords (n_nodes,3) fp64, conn (n_cells,8) int32 with the corner order
 * (-,-,-)(+,-,-)(+,+,-)(-,+,-)(-,-,+)(+,-,+)(+,+,+)(-,+,+); qpoints (nqp,3) in [-1,1]^3, nqp <= 27. */
dxm_mesh* dxm_mesh_create_hex8(const double* coords, int64_t n_nodes, const int32_t* conn,
                               int64_t n_cells, const double* qpoints, int nqp, int device);
/* First-order tetrahedra: conn (n_cells,4); the gradient is constant per cell and repeated at the
 * cell's nqp Gauss points (point = cell * nqp + q). */
dxm_mesh* dxm_mesh_create_tet4(const double* coords, int64_t n_nodes, const int32_t* conn,
                               int64_t n_cells, int nqp, int device);
/* Lagrange displacement of any order on STRAIGHT-SIDED simplices (tdim 3: tetrahedra, tdim 2: triangles, embedded as
 * plane strain: eps_zz = 0 / F_zz = 1) -- the P2 spaces of the reference's demos
 * (demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:115-117 tet10 with 4 points,
 * demos/jax/elastoplasticity/plane_elastoplasticity.py:96-100 tri6 with 3 points).  The geometry map is affine and
 * comes from the tdim+1 vertices of each cell: coords (n_vertices,3) [dolfinx geometry.x is 3-wide in 2-D too],
 * geom_conn (n_cells, tdim+1).  The displacement has its own dofmap (n_cells, nd) into a vector of n_dofs blocks of tdim
 * components [u.x.array of a (tdim,)-shaped space], and dphi (nqp, nd, tdim) holds the reference derivatives
 * d N_m / d xi_d of its nd shape functions at the nqp quadrature points, in the dofmap's local ordering
 * [basix: element.tabulate(1, points)[1:, :, :, 0] transposed to (point, dof, direction)]. */
dxm_mesh* dxm_mesh_create_simplex(int tdim, const double* coords, int64_t n_vertices, const int32_t* geom_conn,
                                  int64_t n_cells, const int32_t* dofmap, int nd, int64_t n_dofs,
                                  const double* dphi, int nqp, int device);
int dxm_mesh_destroy(dxm_mesh* mesh);
int64_t dxm_mesh_npoints(const dxm_mesh* mesh);
/* doubles in the displacement vector the mesh expects (3 per node; tdim per dof for dxm_mesh_create_simplex) */
int64_t dxm_mesh_displacement_size(const dxm_mesh* mesh);
/* kind 0: Mandel strain (6); kind 1: deformation gradient F = I + grad u (9).  u_dev: device
 * (dxm_mesh_displacement_size doubles); grad_dev: device (npoints, 6|9).  Asynchronous on hip_stream. */
int dxm_mesh_gradient_device(dxm_mesh* mesh, const double* u_dev, int kind, double* grad_dev,
                             void* hip_stream);
/* dxm_integrate with the gradient computed on the device from the host displacement vector
 * u_host (dxm_mesh_displacement_size doubles): uploads u, evaluates the law's gradient, runs the constitutive update,
 * downloads flux / isv / tangent (any may be NULL).  mesh npoints must equal the handle's. */
int dxm_integrate_displacement(dxm_material* m, dxm_mesh* mesh, const double* u_host, double dt,
                               double* flux_aos, double* isv_aos, double* ct_aos, dxm_stats* stats);
/* The same with the results delivered into rows of larger arrays, as dxm_integrate_rows does: `mesh` holds the cells of ONE
 * material's map (connectivity restricted to its cells, coordinates / displacement vector of the whole mesh), flux_rows /
 * ct_rows are the quadrature Functions over all cells, point i goes to their row rows[i]. */
int dxm_integrate_displacement_rows(dxm_material* m, dxm_mesh* mesh, const double* u_host, double dt, double* flux_rows,
                                    double* ct_rows, const int64_t* rows, dxm_stats* stats);

/* Device-resident form of dxm_integrate_displacement: u_dev (dxm_mesh_displacement_size doubles), flux_dev, ct_dev are device
 * arrays, the call is asynchronous on hip_stream and capturable like dxm_integrate_device.  For
 * hexahedra with 8 Gauss points per cell, for tetrahedra and for Lagrange simplices the gradient is evaluated inside the
 * update kernel (no strain / deformation-gradient array is written or read); otherwise the gradient kernel fills a scratch
 * array owned by the handle and the update kernel follows on the same stream. */
int dxm_integrate_displacement_device(dxm_material* m, dxm_mesh* mesh, const double* u_dev, double dt,
                                      double* flux_dev, double* ct_dev, void* hip_stream);


#ifdef __cplusplus
}
#endif
#endif /* DXMAT_H */
